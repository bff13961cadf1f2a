"""Instruction mix of the innermost loops of a kernel (diagnostic: spots accumulator copies, register moves and VALU
address arithmetic inside MFMA loops).  Usage: python tools/isa_loop_mix.py ct-vae_amd/csrc/wino.hip wino_wgrad_kernel"""
import collections, re, subprocess, sys, tempfile

src, pat = sys.argv[1], sys.argv[2]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", f.name],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)]
for s in starts:
    e = s
    while e < len(lines) and "s_endpgm" not in lines[e]:
        e += 1
    body = lines[s:e]
    print("==", lines[s].split(":")[0][:100], f"({len(body)} lines)")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for i, l in enumerate(body):
        m = re.search(r"s_(?:cbranch\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:          # backward branch = loop
            seg = [x.split()[0] for x in body[labels[m.group(1)]:i] if x.strip() and not x.strip().startswith((";", "."))]
            c = collections.Counter(seg)
            n_mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            if n_mfma:
                valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith(("v_mfma", "v_accvgpr")))
                print(f"  loop {m.group(1)}: {len(seg)} instrs, {n_mfma} MFMA, {valu} other VALU, "
                      f"{sum(v for k, v in c.items() if k.startswith('v_accvgpr'))} accvgpr, "
                      f"{sum(v for k, v in c.items() if k.startswith('ds_'))} LDS, "
                      f"{sum(v for k, v in c.items() if k.startswith(('buffer_', 'global_')))} global, {c.get('s_barrier', 0)} barriers")
                print("     top:", ", ".join(f"{k} {v}" for k, v in c.most_common(10)))
