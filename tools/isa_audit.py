"""What the compiler made of the kernels' memory pipelines -- the checks that found the round-3 stalls:

    python tools/isa_audit.py [file.hip ...]            (no GPU needed: hipcc -S for gfx950, all of csrc/ by default)

Per kernel: VGPR / AGPR / scratch bytes, instruction and branch counts (an unrolled epilogue that asks `if (a.option)` per value is a
branch tree per value: 600 branches for the tile kernel's sixteen values, DESIGN.md 4.8 "Epilogues"), the size of the kernel-argument segment and the number of scalar-load batches (s_load ...
s_waitcnt lgkmcnt(0)) in front of the first vector load -- with large argument structs each batch is a dependent round trip to memory
behind the launch's cache invalidate (tile kernels: eight, 2.9 us per workgroup, until common.hpp kernarg_warm fetched all lines at
once).  Per loop of a kernel (label to back-branch):
  * `scratch` -- spill reloads inside the loop: each is a vector-memory operation, and the `s_waitcnt vmcnt(0)` in front of its use
    also drains every load issued ahead for the next iteration (up_wgrad_kernel: 45 -> 29 us after a budget of one wave per SIMD);
  * `vmcnt(0)` next to global / buffer loads in a loop that also stores -- a predicated store (or load) is a branch, behind which
    the waitcnt pass cannot count outstanding operations and waits for all of them, i.e. for the previous iteration's store
    (gat_proj_bwd, pair_mlp_bwd64: unconditional stores / clamped loads, first iteration outside the loop);
  * `serial` -- a short loop with ONE load, a `vmcnt(0)` and an LDS write or add: `for (e = tid; e < N; e += 256) lds[e] = g[e]`
    is not unrolled (the trip count depends on tid), so it is a memory round trip per iteration (vq_cb_reduce 19.8 -> 4.7 us);
  * `v_mov` -- rotating register copies for read-ahead that were not renamed away (GATv2 pair loop: ten per channel).
Flags are hints for reading the ISA, not verdicts: an element-wise grid-stride loop legitimately waits for its own loads."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ct-vae_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"ctvae::\(anonymous namespace\)::|ctvae::", "", out)


def audit(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + CSRC,
                            "-I" + os.path.join(ROOT, "include"), path, "-o", tmp.name], capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{os.path.basename(path)}: hipcc failed\n{r.stderr[-400:]}")
            return
        txt = open(tmp.name).read()
    for fn in re.split(r"\n(?=_Z\w+:\s)", txt):
        m = re.match(r"(_Z\w+):", fn)
        if not m:
            continue
        sym = m.group(1)

        def meta(key):
            mm = re.search(re.escape(sym) + r"\." + key + r", (\d+)", txt)
            return int(mm.group(1)) if mm else 0

        lines = fn.split("s_endpgm")[0].split("\n")
        batches, pending, warm = 0, False, False
        for l in lines:
            if re.search(r"\b(global|buffer)_load", l):
                break
            if ".Lkw_off" in l:
                warm = True
            if "s_load" in l:
                pending = True
            if "s_waitcnt" in l and "lgkmcnt(0)" in l and pending:
                batches, pending = batches + 1, False
        ninstr = sum(1 for l in lines if l.strip() and not l.strip().startswith((";", ".")) and not l.rstrip().endswith(":"))
        nbranch = sum("s_cbranch" in l for l in lines)
        mk = re.search(r"\.kernarg_segment_size: (\d+)\n(?:.*\n){0,40}?\s+\.symbol:\s+" + re.escape(sym) + r"\.kd", txt)
        karg = int(mk.group(1)) if mk else 0
        rows = []
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)", l)
            if not mm:
                continue
            lab, j = mm.group(1), i + 1
            while j < len(lines) and not re.search(r"s_c?branch\w* " + re.escape(lab) + r"\b", lines[j]):
                j += 1
            body = "\n".join(lines[i:j + 1])
            n = j - i
            loads = len(re.findall(r"(global|buffer)_load", body))
            stores = len(re.findall(r"(global|buffer)_store", body))
            waits = re.findall(r"vmcnt\((\d+)\)", body)
            scr = len(re.findall(r"scratch_", body))
            mfma = len(re.findall(r"v_mfma", body))
            vmov = len(re.findall(r"v_mov_b", body))
            flags = []
            if scr:
                flags.append("scratch")
            if loads and stores and "0" in waits and (mfma or n < 600):
                flags.append("vmcnt(0)+store")
            if n < 120 and loads == 1 and "0" in waits:
                flags.append("serial")
            if mfma == 0 and vmov >= 16 and n < 600:
                flags.append("v_mov")
            if flags or mfma:
                rows.append(f"    loop {lab:11s} {n:5d} lines  mfma {mfma:3d}  loads {loads:3d}  stores {stores:3d}  vmcnt {','.join(waits[:10]) or '-':22s}"
                            f"  scratch {scr:2d}  v_mov {vmov:3d}  {' '.join(flags)}")
        scratch = meta("private_seg_size")
        print(f"{os.path.basename(path)}: {demangle(sym)[:100]}\n    vgpr {meta('num_vgpr')}  agpr {meta('num_agpr')}  scratch {scratch} B   {ninstr} instructions, {nbranch} branches   kernarg {karg} B, {batches} scalar-load batches before the first vector load{' (argument lines fetched in one batch)' if warm else ''}")
        if rows:
            print("\n".join(rows))


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for f in files:
        audit(f)
