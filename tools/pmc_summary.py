"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: TCC has 4 slots, they cost 3 + 2).  Usage:
    python tools/pmc_summary.py <dir with FETCH pass> <dir with WRITE pass> <steps profiled> [workload label] > profiles/rNN_pmc_traffic.json
(the label, e.g. "VanillaVAE bs=64", is what bench.py matches a summary to a configuration by)
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: the counters are in KB; on gfx950
FETCH_SIZE reports half of the bytes of coalesced streaming reads (128-B requests tallied at 64 B) -> doubled;
WRITE_SIZE is taken as is."""
import collections, csv, glob, json, os, sys


def norm(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ctvae::", "")


def load(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if files:
        for r in csv.DictReader(open(files[0])):
            agg[norm(r["Kernel_Name"])][0] += 1
            agg[norm(r["Kernel_Name"])][1] += float(r["Counter_Value"])
        return agg
    import sqlite3   # rocprofv3 of ROCm 7.2 writes a rocpd database instead of CSV files
    con = sqlite3.connect(glob.glob(os.path.join(d, "*_results.db"))[0])
    for name, val in con.execute("select kernel_name, value from counters_collection"):
        agg[norm(name)][0] += 1
        agg[norm(name)][1] += float(val)
    return agg


fd, wd, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
workload = sys.argv[4] if len(sys.argv) > 4 else "VanillaVAE bs=256"
F, W = load(fd), load(wd)
out = {}
for k in sorted(F, key=lambda k: -(2 * F[k][1] + W.get(k, [0, 0])[1])):
    n = F[k][0]
    fetch_raw = F[k][1] * 1024.0 / n
    write = W.get(k, [1, 0.0])[1] * 1024.0 / max(W.get(k, [1, 0.0])[0], 1)
    out[k] = {"launches_per_step": round(n / steps, 2), "fetch_bytes_raw_per_launch": round(fetch_raw),
              "fetch_bytes_corrected_per_launch": round(2 * fetch_raw), "write_bytes_per_launch": round(write),
              "hbm_bytes_per_launch": round(2 * fetch_raw + write)}
json.dump({"note": "FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported; separate --pmc passes, eager launches",
           "workload": workload, "steps": steps, "kernels": out}, sys.stdout, indent=1)
