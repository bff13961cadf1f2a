"""Summarise a rocprofv3 kernel-stats CSV (tools/rocpd_export.py): per-step time per kernel, share of library kernels."""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = [r for r in csv.DictReader(open(path)) if "spin_kernel" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "ctvae" in r["Name"])
print(f"GPU time per step {tot / steps / 1e3:.1f} us; library kernels {ours / tot * 100:.1f} %, other {100 - ours / tot * 100:.1f} %")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[:n]:
    c = int(r["Calls"])
    t = float(r["TotalDurationNs"])
    name = r["Name"].replace("ctvae::(anonymous namespace)::", "").replace("ctvae::", "")
    print(f"{t / tot * 100:5.1f}% {c / steps:6.1f}/step {t / c / 1e3:8.1f}us {t / steps / 1e3:8.1f}us/step {'*' if 'ctvae' in r['Name'] else ' '} {name[:100]}")
