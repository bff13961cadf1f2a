"""rocprofv3 on this image writes a rocpd SQLite database (<dir>/<name>_results.db) instead of CSV files.  This exports what
profiles/ keeps: the --stats style per-kernel table (`kernels` view) as CSV.
    python tools/rocpd_export.py gpurun_out/r01g/stats/run_results.db > profiles/r01g_vanilla_bs256_kernel_stats.csv"""
import collections, csv, math, sqlite3, sys

con = sqlite3.connect(sys.argv[1])
agg = collections.defaultdict(list)
for name, dur in con.execute("select name, duration from kernels"):
    agg[name].append(float(dur))
tot = sum(sum(v) for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    n, s = len(v), sum(v)
    mean = s / n
    sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
    w.writerow([name, n, int(s), f"{mean:.6f}", f"{100.0 * s / tot:.4f}", int(min(v)), int(max(v)), f"{sd:.6f}"])
