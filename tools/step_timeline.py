"""One training step as a timeline from a rocprofv3 --kernel-trace database (rocpd SQLite):
    python tools/step_timeline.py run_results.db [anchor-kernel-substring] [which-occurrence]
Prints every dispatch between two consecutive launches of the anchor kernel (default: adam_kernel): start offset, duration,
idle gap since the previous dispatch ended, grid, name.  The sum of durations and of gaps shows where a step's time goes."""
import sqlite3, sys

con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
pick = lambda *names: next(n for n in names if n in cols)
c_start, c_end = pick("start", "start_timestamp"), pick("end", "end_timestamp")
gcols = [c for c in ("grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x") if c in cols]
rows = list(con.execute(f"select name, {c_start}, {c_end} {''.join(', ' + g for g in gcols)} from kernels order by {c_start}"))
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
lo, hi = marks[which - 1] + 1, marks[which] + 1
t0 = rows[lo][1]
prev_end = rows[lo - 1][2]
busy = idle = 0
for name, s, e, *g in rows[lo:hi]:
    gap = s - prev_end
    busy += e - s
    idle += max(gap, 0)
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("ctvae::", "").split("(")[0]
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  {'x'.join(str(v) for v in g):>12}  {short[:90]}")
    prev_end = max(prev_end, e)
print(f"# {hi - lo} dispatches, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us, span {(rows[hi - 1][2] - rows[lo - 1][2]) / 1e3:.1f} us")
