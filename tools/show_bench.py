"""Print the headline and the top kernels of a bench.py JSON line:  python tools/show_bench.py file.json [n]"""
import json
import sys

d = json.load(open(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print(d["metric"], "|", d["ms_per_step"], "ms/step", d["value"], d["unit"], "| step frac", (d.get("roofline") or {}).get("step_frac_of_f32_mfma_peak"))
for c in d.get("configs", []):
    print("  ", c["workload"][:70], c["per_gpu_batch"], c["ms_per_step"], c["value"], c["unit"], c.get("step_frac_of_f32_mfma_peak"))
for k, v in list((d.get("kernels") or {}).items())[:n]:
    print(f"  {v['ms_per_step'] * 1e3:8.1f} us/step {v['launches_per_step']:5.1f}x {v['avg_us']:7.1f} us  {k}")
