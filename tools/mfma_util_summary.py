"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
(rocpd database): util = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs) -- the gfx94x MfmaUtil formula, GRBM_GUI_ACTIVE
being summed over the 8 XCDs (MI355X_MICROARCH.md).  Short dispatches read high on the clock side (same guide), so the
figure is a per-kernel average over all its launches.
    python tools/mfma_util_summary.py gpurun_out/r01g/pmc_m > profiles/r01g_mfma_util.json"""
import collections, glob, json, os, sqlite3, sys

con = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "*_results.db"))[0])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for name, counter, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ctvae::", "")
    agg[k][counter] += float(val)
    if counter == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {}
for k, v in agg.items():
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 256 * 4)
    out[k] = {"launches": cnt[k], "mfma_busy_cycles_per_launch": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cnt[k]),
              "gui_active_per_xcd_per_launch": round(gui / 8.0 / cnt[k]), "mfma_util": round(util, 4)}
json.dump({"note": "MFMA busy cycles / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs); eager launches, VanillaVAE bs=256",
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))},
          sys.stdout, indent=1)
