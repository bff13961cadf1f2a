# CT-MCQ-VAE step profile: bench lines (A=12, A=20) + rocprofv3 kernel trace of the A=12 step.  TAG names the output dir.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${TAG:-ct}
mkdir -p $out
python bench.py --model CTMCQVAE --batch 128 --no-cpu-baseline --no-configs > $out/ct12.json 2> $out/ct12.err
python bench.py --model CTMCQVAE --batch 128 --action-dim 20 --no-cpu-baseline --no-configs > $out/ct20.json 2> $out/ct20.err
rocprofv3 --kernel-trace --stats -d /tmp/ct_stats_$$ -o run -- python3 bench.py --model CTMCQVAE --batch 128 --no-cpu-baseline --no-configs --no-roofline > $out/ct12_under_rocprof.json 2> $out/ct_stats.log
python tools/rocpd_export.py /tmp/ct_stats_$$/run_results.db > $out/ct12_kernel_stats.csv
python - <<PY
import json
for f in ("ct12", "ct20"):
    d = json.load(open("$out/%s.json" % f))
    print(f, d["ms_per_step"], d["value"])
PY
