set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q > gpurun_out/t_def.log 2>&1 || { tail -20 gpurun_out/t_def.log; exit 1; }
tail -2 gpurun_out/t_def.log
for w in 1 0 1 0; do
  echo "defer with bn rider $w"
  CTVAE_DEFER_WITH_BN_RIDER=$w python bench.py --no-cpu-baseline --no-configs > gpurun_out/b_w_$w.json 2>/dev/null
  python tools/show_bench.py gpurun_out/b_w_$w.json 0
done
