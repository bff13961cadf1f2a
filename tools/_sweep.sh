set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q > gpurun_out/t_def.log 2>&1 || { tail -25 gpurun_out/t_def.log; exit 1; }
tail -2 gpurun_out/t_def.log
for cfg in 1 0 1 0; do
  echo "up_dgrad=$cfg"
  CTVAE_UP_DGRAD=$cfg python bench.py --no-cpu-baseline --no-configs > gpurun_out/b_w.json 2>/dev/null
  python tools/show_bench.py gpurun_out/b_w.json 30 | grep -E "ms/step|up_dgrad|4,1,1,1,true"
done
