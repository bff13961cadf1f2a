set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "deferred or conv_family or lazy" > gpurun_out/t_def.log 2>&1 || { tail -20 gpurun_out/t_def.log; exit 1; }
tail -2 gpurun_out/t_def.log
for w in 65536 262144 100000000; do
  echo "wide_items $w"
  CTVAE_REDUCE_WIDE_ITEMS=$w python bench.py --no-cpu-baseline > gpurun_out/b_w_$w.json 2>/dev/null
  python tools/show_bench.py gpurun_out/b_w_$w.json 0
done
echo "no defer"
CTVAE_NO_DEFER_REDUCE=1 python bench.py --no-cpu-baseline > gpurun_out/b_nodefer.json 2>/dev/null
python tools/show_bench.py gpurun_out/b_nodefer.json 0
