set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_ct_gpu.py tests/test_models_gpu.py -m gpu -x -q -k "adam or harness" > gpurun_out/t_adam.log 2>&1 || { tail -20 gpurun_out/t_adam.log; exit 1; }
tail -2 gpurun_out/t_adam.log
for w in 512 1024 2048; do
  CTVAE_ADAM_WGS=$w python bench.py --no-configs --no-cpu-baseline > gpurun_out/b_adam_$w.json 2>/dev/null
  echo "wgs $w"; python tools/show_bench.py gpurun_out/b_adam_$w.json 40 | grep -E "ms/step|adam"
done
