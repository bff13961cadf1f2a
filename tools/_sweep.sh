set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -25 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
for cfg in 0 1 0 1; do
  echo "no_loss_grad_in_fwd=$cfg"
  CTVAE_NO_LOSS_GRAD_IN_FWD=$cfg python bench.py --no-cpu-baseline --no-configs > gpurun_out/b_w.json 2>/dev/null
  python tools/show_bench.py gpurun_out/b_w.json 40 | grep -E "ms/step|mse_partial|loss_bwd"
done
