set -e
cd $GRAFT_REPO_ROOT
for w in 0 1 2 0 1; do
  echo "dephase $w"
  CTVAE_IMG_FWD_DEPHASE=$w python bench.py --no-cpu-baseline --no-configs > gpurun_out/b_w_$w.json 2>/dev/null
  python tools/show_bench.py gpurun_out/b_w_$w.json 30 | grep -E "ms/step|img_fwd"
done
