# usage: VAR=CTVAE_WGRAD_WGS VALS="1024 768 512 384" [ARGS="--model MCQVAE"] bash tools/sweep_env.sh   (bench ms/step per value, 3 runs each)
cd $GRAFT_REPO_ROOT
for v in $VALS; do
  for i in 1 2 3; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-configs --no-roofline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'])"
  done
done
