# usage: VAR=CTVAE_IMG_FWD_WGS VALS="768 1536" KERNEL=img_fwd_kernel bash tools/sweep_kernel.sh   (per-kernel avg us from the bench's event table)
cd $GRAFT_REPO_ROOT
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-configs $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=[(k,v['avg_us']) for k,v in d['kernels'].items() if '$KERNEL' in k]
print('$VAR=$v', d['ms_per_step'], ks)"
done
