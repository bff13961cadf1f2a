# One-step dispatch timeline of a bench configuration on the GPU box:  gpurun -- 'bash tools/tl.sh NAME [bench args...]'
#   -> gpurun_out/tl_NAME.txt (tools/step_timeline.py of a rocprofv3 kernel trace) and the untraced ms/step on stdout
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
name=$1; shift
rm -rf /tmp/tl_$name
python bench.py --no-configs --no-cpu-baseline --no-roofline --steps 100 "$@" 2>/dev/null | python -c "import json,sys; print('$name ms/step', json.loads(sys.stdin.read())['ms_per_step'])"
rocprofv3 --kernel-trace -d /tmp/tl_$name -o run -- python3 bench.py --no-configs --no-cpu-baseline --no-roofline "$@" > /dev/null 2> gpurun_out/tl_$name.log
python tools/step_timeline.py /tmp/tl_$name/run_results.db adam_kernel -5 > gpurun_out/tl_$name.txt
tail -1 gpurun_out/tl_$name.txt
