set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -30 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
rocprofv3 --kernel-trace -d /tmp/tlv -o run -- python3 bench.py --no-configs --no-cpu-baseline --no-roofline --model MCQVAE > gpurun_out/tl_m.json 2> gpurun_out/tl_m.err
python tools/step_timeline.py /tmp/tlv/run_results.db > gpurun_out/timeline_mcq2.txt
python bench.py --no-cpu-baseline > gpurun_out/b_all.json 2> gpurun_out/b_all.err
python tools/show_bench.py gpurun_out/b_all.json 4
