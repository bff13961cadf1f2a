set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/${TAG:-r01g}
python bench.py > gpurun_out/${TAG:-r01g}/bench_line.json 2> gpurun_out/${TAG:-r01g}/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG:-r01g}/stats -o run -- python3 bench.py > gpurun_out/${TAG:-r01g}/bench_line_under_rocprof.json 2> gpurun_out/${TAG:-r01g}/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/${TAG:-r01g}/pmc_f -o run -- python3 bench.py --no-roofline --no-cpu-baseline --no-graph --steps 5 --warmup 2 > gpurun_out/${TAG:-r01g}/pmc_f.json 2> gpurun_out/${TAG:-r01g}/pmc_f.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/${TAG:-r01g}/pmc_w -o run -- python3 bench.py --no-roofline --no-cpu-baseline --no-graph --steps 5 --warmup 2 > gpurun_out/${TAG:-r01g}/pmc_w.json 2> gpurun_out/${TAG:-r01g}/pmc_w.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/${TAG:-r01g}/pmc_m -o run -- python3 bench.py --no-roofline --no-cpu-baseline --no-graph --steps 5 --warmup 2 > gpurun_out/${TAG:-r01g}/pmc_m.json 2> gpurun_out/${TAG:-r01g}/pmc_m.log
python bench.py --model MCQVAE > gpurun_out/${TAG:-r01g}/bench_line_mcqvae_bs256.json 2> gpurun_out/${TAG:-r01g}/mcq.err
python bench.py --model CTMCQVAE --batch 128 > gpurun_out/${TAG:-r01g}/bench_line_ctmcqvae_action_bs128.json 2> gpurun_out/${TAG:-r01g}/ct.err
find gpurun_out/${TAG:-r01g} -name "*.csv" | head -20
cut -c1-400 gpurun_out/${TAG:-r01g}/bench_line.json
