# One round's measurement pass on the GPU box:  gpurun -- 'TAG=r02 bash tools/profile_round.sh'
#   bench line (headline + configs array), rocprofv3 kernel traces of the four configurations, PMC passes (separate runs, with
#   --kernel-trace only, as gpurun requires) for HBM traffic and MFMA busy cycles on VanillaVAE bs=256 and CT-MCQ-VAE A=12.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${TAG:-r02}
out=gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_line.json 2> $out/bench.err
prof() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/$tag.$name -o run -- python3 bench.py --no-configs --no-cpu-baseline "$@" > $out/bench_line_${name}_under_rocprof.json 2> $out/$name.log
  python tools/rocpd_export.py /tmp/$tag.$name/run_results.db > $out/${name}_kernel_stats.csv
}
prof vanilla_bs256
prof mcqvae_bs256 --model MCQVAE
prof ctmcqvae_a12_bs128 --model CTMCQVAE --batch 128
prof ctmcqvae_a20_bs128 --model CTMCQVAE --batch 128 --action-dim 20
pmc() {    # name, counters, bench args...
  local name=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace -d $out/pmc_$name -o run -- python3 bench.py --no-roofline --no-cpu-baseline --no-configs --no-graph --steps 5 --warmup 2 "$@" > $out/pmc_$name.json 2> $out/pmc_$name.log
}
pmc f FETCH_SIZE
pmc w WRITE_SIZE
pmc m "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc ct_f FETCH_SIZE --model CTMCQVAE --batch 128
pmc ct_w WRITE_SIZE --model CTMCQVAE --batch 128
python tools/pmc_summary.py $out/pmc_f $out/pmc_w 7 > $out/pmc_traffic.json
python tools/pmc_summary.py $out/pmc_ct_f $out/pmc_ct_w 7 > $out/ct_pmc_traffic.json
python tools/mfma_util_summary.py $out/pmc_m > $out/mfma_util.json
rm -rf $out/pmc_f $out/pmc_w $out/pmc_m $out/pmc_ct_f $out/pmc_ct_w     # the rocpd databases are large; the summaries are what profiles/ keeps
python tools/show_bench.py $out/bench_line.json 6
