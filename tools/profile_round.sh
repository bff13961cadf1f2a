# One round's measurement pass on the GPU box:  gpurun -- 'TAG=r03 bash tools/profile_round.sh'
#   bench line (headline = the metric's own VanillaVAE bs=64 + configs array), rocprofv3 kernel traces and one-step timelines of
#   the five configurations, PMC passes (separate runs, with --kernel-trace only, as gpurun requires) for HBM traffic and MFMA
#   busy cycles on VanillaVAE bs=64 (headline), bs=256 and CT-MCQ-VAE A=12.   PARTS="bench trace pmc" selects a subset.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${TAG:-r03}
parts=${PARTS:-"bench trace pmc"}
out=gpurun_out/$tag
mkdir -p $out
case " $parts " in *" bench "*)
python bench.py > $out/bench_line.json 2> $out/bench.err
python tools/show_bench.py $out/bench_line.json 6 || true
;; esac
prof() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/$tag.$name -o run -- python3 bench.py --no-configs --no-cpu-baseline "$@" > $out/bench_line_${name}_under_rocprof.json 2> $out/$name.log
  python tools/rocpd_export.py /tmp/$tag.$name/run_results.db > $out/${name}_kernel_stats.csv
  python tools/step_timeline.py /tmp/$tag.$name/run_results.db adam_kernel -40 > $out/step_timeline_${name}.txt || true
  echo "traced $name"
}
case " $parts " in *" trace "*)
prof vanilla_bs64 --no-roofline
prof vanilla_bs256 --batch 256 --no-roofline
prof mcqvae_bs256 --model MCQVAE --no-roofline
prof ctmcqvae_a12_bs128 --model CTMCQVAE --batch 128 --no-roofline
prof ctmcqvae_a20_bs128 --model CTMCQVAE --batch 128 --action-dim 20 --no-roofline
;; esac
pmc() {    # name, counters, bench args...
  local name=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace -d $out/pmc_$name -o run -- python3 bench.py --no-roofline --no-cpu-baseline --no-configs --no-graph --steps 5 --warmup 2 "$@" > $out/pmc_$name.json 2> $out/pmc_$name.log
  echo "pmc $name"
}
case " $parts " in *" pmc "*)
pmc f64 FETCH_SIZE
pmc w64 WRITE_SIZE
pmc m64 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc f FETCH_SIZE --batch 256
pmc w WRITE_SIZE --batch 256
pmc m "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --batch 256
pmc ct_f FETCH_SIZE --model CTMCQVAE --batch 128
pmc ct_w WRITE_SIZE --model CTMCQVAE --batch 128
python tools/pmc_summary.py $out/pmc_f64 $out/pmc_w64 7 "VanillaVAE bs=64" > $out/pmc_traffic_bs64.json
python tools/pmc_summary.py $out/pmc_f $out/pmc_w 7 "VanillaVAE bs=256" > $out/pmc_traffic.json
python tools/pmc_summary.py $out/pmc_ct_f $out/pmc_ct_w 7 "CTMCQVAE bs=128 a12" > $out/ct_pmc_traffic.json
python tools/mfma_util_summary.py $out/pmc_m64 > $out/mfma_util_bs64.json
python tools/mfma_util_summary.py $out/pmc_m > $out/mfma_util.json
rm -rf $out/pmc_f $out/pmc_w $out/pmc_m $out/pmc_f64 $out/pmc_w64 $out/pmc_m64 $out/pmc_ct_f $out/pmc_ct_w     # the rocpd databases are large; the summaries are what profiles/ keeps
;; esac
