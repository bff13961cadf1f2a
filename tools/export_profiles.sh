#!/bin/bash
# After `gpurun -- 'TAG=r03 bash tools/profile_round.sh'` merged gpurun_out/<tag>/ back: copy what profiles/ keeps.
#   bash tools/export_profiles.sh r03
set -e
tag=${1:-r03}
src=gpurun_out/$tag
[ -f $src/bench_line.json ] && cp $src/bench_line.json profiles/${tag}_bench_line.json
for n in vanilla_bs64 vanilla_bs256 mcqvae_bs256 ctmcqvae_a12_bs128 ctmcqvae_a20_bs128; do
  [ -f $src/${n}_kernel_stats.csv ] || continue
  cp $src/${n}_kernel_stats.csv profiles/${tag}_${n}_kernel_stats.csv
  cp $src/step_timeline_${n}.txt profiles/${tag}_step_timeline_${n}.txt
  cp $src/bench_line_${n}_under_rocprof.json profiles/${tag}_bench_line_${n}_under_rocprof.json
done
[ -f $src/pmc_traffic_bs64.json ] && cp $src/pmc_traffic_bs64.json profiles/${tag}_vanilla_bs64_pmc_traffic.json
[ -f $src/pmc_traffic.json ] && cp $src/pmc_traffic.json profiles/${tag}_pmc_traffic.json
[ -f $src/ct_pmc_traffic.json ] && cp $src/ct_pmc_traffic.json profiles/${tag}_ctmcqvae_a12_pmc_traffic.json
[ -f $src/mfma_util_bs64.json ] && cp $src/mfma_util_bs64.json profiles/${tag}_vanilla_bs64_mfma_util.json
[ -f $src/mfma_util.json ] && cp $src/mfma_util.json profiles/${tag}_mfma_util.json
ls -la profiles/${tag}_*
