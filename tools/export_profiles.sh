#!/bin/bash
# After `gpurun -- 'TAG=r02 bash tools/profile_round.sh'` merged gpurun_out/<tag>/ back: copy what profiles/ keeps.
#   bash tools/export_profiles.sh r02
set -e
tag=${1:-r02}
src=gpurun_out/$tag
cp $src/bench_line.json profiles/${tag}_bench_line.json
for n in vanilla_bs256 mcqvae_bs256 ctmcqvae_a12_bs128 ctmcqvae_a20_bs128; do
  cp $src/${n}_kernel_stats.csv profiles/${tag}_${n}_kernel_stats.csv
  cp $src/bench_line_${n}_under_rocprof.json profiles/${tag}_bench_line_${n}_under_rocprof.json
done
cp $src/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $src/ct_pmc_traffic.json profiles/${tag}_ctmcqvae_a12_pmc_traffic.json
cp $src/mfma_util.json profiles/${tag}_mfma_util.json
ls -la profiles/${tag}_*
