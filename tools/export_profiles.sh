#!/bin/bash
# After `gpurun -- 'bash tools/profile_round.sh'` merged gpurun_out/<tag>/ back: export what profiles/ keeps.
#   bash tools/export_profiles.sh r01g
set -e
tag=${1:-r01g}
src=gpurun_out/$tag
python tools/rocpd_export.py $src/stats/run_results.db > profiles/${tag}_vanilla_bs256_kernel_stats.csv
python tools/pmc_summary.py $src/pmc_f $src/pmc_w 7 > profiles/${tag}_pmc_traffic.json
python tools/mfma_util_summary.py $src/pmc_m > profiles/${tag}_mfma_util.json
for f in bench_line bench_line_under_rocprof bench_line_mcqvae_bs256 bench_line_ctmcqvae_action_bs128; do
  cp $src/$f.json profiles/${tag}_$f.json
done
ls -la profiles/${tag}_*
