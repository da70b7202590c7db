# Copies the files of the last tools/gpu_r3_profiles.sh / gpu_r3_check.sh run (gpurun_out/, scratch) into profiles/ (tracked)
# under round-3 names.  usage: bash tools/keep_profiles.sh
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
S=gpurun_out/r3prof
for CFG in eagle_catch displacement push_slide; do
  cp $S/bench_$CFG.json profiles/r03_bench_$CFG.json
  cp $S/bench_under_rocprof_$CFG.json profiles/r03_bench_under_rocprof_$CFG.json
  cp $S/kernel_stats_$CFG.csv profiles/r03_kernel_stats_${CFG}_B1024.csv
  for K in linearize backward rollout; do cp $S/traffic_${CFG}_$K.json profiles/traffic_${CFG}_$K.json; done
done
cp $S/pmc_sq_eagle_catch.csv profiles/r03_pmc_sq_eagle_catch.csv
cp $S/pmc_sq_push_slide.csv profiles/r03_pmc_sq_push_slide.csv
mkdir -p profiles/r03_stepwise
cp gpurun_out/parity/r03_stepwise_*.json profiles/r03_stepwise/
ls -la profiles | wc -l
