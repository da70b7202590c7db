# Copies what the last `gpurun -- bash tools/gpu_r4.sh final <tag>` left under gpurun_out/ (scratch) to profiles/ under round-5 names.
#   bash tools/keep_r05.sh <tag>
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
TAG="${1:-r05}"
O="gpurun_out/${TAG}_prof"
for c in eagle_catch displacement push_slide; do
  [ -f "$O/pmc_$c.json" ] && cp "$O/pmc_$c.json" "profiles/r05_pmc_$c.json"
  [ -f "$O/kernel_stats_$c.csv" ] && cp "$O/kernel_stats_$c.csv" "profiles/r05_kernel_stats_${c}_B1024.csv"
  [ -f "$O/bench_$c.json" ] && cp "$O/bench_$c.json" "profiles/r05_bench_$c.json"
  [ -f "$O/bench_under_rocprof_$c.json" ] && cp "$O/bench_under_rocprof_$c.json" "profiles/r05_bench_under_rocprof_$c.json"
done
for c in hover carrot_mpc rail_mpc weighted_mpc; do
  [ -s "gpurun_out/${TAG}_bench_$c.json" ] && cp "gpurun_out/${TAG}_bench_$c.json" "profiles/r05_bench_$c.json"
done
[ -s "gpurun_out/${TAG}_bench_default.json" ] && cp "gpurun_out/${TAG}_bench_default.json" "profiles/r05_bench_default_with_traffic.json"
[ -f "gpurun_out/${TAG}_pytest.log" ] && cp "gpurun_out/${TAG}_pytest.log" "profiles/r05_pytest_gpu.log"
mkdir -p profiles/r05_stepwise
cp gpurun_out/parity/r05_stepwise_*.json profiles/r05_stepwise/ 2>/dev/null || true
# round 5: first-run reports, variant A/B lines, probes, stamps
for f in r05_margin_profile.json r05_poisoned_inputs_displacement.json r05_poisoned_inputs_eagle_catch.json; do
  [ -s "gpurun_out/parity/$f" ] && cp "gpurun_out/parity/$f" "profiles/$f"
done
for f in gpurun_out/${TAG}_pytest_*.log gpurun_out/${TAG}_probes.log gpurun_out/${TAG}_stamps.log; do
  [ -s "$f" ] && cp "$f" "profiles/r05_$(basename "$f" | sed "s/^${TAG}_//")"
done
mkdir -p profiles/r05_variants
for f in gpurun_out/${TAG}_bench_*_shipped.json gpurun_out/${TAG}_bench_*_r4b.json gpurun_out/${TAG}_bench_*_gap.json gpurun_out/${TAG}_bench_*_bits.json; do
  [ -s "$f" ] && cp "$f" "profiles/r05_variants/$(basename "$f" | sed "s/^${TAG}_//")"
done
ls -la profiles | grep r05_ | head -60
