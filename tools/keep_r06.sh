#!/bin/bash
# Copies what one phase of tools/gpurun_r6.sh left under gpurun_out/ (scratch) to profiles/ under round-6 names.
#   bash tools/keep_r06.sh <tag> <phase>
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
TAG="${1:?tag}"; PH="${2:-check}"; O="gpurun_out/${TAG}_prof"
cp "gpurun_out/${TAG}_call.log" "profiles/r06_${PH}_call.log" 2>/dev/null || true
case "$PH" in
  check)
    [ -s "gpurun_out/${TAG}_pytest.log" ] && cp "gpurun_out/${TAG}_pytest.log" profiles/r06_pytest_gpu.log
    [ -s "gpurun_out/${TAG}_bench_default.json" ] && cp "gpurun_out/${TAG}_bench_default.json" profiles/r06_bench_default.json
    mkdir -p profiles/r06_stepwise; cp gpurun_out/parity/*.json profiles/r06_stepwise/ 2>/dev/null || true ;;
  profiles)
    for c in eagle_catch displacement push_slide; do
      [ -f "$O/pmc_$c.json" ] && cp "$O/pmc_$c.json" "profiles/r06_pmc_$c.json"
      [ -f "$O/kernel_stats_$c.csv" ] && cp "$O/kernel_stats_$c.csv" "profiles/r06_kernel_stats_${c}_B1024.csv"
      [ -f "$O/bench_$c.json" ] && cp "$O/bench_$c.json" "profiles/r06_bench_$c.json"
      [ -f "$O/bench_under_rocprof_$c.json" ] && cp "$O/bench_under_rocprof_$c.json" "profiles/r06_bench_under_rocprof_$c.json"
    done ;;
  experimental) [ -s "gpurun_out/${TAG}_pytest_experimental.log" ] && cp "gpurun_out/${TAG}_pytest_experimental.log" profiles/r06_pytest_experimental.log
                [ -s "gpurun_out/${TAG}_pytest_two_contacts.log" ] && cp "gpurun_out/${TAG}_pytest_two_contacts.log" profiles/r06_pytest_two_contacts.log ;;
  stamps) [ -s "gpurun_out/${TAG}_stamps.log" ] && cp "gpurun_out/${TAG}_stamps.log" profiles/r06_stamps.log ;;
  probes) [ -s "gpurun_out/${TAG}_probes.log" ] && cp "gpurun_out/${TAG}_probes.log" profiles/r06_probes.log ;;
  variants)
    mkdir -p profiles/r06_variants
    for f in gpurun_out/${TAG}_bench_*.json gpurun_out/${TAG}_pytest_*.log; do
      [ -s "$f" ] && cp "$f" "profiles/r06_variants/$(basename "$f" | sed "s/^${TAG}_//")"
    done ;;
  lines)
    for c in hover carrot_mpc rail_mpc weighted_mpc; do
      [ -s "gpurun_out/${TAG}_bench_$c.json" ] && cp "gpurun_out/${TAG}_bench_$c.json" "profiles/r06_bench_$c.json"
    done ;;
esac
ls profiles | grep "^r06_" | head -40
