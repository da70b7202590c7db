# GPU box, round 3: the files kept under profiles/ -- bench lines (stream mode), rocprofv3 kernel-trace summaries of the same
# command, HBM traffic from separate --pmc passes, SQ counters of the hot kernels, for eagle_catch (headline), displacement and
# push_slide.  usage: bash tools/gpu_r3_profiles.sh
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
O="$ROOT/gpurun_out/r3prof"
rm -rf "$O"; mkdir -p "$O"
for CFG in eagle_catch displacement push_slide; do
  STEPS=20; [ "$CFG" = push_slide ] && STEPS=5
  timeout 900 python3 bench.py --config $CFG --steps $STEPS --warmup 1 --no-secondary > $O/bench_$CFG.json 2> $O/bench_$CFG.err
  tail -c 600 $O/bench_$CFG.json; echo
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-single-batch > $O/bench_under_rocprof_$CFG.json 2> $O/stats_$CFG.err
  python3 tools/profile_summarize.py stats $O/stats_$CFG $O/kernel_stats_$CFG.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$CFG -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-single-batch --mode batch > $O/fetch_$CFG.json 2> $O/fetch_$CFG.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$CFG -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-single-batch --mode batch > $O/write_$CFG.json 2> $O/write_$CFG.err
  python3 tools/profile_summarize.py traffic $O/fetch_$CFG $O/write_$CFG r03 $CFG
  cp profiles/traffic_${CFG}_*.json $O/ 2>/dev/null
done
for CFG in eagle_catch push_slide; do
  bash tools/run_pmc_sq.sh $CFG > $O/pmc_sq_$CFG.log 2>&1
  cp gpurun_out/pmc_sq/sq_summary.csv $O/pmc_sq_$CFG.csv
  tail -5 $O/pmc_sq_$CFG.log
done
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
find gpurun_out/pmc_sq -name "*.csv" -size +1M -delete 2>/dev/null
du -sh $O
