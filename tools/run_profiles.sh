set -x
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
O="$ROOT/gpurun_out/prof_final"
rm -rf "$O"; mkdir -p "$O"
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.json 2> $O/write.err
python3 tools/profile_summarize.py stats $O/stats $O/kernel_stats.csv
python3 tools/profile_summarize.py traffic $O/fetch $O/write r01
cp profiles/traffic_*.json $O/
# keep only the small summaries
find $O -name "*.csv" -size +2M -delete
du -sh $O
