# GPU box: the profiles kept under profiles/ for one workload (default: the bench default, eagle_catch B=1024):
#   bench line, rocprofv3 --kernel-trace --stats summary of the same command, HBM traffic from separate --pmc passes.
# usage: bash tools/run_profiles.sh [config] [tag]      (outputs under gpurun_out/prof_<config>/ and profiles/traffic_*)
set -euo pipefail
set -x
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
CFG=${1:-eagle_catch}
TAG=${2:-r02}
O="$ROOT/gpurun_out/prof_$CFG"
rm -rf "$O"; mkdir -p "$O"
python3 bench.py --config $CFG --steps 5 --warmup 1 --no-secondary > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/write.json 2> $O/write.err
python3 tools/profile_summarize.py stats $O/stats $O/kernel_stats.csv
python3 tools/profile_summarize.py traffic $O/fetch $O/write $TAG $CFG
cp profiles/traffic_${CFG}_*.json $O/
# keep only the small summaries
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
du -sh $O
