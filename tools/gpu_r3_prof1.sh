# GPU box, round 3: where the cycles go now -- in-kernel stamps of the three hot kernels (both configs), and the kernel-trace
# summary of the default bench (stream mode) so that select / calc / gaps show up next to the hot three.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r3prof1
bash tools/gpu_stamps.sh displacement eagle_catch 2>&1 | tee gpurun_out/r3prof1/stamps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof1/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3prof1/bench_under_rocprof.json 2> gpurun_out/r3prof1/stats.err
python3 tools/profile_summarize.py stats gpurun_out/r3prof1/stats gpurun_out/r3prof1/kernel_stats.csv
cat gpurun_out/r3prof1/kernel_stats.csv
find gpurun_out/r3prof1 -name "*.csv" -size +2M -delete
find gpurun_out/r3prof1 -name "*.db" -delete
