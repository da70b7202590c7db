# GPU box: round-2 check -- all GPU tests, the default bench line (eagle_catch + secondary displacement), 2-rank dry run
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|Socket|Core|Thread" | head -5
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/r2_pytest.log
timeout 600 python3 bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc $?"
tail -c 6000 gpurun_out/r2_bench.json
tail -3 gpurun_out/r2_bench.err
timeout 600 bash tools/gpu_multi_dryrun.sh 2>&1 | tail -4 | tee gpurun_out/r2_multi.log
