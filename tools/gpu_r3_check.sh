# GPU box: round-3 check -- the new step-wise / stream / 2-rank tests first (fail fast), then the whole GPU suite, then the
# default bench line (stream mode + single_batch + secondary displacement + cpu baseline + parity block).
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|Socket|Core|Thread" | head -5
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_teacher_forced.py tests/test_gpu_multirank.py -q -m gpu --durations=12 2>&1 | tail -60 | tee gpurun_out/r3_pytest_new.log
timeout 1500 python -m pytest tests -q -m gpu --durations=8 --deselect tests/test_gpu_teacher_forced.py --deselect tests/test_gpu_stream.py --deselect tests/test_gpu_multirank.py 2>&1 | tail -25 | tee gpurun_out/r3_pytest_rest.log
timeout 900 python3 bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc $?"
tail -c 9000 gpurun_out/r3_bench.json
tail -3 gpurun_out/r3_bench.err
