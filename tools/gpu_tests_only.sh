# GPU box: the GPU test suite only (no -x: every failure is reported)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu --durations=8 ${@:-} 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
