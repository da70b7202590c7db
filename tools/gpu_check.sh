# GPU box: parity tests + phase timings + bench line (usage: gpurun -- 'bash tools/gpu_check.sh')
set -uo pipefail  # no -e: every step reports, a failing step does not hide the later ones
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 tools/phase_bench.py --reps 3 2>&1 | grep "^{"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1000
