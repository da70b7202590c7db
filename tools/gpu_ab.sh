# GPU box: A/B of the backward forms (vector vs matrix-core)
set -uo pipefail  # no -e: every step reports, a failing step does not hide the later ones
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
for v in 2 3; do echo -n "EMPC_BACKWARD=$v: "; EMPC_BACKWARD=$v python3 tools/phase_bench.py --reps 3 2>&1 | grep "^{"; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140
