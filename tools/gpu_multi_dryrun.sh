# GPU box (1 GPU): the N > 1 path of bench.py started exactly as the driver starts it -- `python3 bench.py --gpus 2` --
# with both ranks sharing the one device (gloo for the gather: RCCL refuses two ranks on one GPU).
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
python3 bench.py --gpus 2 --steps 1 --warmup 1 --batch 256 --backend gloo 2>&1 | tail -2 | cut -c1-600
echo "exit code: ${PIPESTATUS[0]}"
