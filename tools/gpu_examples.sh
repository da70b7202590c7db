# GPU box: run the examples and the bench line with its parity block
set -uo pipefail  # no -e: every step reports, a failing step does not hide the later ones
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
python3 examples/python/trajectory.py 2>&1 | tail -3
python3 examples/python/mpc.py 2>&1 | tail -2
./examples/cpp/trajectory . 2>&1 | tail -2
./examples/cpp/mpc . 2>&1 | tail -1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python3 bench.py --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value %.1f'%d['value']); print('parity',d['parity']); print('cpu_baseline',d['cpu_baseline']['value'],d['cpu_baseline']['cores'])"
