set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
run() { echo "== $*"; env "$@" python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print({k: j[k] for k in ('value', 'ms_per_step', 'ms_per_sweep')}, {k: round(v['avg_ms'], 4) for k, v in j['kernels'].items()}, 'secondary', round(j['secondary']['value'], 1), j['secondary']['kernel_ms_per_launch'])
"; }
run EMPC_LIN_MERGED=1
run EMPC_LIN_MERGED=2
run EMPC_LIN_MERGED=0
