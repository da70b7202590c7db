# GPU box: GPU tests + the two bench lines (no CPU baseline), compact
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
timeout 1500 python -m pytest tests -q -m gpu ${PYTEST_ARGS:-} 2>&1 | tail -6
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:50], '| value %.1f | ms/step %.1f | sweeps %.0f | kernel ms/solve %s | secondary %s'%(d['value'],d['ms_per_step'],d['sweeps_per_solve'],{k:round(v,1) for k,v in d['kernel_ms_per_solve'].items()}, d.get('secondary',{}).get('value')), d.get('secondary',{}).get('kernel_ms_per_launch'))"
