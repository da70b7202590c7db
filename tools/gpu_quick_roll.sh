# GPU box: quick look at the rollout forms: phase timings + the rollout-related parity tests
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
for v in ${ROLL_VERSIONS:-6}; do for c in displacement eagle_catch; do echo -n "EMPC_ROLLOUT=$v $c: "; EMPC_ROLLOUT=$v timeout 300 python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep "^{"; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_eagle_catch.py -q -m gpu -x 2>&1 | tail -5
EMPC_ROLLOUT=6 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:50], '| value %.1f | ms/step %.1f | sweeps %.0f | kernel ms/solve %s | secondary %s'%(d['value'],d['ms_per_step'],d['sweeps_per_solve'],{k:round(v,1) for k,v in d['kernel_ms_per_solve'].items()}, d.get('secondary',{}).get('value')), d.get('secondary',{}).get('kernel_ms_per_launch'))"
