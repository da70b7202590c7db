# GPU box: packed role-split rollout (EMPC_ROLLOUT=6, default) against the wave-per-trajectory form (5): phase timings,
# test suite, bench lines
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
for v in 5 6; do for c in displacement eagle_catch; do echo -n "EMPC_ROLLOUT=$v $c: "; EMPC_ROLLOUT=$v timeout 300 python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep "^{"; done; done
timeout 1700 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
for v in 5 6; do EMPC_ROLLOUT=$v timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ROLLOUT=$v', d['config']['workload'][:50], '| value %.1f | ms/step %.1f | sweeps %.0f | kernel ms/solve %s | secondary %s'%(d['value'],d['ms_per_step'],d['sweeps_per_solve'],{k:round(v,1) for k,v in d['kernel_ms_per_solve'].items()}, d.get('secondary',{}).get('value')), d.get('secondary',{}).get('kernel_ms_per_launch'))"; done
