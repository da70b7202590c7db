# GPU box: bench line of every BASELINE config that fits one GPU (usage: bash tools/gpu_configs.sh [configs...])
set -uo pipefail  # no -e: every step reports, a failing step does not hide the later ones
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
for c in ${@:-displacement eagle_catch push_slide hover}; do python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], '| value %.1f it/s | ms/step %.1f | sweeps %.0f | mean iters %.1f | kernel ms %s'%(d['value'],d['ms_per_step'],d['sweeps_per_solve'],d['mean_iters_per_trajectory'],{k:round(v,1) for k,v in d['kernel_ms_per_solve'].items()}))"; done
