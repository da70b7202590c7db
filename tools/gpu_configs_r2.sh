# GPU box: bench lines of the other BASELINE configs (push_slide = configs[3]; the three MPC loops = configs[4] and its siblings)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
for c in push_slide hover carrot_mpc rail_mpc weighted_mpc; do
  timeout 600 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_$c.json
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_$c.json').read())
print('$c', '| value %.1f %s | ms/step %.1f | kernel ms %s' % (d['value'], d['unit'][:40], d['ms_per_step'], {k: round(v, 2) for k, v in d.get('kernel_ms_per_solve', {}).items()}))"
done
