# GPU box, round 3: bench lines of the other BASELINE configs (hover; the three MPC loops = configs[4] and its siblings) and the
# driver's own command line for the headline (--steps 20 --warmup 5).  usage: bash tools/gpu_configs_r3.sh
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
O=gpurun_out/r3cfg
mkdir -p $O
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_eagle_catch_steps20.json
for c in hover carrot_mpc rail_mpc weighted_mpc; do
  timeout 600 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_$c.json
done
for f in $O/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read())
print('$f', '| value %.1f | ms/step %.2f | mode %s | sweeps/step %s' % (d['value'], d['ms_per_step'], d['config'].get('mode'), d.get('sweeps_per_step')))"; done
