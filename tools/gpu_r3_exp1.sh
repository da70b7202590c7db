# GPU box: event-sampling and chunked-stream experiments on the stream-mode bench
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
run() { echo "== $*"; env "$@" python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print({k: j[k] for k in ('value', 'ms_per_step', 'sweeps_per_step', 'ms_per_sweep', 'other_kernels_avg_ms')}, {k: round(v['avg_ms'], 4) for k, v in j['kernels'].items()}, 'single', round(j['single_batch']['value'], 1))
"; }
run EMPC_TIMING_EVERY=1
run EMPC_TIMING_EVERY=4
run EMPC_TIMING_EVERY=1000000
run EMPC_TIMING_EVERY=4 EMPC_STREAMS=2
run EMPC_TIMING_EVERY=4 EMPC_STREAMS=3
run EMPC_TIMING_EVERY=4 EMPC_STREAMS=4
