# GPU box: A/B of the backward forms (3 = r01 matrix-core form, 4 = padded tiles): phase timings, then parity tests + bench
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
for v in 3 4; do for c in displacement eagle_catch push_slide; do echo -n "EMPC_BACKWARD=$v $c: "; EMPC_BACKWARD=$v timeout 300 python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep "^{"; done; done
bash tools/gpu_quick.sh
