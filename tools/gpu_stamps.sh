# GPU box: phase timings + in-kernel cycle stamps from the diagnostic library (`make -C eagle-mpc_amd stamps`, built on the
# host and shipped with the snapshot); usage: bash tools/gpu_stamps.sh [config ...]
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
test -f eagle-mpc_amd/libempc_stamps.so || make -C eagle-mpc_amd -j16 -s stamps
for c in ${@:-displacement eagle_catch}; do EMPC_LIB_PATH="$ROOT/eagle-mpc_amd/libempc_stamps.so" python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep -E "^\{|stage|role"; done
