# GPU box: diagnostic build with in-kernel cycle stamps (-DEMPC_STAMPS), phase timings of one config; restores nothing
# (the box is a scratch copy)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
rm -rf eagle-mpc_amd/build/csrc/empc_inst_4_6*.o
make -C eagle-mpc_amd -j16 -s EXTRA=-DEMPC_STAMPS 2>&1 | grep -E "error" | head
for c in ${@:-displacement eagle_catch}; do python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep -E "^\{|rollout6|backward stage"; done
