#!/bin/bash
# CPU side: one gpurun call of tools/gpu_r5.sh with the commit of the working tree stamped into the environment of the box.
#   bash tools/gpurun_r5.sh <mode> <tag> [timeout seconds] [VAR=value ...]
# The call's console goes to gpurun_out/<tag>_call.log; a tree with uncommitted changes is marked "<sha>+dirty".
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
MODE="${1:-check}"; TAG="${2:-r05}"; TMO="${3:-2700}"; shift $(( $# < 3 ? $# : 3 ))
SHA=$(git rev-parse --short HEAD)
[ -n "$(git status --porcelain -- eagle-mpc_amd include bench.py tests tools oracle | head -1)" ] && SHA="${SHA}+dirty"
mkdir -p gpurun_out
# DRY_RUN=1: print the command line gpurun would get and stop (tests/test_bench_launch.py checks it for 0, 2 and 4+ arguments)
if [ -n "${DRY_RUN:-}" ]; then echo "gpurun --timeout $TMO -- EMPC_COMMIT=$SHA $* bash tools/gpu_r5.sh $MODE $TAG"; exit 0; fi
/usr/local/graft/bin/gpurun --timeout "$TMO" -- "EMPC_COMMIT=$SHA $* bash tools/gpu_r5.sh $MODE $TAG" > "gpurun_out/${TAG}_call.log" 2>&1
rc=$?
tail -${TAIL:-60} "gpurun_out/${TAG}_call.log"
exit $rc
