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
# What travels: the object directories of the variant libraries never (32 MB each); the variant libraries themselves (32 MB each,
# eagle-mpc_amd/libempc_<tag>.so) only with the modes that run them.  .gpurunignore is rewritten for the call and restored after it.
IGN=.gpurunignore; cp "$IGN" "$IGN.keep" 2>/dev/null || : > "$IGN.keep"
{
  echo "# written by tools/gpurun_r5.sh for one call (the committed list is restored afterwards)"
  for d in eagle-mpc_amd/build_*/; do [ -d "$d" ] && echo "$d"; done
  for f in tests/csrc/liblane_emulator_*.so; do [ -f "$f" ] && echo "$f"; done
  case "$MODE" in
    variants|stamps|ab|abgap|slots|final) ;;
    *) for f in eagle-mpc_amd/libempc_*.so; do [ -f "$f" ] && echo "$f"; done ;;
  esac
} > "$IGN"
trap 'mv -f "$IGN.keep" "$IGN"' EXIT
# DRY_RUN=1: print the command line gpurun would get and stop (tests/test_bench_launch.py checks it for 0, 2 and 4+ arguments)
if [ -n "${DRY_RUN:-}" ]; then echo "gpurun --timeout $TMO -- EMPC_COMMIT=$SHA $* bash tools/gpu_r5.sh $MODE $TAG"; exit 0; fi
/usr/local/graft/bin/gpurun --timeout "$TMO" -- "EMPC_COMMIT=$SHA $* bash tools/gpu_r5.sh $MODE $TAG" > "gpurun_out/${TAG}_call.log" 2>&1
rc=$?
tail -${TAIL:-60} "gpurun_out/${TAG}_call.log"
exit $rc
