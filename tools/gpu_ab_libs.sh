# GPU box: A/B of kernel variants built as separate libraries: phase timings (ms per launch of the three hot kernels on a full
# batch) per library and config.  usage: bash tools/gpu_ab_libs.sh "libA.so libB.so ..." "config ..."
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
for lib in ${1:-libempc.so}; do for c in ${2:-displacement eagle_catch}; do
  echo -n "$lib $c: "; EMPC_LIB_PATH="$ROOT/eagle-mpc_amd/$lib" timeout 300 python3 tools/phase_bench.py --config $c --reps 5 2>&1 | grep "^{" | head -1
done; done
