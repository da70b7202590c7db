#!/bin/bash
# CPU side, round 6: the hardware programme in SEPARATE gpurun calls, most valuable first, each keeping its results under profiles/
# before the next starts -- a pool that closes half way leaves everything measured so far.
#   bash tools/gpurun_r6.sh [phase ...]        phases (default: all, in this order):
#     check         whole GPU suite on the shipped library + the driver's default bench line            (~25 min)
#     profiles      rocprofv3 kernel stats over the driver's command + four PMC passes, three workloads (~35 min)
#     experimental  the opt-in problem classes (small classes, mixed (6,6), two contacts per stage), each file in a pytest process of its own (~20 min)
#     probes        v_mfma_f64_4x4x4 layout + cost (decides EMPC_BWD_MFMA4), pipeline latencies         (~3 min)
#     variants      every libempc_<tag>.so: parity core of the suite + bench lines against the shipped  (~15 min per library)
#     lines         bench lines of the other BASELINE configurations
# A refused call (pool closed) ends the programme at once: no polling.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
PHASES="${*:-check profiles experimental probes variants stamps lines}"
declare -A TMO=([check]=2400 [profiles]=3000 [experimental]=2600 [probes]=400 [variants]=5400 [lines]=1500 [stamps]=900)
for ph in $PHASES; do
  tag="r06_${ph}"
  echo "=== phase $ph (tag $tag, limit ${TMO[$ph]} s)"
  # the variants phase: the four combinations first (one hour); the single switches (attribution) only when asked: VARIANTS=... in the environment
  extra=""; [ "$ph" = variants ] && extra="VARIANTS=${VARIANTS:-bwd,bwdm4,all,alltri}"
  bash tools/gpurun_r5.sh "$ph" "$tag" "${TMO[$ph]}" $extra; rc=$?
  if grep -q "status=refused" "gpurun_out/${tag}_call.log" 2>/dev/null; then
    echo "pool closed (gpurun_out/${tag}_call.log): stopping"; exit 2
  fi
  bash tools/keep_r06.sh "$tag" "$ph" || true
  echo "phase $ph rc $rc"
done
