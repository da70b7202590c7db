#!/bin/bash
# CPU side: the variant libraries tools/gpu_r5.sh variants A/Bs against the shipped one (eagle-mpc_amd/libempc_<tag>.so, switches of
# csrc/empc_variants.hpp; ~3 minutes each with 8 cores).  bash tools/build_variants.sh [tag ...]    JOBS=6 by default
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/../eagle-mpc_amd"
# every backward variant on the 16 x 16 x 4 form; bwdm4: the same on the 4 x 4 x 4 form
BWD="-DEMPC_BWD_R4B=1 -DEMPC_BWD_SYMTILES=1 -DEMPC_BWD_GLDS=1 -DEMPC_BOX_LDS=1 -DEMPC_ANY_BALLOT=1 -DEMPC_BWD_OVERLAP=1 -DEMPC_BWD_FUSE=1 -DEMPC_BWD_VPTR=1"
declare -A V=(
  [r4b]="-DEMPC_BWD_R4B=1"
  [sym]="-DEMPC_BWD_SYMTILES=1"
  [glds]="-DEMPC_BWD_GLDS=1"
  [boxlds]="-DEMPC_BOX_LDS=1"
  [mfma4]="-DEMPC_BWD_MFMA4=1"
  [overlap]="-DEMPC_BWD_OVERLAP=1"
  [fuse]="-DEMPC_BWD_FUSE=1"
  [vptr]="-DEMPC_BWD_VPTR=1"
  [rcap]="-DEMPC_ROLL_CAP_LDS=1"
  [tri]="-DEMPC_REC_TRI=1"
  [gap]="-DEMPC_ROLL_GAP_EARLY"
  [bits]="-DEMPC_FSQRT_BITS=1"
  [bwd]="$BWD"
  [bwdm4]="$BWD -DEMPC_BWD_MFMA4=1"
  [all]="$BWD -DEMPC_ROLL_CAP_LDS=1 -DEMPC_ROLL_GAP_EARLY"
  [alltri]="$BWD -DEMPC_ROLL_CAP_LDS=1 -DEMPC_ROLL_GAP_EARLY -DEMPC_REC_TRI=1"
  [ncap3]="-DEMPC_NCAP=3"   # three capture slots: problems whose stages name three distinct frames (not an A/B candidate: a capability)
  [stamps]="-DEMPC_STAMPS"
  [stamps_bwd]="-DEMPC_STAMPS $BWD"
)
TAGS="${*:-bwd bwdm4 all alltri r4b sym glds boxlds mfma4 overlap fuse vptr rcap tri gap bits stamps stamps_bwd}"
for t in $TAGS; do
  [ -n "${V[$t]:-}" ] || { echo "unknown variant $t"; exit 2; }
  echo "== $t: ${V[$t]}"
  make -j"${JOBS:-6}" BUILD=build_$t LIB=libempc_$t.so EXTRA="${V[$t]}" 2>&1 | grep -E "error|Error" | head -5
  ls -la libempc_$t.so | awk '{print $5, $6, $7, $8, $9}'
  # the objects are not needed once the library is linked (38 MB per variant that a snapshot of the tree would carry to a GPU box:
  # the committed .gpurunignore lists the libraries by name, not these directories); KEEP_OBJECTS=1 keeps them for inspection
  [ -n "${KEEP_OBJECTS:-}" ] || rm -rf "build_$t"
done
