# SQ counters of the three hot kernels in isolation (phase-level launches on a full batch); usage: bash tools/run_pmc_sq.sh [config]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
CFG=${1:-displacement}
O="$ROOT/gpurun_out/pmc_sq"
rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/p1 -- python3 tools/phase_bench.py --config $CFG --reps 2 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $O/p2 -- python3 tools/phase_bench.py --config $CFG --reps 2 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_BRANCH -d $O/p3 -- python3 tools/phase_bench.py --config $CFG --reps 2 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/p4 -- python3 tools/phase_bench.py --config $CFG --reps 2 > $O/p4.log 2>&1
tail -3 $O/p1.log
python3 tools/profile_summarize.py sq $O/sq_summary.csv $O/p1 $O/p2 $O/p3 $O/p4
find $O -name "*.csv" -size +1M -delete
