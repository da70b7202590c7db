cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_ic
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d $O/p1 -- python3 tools/phase_bench.py --reps 2 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $O/p2 -- python3 tools/phase_bench.py --reps 2 > $O/p2.log 2>&1
tail -2 $O/p1.log
python3 tools/profile_summarize.py sq $O/ic_summary.csv $O/p1 $O/p2
find $O -name "*.csv" -size +1M -delete
