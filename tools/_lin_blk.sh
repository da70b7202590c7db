cd $GRAFT_REPO_ROOT
EMPC_DEBUG_OCC=1 python3 tools/phase_bench.py --reps 3 2>&1 | grep -v "^backward" | tail -3
