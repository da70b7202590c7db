# GPU box: GPU tests, then the committed profiles of both workloads (eagle_catch = bench default, displacement = configs[1])
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 900 bash tools/run_profiles.sh eagle_catch r02 > gpurun_out/prof_eagle.log 2>&1; echo "profiles eagle rc $?"
timeout 900 bash tools/run_profiles.sh displacement r02 > gpurun_out/prof_disp.log 2>&1; echo "profiles displacement rc $?"
cat gpurun_out/prof_eagle_catch/kernel_stats.csv | cut -c1-110
cat gpurun_out/prof_displacement/kernel_stats.csv | cut -c1-110
cat profiles/traffic_eagle_catch_*.json profiles/traffic_displacement_*.json | grep -E "kernel|hbm_bytes|launches"
mkdir -p gpurun_out/profiles_new; cp profiles/traffic_*_*.json gpurun_out/profiles_new/
