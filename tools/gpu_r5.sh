# GPU box, round 5: ONE script for everything this round measures (grown from round 4's).  Usage (through gpurun, from the repo root):
#   tools/gpu_r5.sh check        whole GPU suite, then the default bench line
#   tools/gpu_r5.sh ab           bench lines with the baked-robot kernels on (default) and off (EMPC_BAKED=0), eagle_catch +
#                                displacement + push_slide, no CPU baseline (kernel comparison only)
#   tools/gpu_r5.sh tests        GPU suite only
#   tools/gpu_r5.sh profiles     rocprofv3 kernel stats + PMC passes behind profiles/r06_*
#   tools/gpu_r5.sh final        check + profiles + lines (then: bash tools/keep_r05.sh <tag> copies the summaries to profiles/)
#   tools/gpu_r5.sh lines        bench lines of hover and the three closed-loop MPC configurations
#   tools/gpu_r5.sh slots        occupancy experiment (slots in flight x build variants)
#   tools/gpu_r5.sh stamps       in-kernel cycle stamps (libempc_stamps.so)
#   tools/gpu_r5.sh experimental GPU tests of the opt-in problem classes (never run on hardware), in their own pytest process
#   tools/gpu_r5.sh probes       tools/probes/*: layout + cost of v_mfma_f64_4x4x4 (not used by the product yet), pipeline latencies
#   tools/gpu_r5.sh variants     prepared variant libraries (libempc_<tag>.so): parity core of the suite + bench lines, each against the shipped one
# EMPC_COMMIT (exported by tools/gpurun_r5.sh) names the commit of the snapshot: it is written into every summary this script leaves.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
MODE="${1:-check}"
TAG="${2:-r05}"
nproc; lscpu | grep -E "Model name" | head -2
echo "commit ${EMPC_COMMIT:-unknown} device code $(python3 tools/device_code_id.py)"

bench_line() {  # name, env, args...
  local name="$1"; shift
  local envs="$1"; shift
  env $envs timeout 900 python3 bench.py "$@" > "gpurun_out/${TAG}_bench_${name}.json" 2> "gpurun_out/${TAG}_bench_${name}.err"
  echo "bench ${name} rc $?"
  python3 - "gpurun_out/${TAG}_bench_${name}.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  no bench line:", e); sys.exit(0)
k = d.get("kernels_ms_per_launch") or d.get("kernel_ms") or {}
print("  value %.1f %s ms/step %.2f" % (d["value"], d["unit"], d["ms_per_step"]), "single_batch", (d.get("single_batch") or {}).get("value"),
      "secondary", (d.get("secondary") or {}).get("value"))
for key in ("roofline", "kernels"):
    if key in d: print("  ", key, json.dumps(d[key])[:600])
PY
}

case "$MODE" in
  tests)
    timeout 1700 python -m pytest tests -q -m gpu --durations=10 ${PYTEST_ARGS:-} 2>&1 | tail -${TAIL:-60} | tee "gpurun_out/${TAG}_pytest.log"
    ;;
  check)
    timeout 1700 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -40 | tee "gpurun_out/${TAG}_pytest.log"
    bench_line default "EMPC_X=0"
    tail -c 6000 "gpurun_out/${TAG}_bench_default.json"
    ;;
  ab)
    for cfg in eagle_catch displacement push_slide; do
      bench_line "${cfg}_baked" "EMPC_BAKED=1" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
      bench_line "${cfg}_generic" "EMPC_BAKED=0" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
    done
    ;;
  profiles)
    # what is kept under profiles/r06_*: the bench line, the rocprofv3 kernel-trace summary of the same command and four
    # --pmc passes (HBM reads, HBM writes, FP64 instruction mix, wavefront activity) over a short stream run
    export TMPDIR=/tmp
    O="$ROOT/gpurun_out/${TAG}_prof"; rm -rf "$O"; mkdir -p "$O"
    for CFG in ${CONFIGS:-eagle_catch displacement push_slide}; do
      T=$(python3 -c "import bench, empc_loader as l; e = l.load(); r, dt = bench.CONFIGS['$CFG']; t = e.Trajectory(); t.autoSetup(e.yaml_path(r)); print(t.createProblem(dt, True, 'IntegratedActionModelEuler').T)")
      STEPS=20; [ "$CFG" = push_slide ] && STEPS=5
      timeout 900 python3 bench.py --config $CFG --gpus 1 --steps $STEPS --warmup 5 --no-secondary > $O/bench_$CFG.json 2> $O/bench_$CFG.err; echo "bench $CFG rc $?"
      tail -c 400 $O/bench_$CFG.json; echo
      # the kernel-trace summary is taken over the DRIVER'S command (python3 bench.py --gpus 1 --steps 20 --warmup 5) so that its
      # per-kernel averages reproduce the bench line's roofline.frac; the counter passes use a shorter queue
      ARGS="--config $CFG --gpus 1 --steps $STEPS --warmup 5 --no-cpu-baseline --no-secondary --no-single-batch --no-slots-sweep"
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -- python3 bench.py $ARGS > $O/bench_under_rocprof_$CFG.json 2> $O/stats_$CFG.err
      python3 tools/profile_summarize.py stats $O/stats_$CFG $O/kernel_stats_$CFG.csv | head -12
      PARGS="--config $CFG --steps 4 --warmup 0 --no-cpu-baseline --no-secondary --no-single-batch --no-slots-sweep"
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch_$CFG -- python3 bench.py $PARGS > /dev/null 2> $O/p_fetch_$CFG.err
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write_$CFG -- python3 bench.py $PARGS > /dev/null 2> $O/p_write_$CFG.err
      rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/p_fp64_$CFG -- python3 bench.py $PARGS > /dev/null 2> $O/p_fp64_$CFG.err
      rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $O/p_act_$CFG -- python3 bench.py $PARGS > /dev/null 2> $O/p_act_$CFG.err
      python3 tools/profile_summarize.py pmc $CFG $O/pmc_$CFG.json 1024 $T 10 $O/p_fetch_$CFG $O/p_write_$CFG $O/p_fp64_$CFG $O/p_act_$CFG | cut -c1-1500
    done
    find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.csv" -size +6M -delete; find $O -name "*.db" -delete; du -sh $O
    ;;
  slots)
    # occupancy experiment: slots in flight per GPU x {shipped kernels (one wavefront per SIMD for the chain
    # kernels), libempc_w2.so = backward and rollout compiled for two (256 registers, spills)}; trajectory-iterations/s / 1024
    # (the two-wavefront variant is not kept built: make -C eagle-mpc_amd BUILD=build_w2 LIB=libempc_w2.so EXTRA="-DEMPC_ROLL_WAVES=2 -DEMPC_BWD_WAVES=2",
    #  then SLOT_LIBS="default $PWD/eagle-mpc_amd/libempc_w2.so")
    for LIBV in ${SLOT_LIBS:-default}; do
      [ "$LIBV" = default ] && LIBV=""
      for CFG in ${CONFIGS:-eagle_catch displacement}; do
        for B in 1024 2048 4096; do
          ST=${SLOT_STEPS:-20}  # the same number of steps (queue = steps x slots) for every size: equal share of drain sweeps
          EMPC_LIB_PATH="$LIBV" timeout 900 python3 bench.py --config $CFG --batch $B --steps $ST --warmup 1 --no-cpu-baseline --no-secondary --no-single-batch --no-slots-sweep > gpurun_out/${TAG}_slots.json 2> gpurun_out/${TAG}_slots.err
          python3 - "gpurun_out/${TAG}_slots.json" "$CFG" "$B" "${LIBV:-shipped}" <<'PY' | tee -a "gpurun_out/${TAG}_slots_sweep.jsonl"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels"]
    print(json.dumps({"config": sys.argv[2], "slots": int(sys.argv[3]), "library": sys.argv[4].split("/")[-1], "iters_per_s_per_1024": d["trajectory_iters_per_s"] / 1024.0,
                      "ms_per_sweep": d["ms_per_sweep"], "ms": {n: k[n]["avg_ms"] for n in k}}))
except Exception as e:
    print(json.dumps({"config": sys.argv[2], "slots": int(sys.argv[3]), "library": sys.argv[4].split("/")[-1], "error": str(e)}))
PY
        done
      done
    done
    ;;
  lines)
    # the bench lines of the other configurations (profiles/r05_bench_<config>.json)
    for cfg in ${CONFIGS:-hover carrot_mpc rail_mpc weighted_mpc}; do
      bench_line "$cfg" "EMPC_X=0" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep
    done
    ;;
  overlap)
    # chunks of the batch on their own streams x the CU-exclusive backward launch (four trajectories per workgroup): does the
    # rollout of one chunk (171 of 256 CUs at a full batch) run beside the backward pass of another?
    for CFG in ${CONFIGS:-eagle_catch displacement}; do
      for V in ${VARIANTS:-1:1 4:1 1:2 4:2 4:3 4:4}; do
        W=${V%%:*}; NS=${V##*:}
        bench_line "${CFG}_wpb${W}_streams${NS}" "EMPC_BWD_WPB=$W EMPC_STREAMS=$NS" --config $CFG --no-cpu-baseline --no-secondary --no-slots-sweep --no-single-batch --steps ${OV_STEPS:-10}
      done
    done
    ;;
  final)
    # everything the committed round-4 artefacts come from, in one call: GPU suite, default bench line, profiles, other lines
    bash "$0" check "$TAG"
    bash "$0" profiles "$TAG"
    bash "$0" lines "$TAG"
    bash "$0" variants "$TAG"
    bash "$0" experimental "$TAG"
    bash "$0" probes "$TAG"
    bash "$0" stamps "$TAG"
    ;;
  abgap)
    # the prepared rollout experiment (LABNOTES.md section 6.1): shipped library against libempc_gap.so (-DEMPC_ROLL_GAP_EARLY)
    if [ -f "$ROOT/eagle-mpc_amd/libempc_gap.so" ]; then
      for cfg in eagle_catch displacement push_slide; do
        bench_line "${cfg}_shipped" "EMPC_X=0" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
        bench_line "${cfg}_gap_early" "EMPC_LIB_PATH=$ROOT/eagle-mpc_amd/libempc_gap.so" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
      done
    fi
    ;;
  variants)
    # every prepared variant library (eagle-mpc_amd/libempc_<tag>.so, empc_variants.hpp switches; built on the CPU side, shipped
    # with the snapshot): the parity core of the GPU suite through that library, then bench lines against the shipped one.
    # VARIANTS="gap r4b ..." restricts the list; QUICK_TESTS is the parity core (phase parity, step-wise parity, golden vectors,
    # failure exits, box solvers, stream bitwise)
    QT="${QUICK_TESTS:-tests/test_gpu_parity.py tests/test_gpu_teacher_forced.py tests/test_gpu_eagle_catch.py tests/test_gpu_branches.py tests/test_gpu_box_solvers.py tests/test_gpu_stream.py tests/test_gpu_baked.py tests/test_gpu_contact_arm5.py tests/test_gpu_contact_options.py tests/test_gpu_all_problems.py}"
    for cfg in ${CONFIGS:-eagle_catch displacement push_slide}; do
      bench_line "${cfg}_shipped" "EMPC_X=0" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
    done
    for lib in $ROOT/eagle-mpc_amd/libempc_*.so; do
      v=$(basename $lib .so); v=${v#libempc_}
      case "$v" in stamps*) continue;; esac
      if [ -n "${VARIANTS:-}" ] && ! echo " ${VARIANTS//,/ } " | grep -q " $v "; then continue; fi  # (VARIANTS=bwd,all: a comma list survives the gpurun command line)
      echo "=== variant $v ($(python3 tools/device_code_id.py $lib))"
      EMPC_LIB_PATH=$lib timeout 1500 python -m pytest $QT -q -m gpu -x 2>&1 | tail -6 | tee "gpurun_out/${TAG}_pytest_${v}.log"
      for cfg in ${CONFIGS:-eagle_catch displacement push_slide}; do
        bench_line "${cfg}_${v}" "EMPC_LIB_PATH=$lib" --config $cfg --no-cpu-baseline --no-secondary --no-slots-sweep --steps 10
      done
    done
    ;;
  experimental)
    # GPU tests of kernels that have never run on hardware (opt-in problem classes), in a pytest process of their own: a fault
    # here must not take the suite with it
    EMPC_RUN_EXPERIMENTAL_GPU_TESTS=1 timeout 1200 python -m pytest tests/test_zz_gpu_contact_small_classes.py -q -m gpu --durations=5 2>&1 | tail -30 | tee "gpurun_out/${TAG}_pytest_experimental.log"
    # (round 6: two contacts per stage, CT_PAIR3 -- again its own process)
    EMPC_RUN_EXPERIMENTAL_GPU_TESTS=1 timeout 1200 python -m pytest tests/test_zz_gpu_two_contacts.py -q -m gpu --durations=5 2>&1 | tail -30 | tee "gpurun_out/${TAG}_pytest_two_contacts.log"
    ;;
  probes)
    # one-wavefront probes of instructions the product does not use yet (built on the CPU side, shipped with the snapshot)
    for pb in mfma_f64_4x4_probe latency_probe; do
      [ -x "$ROOT/tools/probes/$pb" ] && { echo "== $pb"; timeout 120 "$ROOT/tools/probes/$pb"; echo "rc $?"; }
    done 2>&1 | tee "gpurun_out/${TAG}_probes.log" | tail -${TAIL:-120}
    ;;
  stamps)
    # phase-level launches: product library first (ms per launch), then the diagnostic build with in-kernel cycle stamps
    for c in displacement eagle_catch; do
      echo "== $c (product)"; python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep -E "^\{"
      for sl in $ROOT/eagle-mpc_amd/libempc_stamps*.so; do  # (libempc_stamps_<variant>.so: the same stamps inside a variant's kernels)
        echo "== $c (stamps build $(basename $sl))"; EMPC_LIB_PATH="$sl" python3 tools/phase_bench.py --config $c --reps 3 2>&1 | grep -E "^\{|stage|role"
      done
    done 2>&1 | tee "gpurun_out/${TAG}_stamps.log"
    ;;
  *)
    echo "unknown mode $MODE"; exit 2
    ;;
esac
