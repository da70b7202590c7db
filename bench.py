#!/usr/bin/env python3
"""bench.py -- DDP iterations/s of the batched Squash-box FDDP solver on MI355X.

Default workload (the north-star target of BASELINE.json): hexacopter370_flying_arm_3 `eagle_catch.yaml` (contact
dynamics), dt = 32 ms -> 100 nodes, 1024 rollouts in flight per GPU from perturbed initial states (recipe of
benchmark/utils/utils.hpp:15-27), empty initial guess, SolverSbFDDP.solve(maxiter = 100) for each of them.

One "step" = 1024 solves per GPU.  Two ways of running K steps (`--mode`):
  stream (default)  the K x 1024 initial states of a rank form a queue that is pushed through the solver's 1024 slots
                    (empc_solver_stream_*, "continuous batching"): a slot whose rollout finishes hands its result row over
                    and takes the next initial state inside the same sweep, so every sweep works on a full batch.  The
                    queue and the result rows are resident in HBM; each rollout's arithmetic is bitwise that of a plain
                    solve (tests/test_gpu_stream.py).
  batch             K plain batched solves one after the other (the round-1/2 form): every solve ends with ~150 sweeps on
                    the few rollouts that need many iterations.  At N = 1 the stream run also reports this form
                    ("single_batch") next to the headline.
`--config displacement` is BASELINE.json configs[1]; at N = 1 a short run of it is appended under "secondary".  configs[2]
(eagle_catch, 4096 rollouts over 8 GPUs) is `python bench.py --gpus 8 --batch 512`.

metric/value: batched DDP iterations per second = (sum over rollouts of DDP iterations executed) / batch / time, i.e.
trajectory-iterations/s divided by the per-GPU batch; with N GPUs every rank works on its own queue (weak scaling) and
`value` is the whole-job aggregate.  The only collective is the gather of result rows to rank 0 (RCCL), inside the timed region.

`python bench.py --gpus N` starts itself: the parent spawns N worker processes (one per GPU, torch.distributed.run on
127.0.0.1) BEFORE it touches torch or the HIP library, relays rank 0's JSON line and exits with the workers' code.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode stream|batch]
                       [--config eagle_catch|displacement|hover|push_slide|*_mpc]
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CONFIGS = {
    "hover": ("hexacopter370/trajectories/hover.yaml", 40),
    "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
    "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
    "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13),
    # BASELINE.json configs[4]: closed-loop Carrot MPC on the displacement trajectory, 50-knot horizon, RK4 plants
    "carrot_mpc": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
    "rail_mpc": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
    "weighted_mpc": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
}
MPC_YAML = os.path.join(ROOT, "eagle-mpc_amd", "data", "mpc", "carrot_50knots.yaml")
MPC_CYCLES_PER_STEP = 20   # one bench step of the *_mpc configs = 20 controller cycles (updateProblem, solve, plant)
MPC_DT_SIM = 2             # ms, examples/python/mpc.py:41
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # same guide: measured copy bandwidth ("also report vs 6.29e12")
FP64_PEAK_TFLOPS = 78.6  # FP64 vector: 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz


def algorithmic_words(nx, ndx, nu):
    """FP64 words per (trajectory, knot, iteration) for each phase."""
    a_in = nx + nu
    a_out = ndx * ndx + ndx * nu + ndx + nu + ndx * (ndx + 1) // 2 + ndx * nu + nu * (nu + 1) // 2 + ndx + 1
    b_in = a_out - 1
    b_out = nu * ndx + nu + ndx
    c_in = nx + nu + nu * ndx + nu + 2 * ndx
    c_out = nx + nu
    return dict(linearize=a_in + a_out, backward=b_in + b_out, rollout=c_in + c_out,
                iteration=a_in + a_out + b_in + b_out + c_in + c_out)


def host_threads():
    """Threads for the all-core CPU baseline: the physical cores this process may run on (SMT siblings counted once,
    CPU affinity and the cgroup quota respected) -- more threads than that only slow the FP64 oracle down."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    n = len(cores) or len(allowed)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def secondary_displacement(empc, B, maxiter, device):
    """BASELINE.json configs[1] (the round-1 headline) measured in the same process: a stream of 3 x B rollouts (and the
    same rollouts as 3 plain batched solves), after one warm-up solve."""
    import torch
    rel, dt = CONFIGS["displacement"]
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(rel))
    problem = traj.createProblem(dt, True, "IntegratedActionModelEuler")
    d = problem.desc
    steps = 3
    x0s = empc.perturbed_x0s(problem.x0, B * steps, nq=d.model.nq)
    solver = empc.SolverSbFDDP(problem, batch=B, device=device)
    solver.solve([], [], maxiter, x0s=x0s[:B])
    solver.stream_begin(x0s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.stream_run(maxiter)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = solver.stats()
    t0 = time.perf_counter()
    agg = {}
    for k_ in range(steps):
        solver.solve([], [], maxiter, x0s=x0s[k_ * B:(k_ + 1) * B])
        for k, v in solver.stats().items():
            agg[k] = agg.get(k, 0) + v
    torch.cuda.synchronize()
    el_b = time.perf_counter() - t0
    return {"workload": "%s dt=%dms T=%d, %d rollouts through %d slots (BASELINE configs[1])" % (rel, dt, d.T, B * steps, B),
            "value": st["total_iters"] / B / el, "ms_per_step": el / steps * 1e3, "steps": steps,
            "sweeps_per_step": st["sweeps"] / steps,
            "kernel_ms_per_launch": {k: st["ms_" + k] / max(st["n_" + k], 1) for k in ("linearize", "backward", "rollout")},
            "single_batch": {"value": agg["total_iters"] / B / el_b, "ms_per_step": el_b / steps * 1e3,
                             "sweeps_per_solve": agg["sweeps"] / steps}}


def slots_sweep(empc, problem, d, args, device, B0, value0):
    """The occupancy view: the same stream with 2 x and 4 x the slots in flight on the one GPU, the same
    number of steps each (queue = steps x slots), throughput normalised to 1024 slots.  The headline stays at B0 slots; this
    says what the chain kernels leave on the table there: at 1024 slots the rollout occupies 171 of the 256 CUs and every chain
    kernel runs one wavefront per SIMD (profiles/r04_slots_sweep.md)."""
    import torch
    rows = [{"slots": B0, "value_per_%d_slots" % B0: value0, "relative": 1.0}]
    for mult in (2, 4):
        Bn = B0 * mult
        try:
            x0n = empc.perturbed_x0s(problem.x0, Bn * args.steps, nq=d.model.nq)
            sn = empc.SolverSbFDDP(problem, batch=Bn, device=device)
            sn.stream_begin(np.ascontiguousarray(x0n[:Bn]))
            sn.stream_run(args.maxiter)  # warm-up
            sn.stream_begin(x0n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sn.stream_run(args.maxiter)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = sn.stats()
            v = st["total_iters"] / B0 / el
            rows.append({"slots": Bn, "value_per_%d_slots" % B0: v, "relative": v / value0, "ms_per_sweep": el * 1e3 / max(st["sweeps"], 1),
                         "solves": Bn * args.steps})
            del sn
        except Exception as e:  # e.g. out of memory on a smaller part: the sweep is a side view, never the headline
            rows.append({"slots": Bn, "error": str(e)[:200]})
    return rows


def cpu_baseline_and_parity(empc, solver, problem, d, x0s, B, maxiter, unit):
    """Rank 0, N = 1, after the timed region: the oracle (CPU restatement, `kind: port`) on the host cores of this box as
    the reported CPU baseline, and -- on the same oracle results -- the parity block of the north star.  The oracle is the
    checker / baseline here, never part of what was measured above."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity_criteria as pc
    threads = host_threads()
    # all-core leg: a bounded sample of the batch (whole batch on a box with >= 64 cores), OpenMP over rollouts
    n_sample = int(min(B, max(64, threads * 16)))
    # Two builds of the same sources (oracle/Makefile): the CHECKER (contraction-free, -march=x86-64-v3: parity below runs
    # against it) and the TIMED one (-O3 -march=native, FMA on, compiled on this box: SURVEY.md section 8(d)) -- the baseline
    # numbers come from the timed build only; its rounding differs, so nothing is compared against it.
    ref = ob.solve_batch(d, x0s[:n_sample], maxiter, nthreads=threads, want_traj=True)
    try:
        timed = ob.solve_batch(d, x0s[:n_sample], maxiter, nthreads=threads, want_traj=False, variant="native")
        timed_build = "oracle/liboracle_native.so: g++ -O3 -march=native -ffp-contract=fast, built on this host"
        variant = "native"
    except Exception as e:  # no compiler on the box: fall back to the checker build and say so
        timed, variant = ref, None
        timed_build = "oracle/liboracle.so (checker build, -march=x86-64-v3 -ffp-contract=off): native build failed: %s" % str(e)[:120]
    cpu_iters = float((timed["iter"] + 1).sum())
    # single-thread leg (how the reference itself runs): three rollouts, one at a time, us per DDP iteration
    us_per_it, it_single, t_single = [], 0.0, 0.0
    for b in range(min(3, n_sample)):
        r1 = ob.solve_batch(d, x0s[b:b + 1], maxiter, nthreads=1, want_traj=False, variant=variant)
        n_it = float(r1["iter"][0] + 1)
        us_per_it.append(r1["seconds"] * 1e6 / n_it)
        it_single += n_it
        t_single += r1["seconds"]
    us_per_it = np.array(us_per_it)
    out = {"cpu_baseline": {
        "value": cpu_iters / B / timed["seconds"], "unit": unit, "cores": threads, "kind": "port",
        "sample": "%d of the %d rollouts of rank 0 (mean %.1f iterations each), OpenMP over rollouts on %d threads = physical "
                  "cores available to the process; FP64 C++ restatement of the Crocoddyl/Pinocchio arithmetic (oracle/), not the "
                  "reference binary: it cannot be built here" % (n_sample, B, cpu_iters / n_sample, threads),
        "build": timed_build,
        "checker_build_value": float((ref["iter"] + 1).sum()) / B / ref["seconds"],
        "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(),
        "trajectory_iters_per_s": cpu_iters / timed["seconds"], "seconds": timed["seconds"],
        "single_thread": {"trajectory_iters_per_s": it_single / t_single, "rollouts": int(len(us_per_it)),
                          "us_per_iteration": {"AVG": float(us_per_it.mean()), "STDDEV": float(us_per_it.std()),
                                               "MAX": float(us_per_it.max()), "MIN": float(us_per_it.min())},
                          "format": "benchmark/mpc-main-carrot-timings.cpp:42-55 (Avg. time per iteration)"}}}
    try:  # (the CPU baseline above is complete: a failure of the parity leg is reported in its own block)
        # parity.  (1) free-running batch against the oracle: statistics REPORTED, not judged -- the iteration paths of the contact
        # problem are rounding-sensitive (profiles/r02_oracle_sensitivity.json); asserted-on-every-run part: both sides solve the
        # same problem (oracle's cost and dynamics at the GPU's final points).  (2) the decisive, step-wise form on a sample
        # (tests/stepwise.py): every iteration of the oracle's paths reproduced by the GPU from the oracle's iterate, every
        # iteration of the GPU's paths reproduced by the oracle from the GPU's iterate, same minimiser from a common restart.
        import stepwise as sw
        solver.enable_trace(3 * maxiter + 20)
        solver.solve([], [], maxiter, x0s=x0s[:solver.batch])
        gpu = dict(xs=solver.xs_batch[:n_sample], us=solver.us_batch[:n_sample], cost=solver.cost_batch[:n_sample],
                   iter=solver.iter_batch[:n_sample], status=solver.status_batch[:n_sample])
        stats = pc.batch_statistics(gpu, ref)
        sample = sorted(set(int(i) for i in np.linspace(0, n_sample - 1, 8)))
        prm = empc.default_params()
        smp = pc.sample_checks(ob, d, x0s, gpu, {b: solver.trace(b) for b in sample}, sample,
                               final_smooth=prm.smooth_init * prm.smooth_mult, th_stop=prm.convergence_stop, maxiter=maxiter)
        solver.enable_trace(0)
        per_roll = np.maximum(np.abs(gpu["xs"] - ref["xs"]).reshape(n_sample, -1).max(axis=1),
                              np.abs(gpu["us"] - ref["us"]).reshape(n_sample, -1).max(axis=1))
        stepwise = None
        try:
            oprm = ob.default_params()
            rep = sw.stepwise_parity(lambda n, p2: sw.GpuBackend(empc, problem, p2 if p2 is not None else oprm, n), d, oprm,
                                     np.ascontiguousarray(x0s[sample[:6]]), maxiter=maxiter, tape_every=47)
            fr, sm = rep["free_run"], rep.get("same_minimum", {})
            stepwise = {"rollouts": rep["rollouts"], "iterations_teacher_forced": rep["pairs"],
                        "decisions_exact": rep["decisions_checked"] - rep.get("decisions_excused_chaotic", 0) - rep.get("direction_ties_excused", 0) - rep.get("decisions_excused_tied", 0),
                        "decisions_excused_blown_up_trial_or_tie": rep.get("decisions_excused_chaotic", 0) + rep.get("direction_ties_excused", 0) + rep.get("decisions_excused_tied", 0),
                        "trial_costs_checked": rep["trial_costs_checked"], "trial_costs_beyond_1e-9": rep.get("trial_costs_beyond_1e-9", 0),
                        "tapes_checked": rep["tapes_checked"], "iterates_skipped_exploded": rep.get("iterates_skipped_exploded", 0),
                        "max_rel": {k: rep["max_rel"].get(k) for k in ("cost", "tape_Fx", "tape_Lxx", "tape_Lx", "K", "k", "Vx", "cost_try_accepted")},
                        "gpu_iterations_reproduced_by_oracle": fr["oracle_reproduces_device_decision"],
                        "gpu_iterations": fr["device_iterations"], "unexplained": fr["unexplained"],
                        "free_paths_equal": fr["same_path_as_oracle"], "free_paths_diverging": fr["diverging"],
                        "same_minimum_xs_err_max": sm.get("xs_err_max"), "same_minimum_us_err_max": sm.get("us_err_max"),
                        "same_minimum_rollouts": sm.get("converged_on_oracle"), "passed": True}
        except Exception as e:  # the bench line reports it (whatever it is); tests/test_gpu_teacher_forced.py is where it fails a run
            stepwise = {"passed": False, "error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        # the contract of the north star on the unperturbed rollout, in the form smoke() asserts (tests/parity_criteria.py): same
        # minimiser from a common restart <= 1e-4, plain-solve cost within 1e-5 relative, identical iterations; plain xs / us reported
        # with a 2e-4 tripwire next to the oracle-vs-its-own-FMA-build yardstick
        try:
            contract = pc.north_star_contract(empc, ob, sw, problem, gpu["xs"][0], gpu["us"][0], float(gpu["cost"][0]), int(gpu["iter"][0]),
                                              x0=x0s[0], maxiter=maxiter)
        except Exception as e:  # the line reports it; smoke() and tests/test_gpu_eagle_catch.py are where it fails a run
            contract = {"passed": False, "failures": ["%s: %s" % (type(e).__name__, str(e)[:300])]}
        out["parity"] = {"reference": "oracle/liboracle.so (CPU restatement; parity unpinned, DESIGN.md)", "tolerance": 1e-4,
                         "rollouts_compared": n_sample,
                         "unperturbed_rollout_max_abs_err": float(per_roll[0]),
                         "unperturbed_rollout_iterations_equal": bool(gpu["iter"][0] == ref["iter"][0]),
                         "contract": contract,
                         "free_running_batch_statistics_reported_not_judged": stats, "sample_checks": smp,
                         "same_problem": bool(smp["oracle_cost_at_gpu_point_rel_err_max"] is not None and
                                              smp["oracle_cost_at_gpu_point_rel_err_max"] <= 1e-9 and
                                              smp["oracle_dynamics_defect_at_gpu_point_max"] <= 1e-8),
                         "stepwise": stepwise,
                         "method": "tests/stepwise.py: teacher-forced in both directions + same minimiser from a common restart; "
                                   "tests/test_gpu_teacher_forced.py runs it on 64 rollouts"}
    except Exception as e:
        out["parity"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
    return out


def golden_rank_check(empc, device, maxiter):
    """N > 1: every rank solves the unperturbed eagle_catch rollout on ITS GPU and compares it with the committed golden vector
    (tests/golden/solutions/eagle_catch.npz: data, readable on the GPU box) -- the north-star bound per rank."""
    path = os.path.join(ROOT, "tests", "golden", "solutions", "eagle_catch.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(str(g["yaml"])))
    problem = traj.createProblem(int(g["dt_ms"]), True, "IntegratedActionModelEuler")
    s = empc.SolverSbFDDP(problem, batch=1, device=device)
    s.solve([], [], maxiter, x0s=g["x0s"][:1])
    return bool(s.iter_batch[0] == g["iter"][0] and np.abs(s.xs_batch[0] - g["xs"][0]).max() < 1e-4 and
                np.abs(s.us_batch[0] - g["us"][0]).max() < 1e-4)


# what bounds each hot kernel; the measured shares (vector unit active, waiting) are appended from this round's PMC summary
# (profiles/r04_pmc_<config>.json: SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY over SQ_WAVE_CYCLES) when it exists for the workload
LIMITERS = {
    "linearize": "not HBM: register-bound wavefronts (two per SIMD at 256 VGPRs for the 9-dof arm, one at ~500 for the 11-dof arm; a "
                 "third / second one spills) walking ~10 LDS-synchronised stages per unit, of which the role phase (nominal chain, "
                 "Euler step, state differences: one lane per unit) is the longest",
    "backward": "not HBM: one wavefront per trajectory and SIMD walking the knots backwards (dependent chain); per knot 52 FP64 "
                "MFMAs (the gfx950 FP64 MFMA rate equals the vector rate), the redundant per-lane LLT + triangular solves, the "
                "symmetrisation and the gap terms, ~12 LDS hand-overs",
    "rollout": "not HBM: FP64 instruction issue of one wavefront per SIMD along the knot chain (a wave64 FP64 instruction "
               "occupies its SIMD for 4 cycles whatever the number of useful lanes); four role wavefronts per 6 trajectories, "
               "two LDS barriers per knot, 60 of 64 lanes busy, 171 of 256 CUs at 1024 slots",
}


def committed_counters(config, B, is_mpc, code_id):
    """The committed PMC summary of this workload (profiles/r*_pmc_<config>.json, newest round first) -- used only when it was
    taken on the device code that is running now (`device_code_id` inside the file == the id of the loaded library).
    -> (kernels dict, file name or None, note)"""
    import glob
    if B != 1024 or is_mpc:
        return {}, None, "no committed counter pass for this workload / batch"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % config)), reverse=True)
    seen = []
    for f in files:
        try:
            js = json.load(open(f))
        except Exception:
            continue
        if code_id is not None and js.get("device_code_id") == code_id:
            return js.get("kernels", {}), os.path.relpath(f, ROOT), "counters taken on this device code (id %s, commit %s)" % (code_id, js.get("commit"))
        seen.append("%s: %s" % (os.path.basename(f), js.get("device_code_id") or "no id (taken before round 5)"))
    return {}, None, "no committed counter pass matches the running device code (id %s); seen %s" % (code_id, seen or "none")


def limiter_text(kernel, pmc):
    c = (pmc or {}).get(kernel) or {}
    txt = LIMITERS[kernel]
    if c.get("valu_active_frac") is not None:
        txt += "; measured on full-batch launches: vector unit active %.0f %% of a wavefront's cycles, waiting %.0f %%, issue stalls %.0f %%" % (
            100 * c["valu_active_frac"], 100 * c.get("wait_frac", 0.0), 100 * c.get("issue_stall_frac", 0.0))
    return txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="rollouts in flight per GPU (default 1024; 256 for the *_mpc configs)")
    ap.add_argument("--config", default="eagle_catch", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="stream", choices=["stream", "batch"],
                    help="stream: the steps x batch initial states go through the batch slots as one queue; batch: one plain solve per step")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short displacement run appended at N = 1")
    ap.add_argument("--no-single-batch", action="store_true", help="skip the plain batched solves appended to a stream run at N = 1")
    ap.add_argument("--no-slots-sweep", action="store_true", help="skip the 2048- and 4096-slot stream runs appended at N = 1")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched workers (0 = pick a free one)")
    ap.add_argument("--maxiter", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the result gather (nccl = RCCL; gloo only for dry runs of the N > 1 path)")
    ap.add_argument("--dump-rows", default=None, help="rank 0 writes the gathered result rows of the timed region to this .npy file (tests)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and (world > 1 or "WORLD_SIZE" in os.environ):
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Self-launch: N fresh worker processes, one per GPU, started as CHILDREN before this process has imported torch
        # or loaded libempc.so (no exec: under rocprofv3 the parent already holds the GPU).  Rank 0 prints the JSON line
        # straight to the inherited stdout.
        import socket
        import subprocess
        port = args.master_port
        if not port:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import empc_loader
    empc = empc_loader.load()
    if empc.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the solver has no CPU path")
    n_dev = max(torch.cuda.device_count(), 1)
    local_dev = local_rank % n_dev  # one rank per GPU; the modulo only matters for gloo dry runs on fewer GPUs
    torch.cuda.set_device(local_dev)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    is_mpc = args.config.endswith("_mpc")
    stream = (args.mode == "stream") and not is_mpc
    if args.batch is None:
        args.batch = 256 if is_mpc else 1024
    rel, dt = CONFIGS[args.config]
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(rel))
    problem = traj.createProblem(dt, True, "IntegratedActionModelEuler")
    d = problem.desc
    B = args.batch
    import importlib
    sharding = importlib.import_module("eagle_mpc_amd.sharding")
    # The global job list: steps x B x world perturbed initial states (job 0 is the unperturbed YAML state); rank r owns the
    # contiguous shard [r * steps * B, (r + 1) * steps * B).  batch mode and the MPC configs re-solve one B-sized batch.
    n_steps_jobs = args.steps if stream else 1
    x0_all = empc.perturbed_x0s(problem.x0, B * world * n_steps_jobs, nq=d.model.nq)
    x0s = sharding.shard(x0_all, world, rank)  # (steps x B) x nx in stream mode, B x nx otherwise
    solver = empc.SolverSbFDDP(problem, batch=B, device=local_dev) if not is_mpc else None
    mpc_state = {"t": 0}
    if is_mpc:
        # plan once (one rollout), then B controllers-in-one track it from perturbed plant states
        planner = empc.SolverSbFDDP(problem, batch=1, device=local_dev)
        planner.solve([], [], args.maxiter)
        xs_plan, us_plan = np.array(planner.xs), np.array(planner.us)
        if args.config == "carrot_mpc":
            mpc = empc.CarrotMpc(traj, xs_plan, dt, MPC_YAML, batch=B, device=local_dev)
        elif args.config == "rail_mpc":
            mpc = empc.RailMpc(xs_plan, dt, MPC_YAML, batch=B, device=local_dev)
        else:
            mpc = empc.WeightedMpc(traj, dt, MPC_YAML, batch=B, device=local_dev)
        mpc.updateProblem(0)
        solver = mpc.solver
        d = mpc.problem.desc
        solver.plant_states = sharding.shard(empc.perturbed_x0s(xs_plan[0], B * world, nq=d.model.nq, amplitude=0.02), world, rank)
        solver.solve(xs_plan[:d.T + 1], us_plan[:d.T], args.maxiter, x0s="plant")
        solver.convergence_init = 1e-3

    rows_dev = {}
    gathered = {}
    seen_devices = {}

    def gather_device_ids():
        # which GPU each rank really computed on, through the same collective backend as the result rows, AFTER the closing
        # barrier (outside the timed interval): (rank, device index, PCI bus number, hash of the full bus id) of the device
        # libempc.so's solver memory lives on (empc_solver_device_info: hipPointerGetAttributes of its problem image), and
        # torch's own selection beside it.  Two ranks on one GPU, or a solver that fell back to another device than the one torch
        # (and so RCCL) uses, show up in the line (`rccl_ranks_seen`) instead of in a quietly wrong value.
        try:
            dev, busid = solver.device_info()
        except Exception:  # (CPU dry runs over gloo have no solver handle on a device)
            dev, busid = -1, ""
        tdev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        tbus = ""
        if torch.cuda.is_available():
            tp = torch.cuda.get_device_properties(tdev)
            if all(hasattr(tp, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):  # (not in every torch build)
                tbus = "%04x:%02x:%02x" % (tp.pci_domain_id, tp.pci_bus_id, tp.pci_device_id)
        # the self-check that can fail the run is the device ORDINAL (the solver's memory on torch's device); the PCI string of two
        # libraries is compared too, but only reported (1 same, 0 differs, -1 not available): its formatting has never been seen on
        # hardware, and a cosmetic difference must not void a measurement
        same = float(dev == tdev)
        bus_same = -1.0 if not (tbus and busid) else float(busid.lower().startswith(tbus))
        try:
            bus_num = int(busid.split(":")[1], 16) if busid.count(":") >= 2 else -1
        except ValueError:
            bus_num = -1
        mine = torch.tensor([float(rank), float(dev), float(bus_num), float(zlib.crc32(busid.encode()) & 0x7FFFFFFF), same, bus_same],
                            dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        seen_devices["rows"] = [[int(v) for v in e.cpu().tolist()] for e in every]

    def gather_step(stream_rows=False):
        # the only exchange of the algorithm: every rank's result rows to rank 0.  RCCL: rows are packed on the device and
        # gathered GPU to GPU over xGMI; gloo (dry runs): through host memory.
        if args.backend == "nccl":
            if stream_rows:
                if "s" not in rows_dev:
                    rows_dev["s"] = torch.empty((x0s.shape[0], solver.stream_row_doubles()), dtype=torch.float64, device="cuda")
                solver.stream_results_device(rows_dev["s"].data_ptr())
                got = sharding.gather_rows_device(dist, rows_dev["s"], world, rank)
            else:
                if "t" not in rows_dev:
                    rows_dev["t"] = torch.empty((B, solver.pack_results_device()), dtype=torch.float64, device="cuda")
                solver.pack_results_device(rows_dev["t"].data_ptr())
                got = sharding.gather_rows_device(dist, rows_dev["t"], world, rank)
            if rank == 0 and args.dump_rows:
                gathered["rows"] = np.concatenate([g.cpu().numpy() for g in got], axis=0)
        else:
            if stream_rows:
                r_ = solver.stream_results()
                rows = sharding.pack_results(r_["xs"], r_["us_squash"], r_["cost"], r_["iter"])
            else:
                rows = sharding.pack_results(solver.xs_batch, solver.us_squash_batch, solver.cost_batch, solver.iter_batch)
            got = sharding.gather_results(dist, rows, world, rank, device=coll_dev, global_batch=rows.shape[0] * world)
            if rank == 0 and args.dump_rows:
                gathered["rows"] = got

    def mpc_step():
        agg_s = {}
        for _ in range(MPC_CYCLES_PER_STEP):
            mpc.updateProblem(mpc_state["t"])
            solver.solve("previous", "previous", mpc.iters, x0s="plant")
            st = solver.stats()
            for k, v in st.items():
                agg_s[k] = agg_s.get(k, 0) + v
            solver.plant_step(MPC_DT_SIM)
            mpc_state["t"] += MPC_DT_SIM
        if dist is not None:
            gather_step()
        return agg_s

    def one_step():
        if is_mpc:
            return mpc_step()
        solver.solve([], [], args.maxiter, x0s=x0s)
        if dist is not None:
            gather_step()
        return solver.stats()

    agg = {}
    if stream:
        # warm-up: W x B rollouts through the slots (untimed); then the timed queue is made resident and processed
        if args.warmup > 0:
            solver.stream_begin(np.ascontiguousarray(np.resize(x0s, (args.warmup * B, d.nx))))
            solver.stream_run(args.maxiter)
        # (queue set-up -- allocation, upload of the initial states, zeroing of the result rows -- is timed on its own: `value`
        #  starts with the inputs resident in HBM, `value_including_queue_setup` charges the set-up as well)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        solver.stream_begin(x0s)
        torch.cuda.synchronize()
        queue_setup_s = time.perf_counter() - tq
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.stream_run(args.maxiter)
        if dist is not None:
            gather_step(stream_rows=True)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        agg = solver.stats()
        if dist is not None:
            gather_device_ids()
    else:
        for _ in range(args.warmup):
            one_step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = one_step()
            for k, v in st.items():
                agg[k] = agg.get(k, 0) + v
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            gather_device_ids()
    iters_rank = float(agg["total_iters"])
    ranks_seen, ranks_golden_ok = 1, None
    per_rank = None
    if dist is not None:
        # per-rank view (each rank's own iterations over its own clock), so that one line shows a slow or idle GPU
        mine = torch.tensor([iters_rank, elapsed], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "value": float(e[0].item()) / B / float(e[1].item()), "seconds": float(e[1].item())}
                    for r, e in enumerate(every)]
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # after the timed region: every rank checks its own GPU against the committed golden vector; the counts travel through
        # the same collective backend as the result rows
        ok = golden_rank_check(empc, local_dev, args.maxiter)
        tsum = torch.tensor([iters_rank, 1.0, 1.0 if ok else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        iters_total, ranks_seen = float(tsum[0].item()), int(tsum[1].item())
        ranks_golden_ok = int(tsum[2].item()) if ok is not None else None
    else:
        iters_total = iters_rank
    if rank == 0 and args.dump_rows and "rows" in gathered:
        np.save(args.dump_rows, gathered["rows"])

    if rank == 0:
        words = algorithmic_words(d.nx, d.ndx, d.nu)
        value = iters_total / B / elapsed
        import device_code_id as dci
        code_id = dci.device_code_id(empc.LIB_PATH)
        # the rollout kernel's algorithmic bytes: SURVEY.md section 8(d)'s per-(trajectory, knot) figure, once per launch -- the
        # step lengths of the line search share K / k / xs / us and only the accepted trial is a result (VERDICT r04 item 8:
        # charging the inputs to each of the ten step lengths made the figure 3.8 x the bytes the counters see)
        na = max(solver.stats_na(), 1)
        # dominant kernel of the timed region (HIP-event durations recorded on the solver's stream)
        kern = {"linearize": (agg["ms_linearize"], agg["n_linearize"], agg["linearize_units"]),
                "backward": (agg["ms_backward"], agg["n_backward"], agg["backward_units"]),
                "rollout": (agg["ms_rollout"], agg["n_rollout"], agg["rollout_units"] / na)}
        dom = max(kern, key=lambda k: kern[k][0])
        ms, nlaunch, units = kern[dom]
        bytes_per_launch = units / max(nlaunch, 1) * words[dom] * 8.0
        avg_ms = ms / max(nlaunch, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes of this workload at a full batch
        # (FETCH_SIZE / WRITE_SIZE, tools/run_profiles.sh): read from profiles/, not measured in this run
        # (tools/gpu_r5.sh profiles: separate --pmc passes over THIS command line in stream mode, summarised over the
        #  launches with a full grid only; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  No file for the
        #  workload / batch at hand: null.
        pmc, pmc_file, pmc_note = committed_counters(args.config, B, is_mpc, code_id)
        traffic = (pmc.get(dom) or {}).get("hbm_bytes_per_launch")
        per_launch = {k: {"avg_ms": kern[k][0] / max(kern[k][1], 1), "launches": kern[k][1],
                          "algorithmic_GBs": (kern[k][2] / max(kern[k][1], 1) * words[k] * 8.0) / (kern[k][0] / max(kern[k][1], 1) * 1e-3) / 1e9
                          if kern[k][0] > 0 else 0.0} for k in kern}
        for k in per_launch:
            per_launch[k]["frac_of_8TBs"] = per_launch[k]["algorithmic_GBs"] / HBM_PEAK_GBS
            # compute side (the memory side binds only by ~2x): FP64 vector instructions per full-batch
            # launch counted by the SQ (profiles/r04_pmc_<config>.json), x 64 lanes x (2 for an FMA) / this run's launch time
            # against the 78.6 TFLOP/s FP64 vector peak; valu_busy = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES of that pass
            c_ = pmc.get(k) or {}
            if c_.get("fp64_flop_per_launch") and per_launch[k]["avg_ms"] > 0 and c_.get("units_per_launch"):
                scale = (kern[k][2] / max(kern[k][1], 1)) / c_["units_per_launch"]  # this run's launches may be less full
                fl = c_["fp64_flop_per_launch"] * scale
                per_launch[k]["fp64_TFLOPs"] = fl / (per_launch[k]["avg_ms"] * 1e-3) / 1e12
                per_launch[k]["fp64_frac_of_78.6TF"] = per_launch[k]["fp64_TFLOPs"] / FP64_PEAK_TFLOPS
                per_launch[k]["fp64_flop_per_unit"] = c_["fp64_flop_per_launch"] / c_["units_per_launch"]
                per_launch[k]["valu_active_frac_of_wave_cycles"] = c_.get("valu_active_frac")
                per_launch[k]["hbm_bytes_per_launch_measured"] = c_.get("hbm_bytes_per_launch")
        if is_mpc:
            workload = ("%s closed loop on %s: %d-knot horizon dt=%dms, %d plants/GPU (RK4, %d ms), %d cycles/step, "
                        "%d iterations/cycle, warm start and plant states device-resident" %
                        (type(mpc).__name__, rel, d.T + 1, mpc.dt, B, MPC_DT_SIM, MPC_CYCLES_PER_STEP, mpc.iters))
        elif stream:
            workload = ("%s dt=%dms T=%d, SolverSbFDDP.solve(maxiter=%d) from %d perturbed x0 per GPU streamed through %d slots "
                        "(one step = %d solves; a slot that finishes takes the next x0 in the same sweep; queue and result rows "
                        "resident in HBM)" % (rel, dt, d.T, args.maxiter, x0s.shape[0], B, B))
        else:
            workload = ("%s dt=%dms T=%d batch=%d/GPU SolverSbFDDP.solve(maxiter=%d), perturbed x0, one plain batched solve per step" %
                        (rel, dt, d.T, B, args.maxiter))
        out = {
            "metric": "DDP iters/sec (batch=%d per GPU, %d knots)" % (B, d.T),
            "value": value,
            "unit": "batched DDP iterations/s (trajectory-iterations/s / %d)" % B,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "device_code_id": code_id,
            "config": {"workload": workload, "mode": "mpc" if is_mpc else args.mode,
                       "nx": d.nx, "ndx": d.ndx, "nu": d.nu, "parallelism": "batch-sharded x%d" % world},
            "trajectory_iters_per_s": iters_total / elapsed,
            "mean_iters_per_trajectory": iters_total / (B * world * args.steps),
            "sweeps_per_step": agg["sweeps"] / args.steps,
            # (per-launch averages of the timed sweeps x sweeps per step: every EMPC_TIMING_EVERY-th sweep carries events)
            "kernel_ms_per_step": {k: kern[k][0] / max(kern[k][1], 1) * agg["sweeps"] / args.steps for k in kern},
            "other_kernels_avg_ms": {"select": agg["ms_select"] / max(agg["n_select"], 1), "calc": agg["ms_calc"] / max(agg["n_calc"], 1)},
            "ms_per_sweep": elapsed * 1e3 / max(agg["sweeps"], 1),
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_bandwidth": achieved / HBM_COPY_GBS, "traffic": traffic,
                         "traffic_source": ("%s (committed PMC passes over a stream run, full-batch launches only); %s" % (pmc_file, pmc_note)) if traffic else pmc_note,
                         "counters_commit": None if not pmc_file else json.load(open(os.path.join(ROOT, pmc_file))).get("commit"),
                         "achieved_from_counter_bytes_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         "algorithmic_bytes_per_unit": words[dom] * 8, "units_per_launch": units / max(nlaunch, 1),
                         "unit_definition": "(trajectory, knot); a rollout launch tries all %d step lengths of the line search on it" % na
                                            if dom == "rollout" else "(trajectory, knot)",
                         "avg_launch_ms": avg_ms,
                         "limiter": limiter_text(dom, pmc)},
            "kernels": per_launch,
            "iteration_roofline": {"algorithmic_bytes_per_traj_knot_iter": words["iteration"] * 8,
                                   "achieved_GBs": iters_total * d.T * words["iteration"] * 8 / elapsed / 1e9 / world,
                                   "frac_of_8TBs": iters_total * d.T * words["iteration"] * 8 / elapsed / 1e9 / world / HBM_PEAK_GBS},
        }
        if per_launch[dom].get("fp64_TFLOPs") is not None:
            # compute side of the dominant kernel: FP64 flop issued per launch (SQ instruction counters of the committed PMC
            # pass, scaled to this run's units per launch) over this run's launch time, against the FP64 vector peak
            out["roofline"]["compute"] = {"bound": "fp64-valu", "achieved": per_launch[dom]["fp64_TFLOPs"], "peak": FP64_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": per_launch[dom]["fp64_frac_of_78.6TF"],
                                          "flop_per_unit": per_launch[dom]["fp64_flop_per_unit"],
                                          "valu_active_frac_of_wave_cycles": per_launch[dom]["valu_active_frac_of_wave_cycles"],
                                          "source": pmc_file}
        if dist is not None:
            devs = seen_devices.get("rows", [])
            distinct = len(set((r[1], r[2], r[3]) for r in devs))
            out["rccl_ranks_seen"] = {"backend": args.backend, "ranks": len(devs), "distinct_devices": distinct,
                                      "rank_device_busnumber": [[r[0], r[1], r[2]] for r in devs],
                                      "solver_device_is_torch_device": [bool(r[4]) for r in devs],
                                      "pci_bus_id_matches_torch": [r[5] if len(r) > 5 else -1 for r in devs],
                                      "source": "empc_solver_device_info (where the solver's memory lives), gathered after the timed region"}
            if args.backend == "nccl" and (len(devs) != args.gpus or distinct != args.gpus or not all(r[4] for r in devs)):
                raise SystemExit("multi-GPU self-check failed: %d ranks on %d distinct GPUs, --gpus %d: %s" %
                                 (len(devs), distinct, args.gpus, json.dumps(out["rccl_ranks_seen"])))
            out["ranks_seen"] = ranks_seen
            out["ranks_matching_golden_vector"] = ranks_golden_ok
            out["per_rank"] = per_rank
        if is_mpc:
            out["mpc_cycles_per_s"] = args.steps * MPC_CYCLES_PER_STEP / elapsed
            out["plant_controller_cycles_per_s"] = args.steps * MPC_CYCLES_PER_STEP * B * world / elapsed
        if stream:
            out["value_including_queue_setup"] = iters_total / B / (elapsed + queue_setup_s)
            out["queue_setup_ms"] = queue_setup_s * 1e3
            # the line checks its own product: every job of rank 0's queue must have handed over a finished row
            res = solver.stream_results()
            done = (res["status"] & 7) != 0
            out["stream_rows_checked"] = {"rows": int(len(done)), "rows_with_a_final_status": int(done.sum()),
                                          "rows_converged": int(((res["status"] & 1) != 0).sum()),
                                          "rows_finite": int(np.isfinite(res["xs"]).all(axis=(1, 2)).sum()),
                                          "iterations_in_rows": int((res["iter"] + 1).sum()),
                                          "iterations_counted_by_the_device": int(iters_rank)}
            if not done.all():
                raise SystemExit("stream self-check failed: %s" % json.dumps(out["stream_rows_checked"]))
        if stream and world == 1 and not args.no_single_batch:
            # the same rollouts as plain batched solves (the latency view: one batch at a time, stragglers included); rows of
            # the stream against the plain solve of the same initial state: bit for bit (continuous batching changes the
            # schedule, not the arithmetic) -- the comparisons run outside the timed part
            nb = min(2, args.steps)
            aggb, elb, bitwise = {}, 0.0, []
            for k_ in range(nb):
                torch.cuda.synchronize()
                tb = time.perf_counter()
                solver.solve([], [], args.maxiter, x0s=x0s[k_ * B:(k_ + 1) * B])
                torch.cuda.synchronize()
                elb += time.perf_counter() - tb
                for k, v in solver.stats().items():
                    aggb[k] = aggb.get(k, 0) + v
                pxs, pus, pc, pit = solver.xs_batch, solver.us_squash_batch, solver.cost_batch, solver.iter_batch
                for i in ([0, 1, 17 % B, B // 2, B - 1] if k_ == 0 else [0, 5 % B, B - 1]):
                    j = k_ * B + i
                    bitwise.append(bool(np.array_equal(res["xs"][j], pxs[i]) and np.array_equal(res["us_squash"][j], pus[i]) and
                                        res["cost"][j] == pc[i] and res["iter"][j] == pit[i]))
            out["stream_rows_checked"]["rows_compared_bitwise_with_plain_solves"] = len(bitwise)
            out["stream_rows_checked"]["rows_bitwise_equal"] = int(sum(bitwise))
            if not all(bitwise):
                raise SystemExit("stream self-check failed: a streamed row differs from the plain solve of its initial state: %s" %
                                 json.dumps(out["stream_rows_checked"]))
            out["single_batch"] = {"value": aggb["total_iters"] / B / elb, "ms_per_solve": elb / nb * 1e3, "solves": nb,
                                   "sweeps_per_solve": aggb["sweeps"] / nb,
                                   "note": "plain empc_solver_solve of one batch at a time: ~3/4 of its sweeps run on the few "
                                           "rollouts that need many iterations"}
        # The side views and the CPU leg run AFTER the timed region: whatever goes wrong in one of them (a host without a compiler, a
        # box short of memory, an assertion of the parity harness) is reported in its block; it must not take the measured line with it.
        def guarded(key, fn):
            try:
                return fn()
            except Exception as e:
                out[key + "_error"] = "%s: %s" % (type(e).__name__, str(e)[:400])
                return None
        if stream and world == 1 and not args.no_slots_sweep:
            out["slots_sweep"] = guarded("slots_sweep", lambda: slots_sweep(empc, problem, d, args, local_dev, B, iters_total / B / elapsed))
        if world == 1 and not is_mpc and not args.no_secondary and args.config != "displacement":
            out["secondary"] = guarded("secondary", lambda: secondary_displacement(empc, B, args.maxiter, local_dev))
        if not args.no_cpu_baseline and not is_mpc and world == 1:  # rank 0 at N = 1 only
            leg = guarded("cpu_baseline", lambda: cpu_baseline_and_parity(empc, solver, problem, d, x0s[:B], B, args.maxiter, out["unit"]))
            out.update(leg if leg is not None else {"cpu_baseline": None, "parity": None})
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
