#!/usr/bin/env python3
"""GPU box: every rollout of the full-size batches against the CPU oracle (identical inputs), per config.

Writes gpurun_out/parity_sweep.json: max-abs errors on xs / us / us_squash, relative cost error, iteration-count agreement.
The oracle is the checker here (OpenMP over rollouts on the host cores); it is never part of what is measured.
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import empc_loader, oracle_binding as ob
empc = empc_loader.load()
CONFIGS = {"displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80, 1024),
           "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13, 1024),
           "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32, 1024),
           "hover": ("hexacopter370/trajectories/hover.yaml", 40, 256)}
out = {}
for name, (rel, dt, B) in CONFIGS.items():
    t = empc.Trajectory(); t.autoSetup(empc.yaml_path(rel)); p = t.createProblem(dt, True, "IntegratedActionModelEuler")
    d = p.desc
    x0s = empc.perturbed_x0s(p.x0, B, nq=d.model.nq)
    s = empc.SolverSbFDDP(p, batch=B)
    s.solve([], [], 100, x0s=x0s)
    t0 = time.time()
    r = ob.solve_batch(d, x0s, 100, nthreads=min(os.cpu_count(), len(os.sched_getaffinity(0)), 16), want_traj=True)
    same = s.iter_batch == r["iter"]
    ex = np.abs(s.xs_batch - r["xs"]).reshape(B, -1).max(axis=1)
    eu = np.abs(s.us_batch - r["us"]).reshape(B, -1).max(axis=1)
    es = np.abs(s.us_squash_batch - r["us_squash"]).reshape(B, -1).max(axis=1)
    ec = np.abs(s.cost_batch - r["cost"]) / (1 + np.abs(r["cost"]))
    ok = same & (ex < 1e-4) & (eu < 1e-4)
    # rollouts the oracle itself solved: converged flag set, no inner loop gave up at the iteration or regularisation limit
    conv = ((r["status"] & 1) != 0) & ((r["status"] & (2 | 4)) == 0) & np.isfinite(r["cost"]) & (np.abs(r["cost"]) < 1e6)
    gconv = ((s.status_batch & 1) != 0) & ((s.status_batch & (2 | 4)) == 0)
    out[name] = {"rollouts": B, "knots": d.T, "iterations_equal": int(same.sum()), "within_1e-4_xs_us_and_iterations_equal": int(ok.sum()),
                 "xs_max_abs_err_where_iterations_equal": float(ex[same].max()) if same.any() else None,
                 "us_max_abs_err_where_iterations_equal": float(eu[same].max()) if same.any() else None,
                 "us_squash_max_abs_err_where_iterations_equal": float(es[same].max()) if same.any() else None,
                 "cost_max_rel_err_where_iterations_equal": float(ec[same].max()) if same.any() else None,
                 "solved_by_oracle": int(conv.sum()), "solved_by_gpu": int(gconv.sum()), "solved_by_both": int((conv & gconv).sum()),
                 "solved_by_oracle_and_within_tolerance_on_gpu": int((conv & ok).sum()),
                 "xs_max_abs_err_among_those": float(ex[conv & ok].max()) if (conv & ok).any() else None,
                 "solved_by_both_but_other_iteration_count": int((conv & gconv & ~same).sum()),
                 "xs_err_quantiles_among_those_50_90_100": [float(q) for q in np.quantile(ex[conv & gconv & ~same], [0.5, 0.9, 1.0])] if (conv & gconv & ~same).any() else None,
                 "cost_rel_err_quantiles_among_those_50_90_100": [float(q) for q in np.quantile(ec[conv & gconv & ~same], [0.5, 0.9, 1.0])] if (conv & gconv & ~same).any() else None,
                 "xs_err_quantiles_solved_by_both_same_iterations_50_90_100": [float(q) for q in np.quantile(ex[conv & gconv & same], [0.5, 0.9, 1.0])] if (conv & gconv & same).any() else None,
                 "iteration_range_gpu": [int(s.iter_batch.min()), int(s.iter_batch.max())], "oracle_seconds": round(time.time() - t0, 1)}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_sweep.json"), "w"), indent=1)
