#!/usr/bin/env python3
"""Which rule of the restated algorithm makes perturbed rollouts blow up?  (CPU only; test infrastructure.)

VERDICT r02 weak #3: from x0 +- 0.05 only 58 of 256 hovers and 952 of 1024 eagle_catch rollouts are solved by the oracle (the
GPU agrees), with xs of 1e13 on the others.  This tool follows every rollout of a batch through the oracle's iteration trace
and answers, per rollout that is not solved:
  * at which iteration the cost first explodes (x 100 within one iteration, or beyond 1e6),
  * which acceptance branch of solveFDDP took that step: the ascent branch (dVexp < 0, dV > th_acceptnegstep dVexp,
    src/sbfddp.cpp:280-288), the descent branch (:271-279: d0 < th_grad or dV > th_acceptstep dVexp), or the DDP clean-up's
    accept-anything-while-infeasible (:359),
and then re-solves the batch under each single change of an option SURVEY Appendix A.8 marks as uncertain (U1-U4) or of the
ascent threshold, reporting how many rollouts each variant solves.

Usage: python tools/divergence_study.py [--batch 256] [--out profiles/r03_divergence_study.json]
"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import empc_loader  # noqa: E402
import oracle_binding as ob  # noqa: E402

empc = empc_loader.load()
CONFIGS = {"hover": ("hexacopter370/trajectories/hover.yaml", 40),
           "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
           "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80)}


def solved(status, cost):
    return bool((status & 1) and not (status & 6) and np.isfinite(cost) and abs(cost) < 1e6)


def follow(d, prm, x0, maxiter):
    o = ob.OracleSolver(d, prm)
    o.set_x0(x0)
    o.solve(None, None, maxiter)
    r, tr = o.result(), o.trace()
    out = dict(solved=solved(r["status"], r["cost"]), iters=int(r["iter"]) + 1, status=int(r["status"]), cost=float(r["cost"]))
    if out["solved"] or len(tr) < 2:
        return out
    # record: phase iter cost stop xreg steplength feasible dV dVexp gapnorm d0 d1
    cost = tr[:, 2]
    first = None
    for i in range(1, len(tr)):
        if not np.isfinite(cost[i]) or cost[i] > 1e6 or (cost[i] > 100.0 * max(cost[i - 1], 1e-3)):
            first = i
            break
    if first is None:
        out["explosion"] = None  # did not converge but never exploded (iteration limit / regularisation limit)
        return out
    rec = tr[first]
    ddp = rec[0] == 100
    dV, dVexp, d0 = rec[7], rec[8], rec[10]
    if ddp:
        branch = "ddp_accept_while_infeasible" if tr[first - 1][6] == 0 else "ddp_descent"
    elif dVexp < 0:
        branch = "ascent_branch"
    elif d0 < prm.th_grad:
        branch = "descent_branch_d0_below_th_grad"
    else:
        branch = "descent_branch"
    out["explosion"] = dict(iteration=int(first), pass_=int(rec[0]), steplength=float(rec[5]), cost_before=float(cost[first - 1]),
                            cost_after=float(cost[first]), dV=float(dV), dVexp=float(dVexp), d0=float(d0), d1=float(rec[11]),
                            xreg=float(rec[4]), feasible_before=int(tr[first - 1][6]), branch=branch,
                            cost_increase_accepted=bool(dV < 0))
    return out


def study(name, B, maxiter, workers):
    rel, dt = CONFIGS[name]
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(rel))
    problem = t.createProblem(dt, True, "IntegratedActionModelEuler")
    d = problem.desc
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq)
    base = ob.default_params()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        rows = list(pool.map(lambda b: follow(d, base, x0s[b], maxiter), range(B)))
    unsolved = [r for r in rows if not r["solved"]]
    branches = {}
    for r in unsolved:
        key = r["explosion"]["branch"] if r.get("explosion") else "no_explosion_iteration_or_regularisation_limit"
        branches[key] = branches.get(key, 0) + 1
    ex = [r["explosion"] for r in unsolved if r.get("explosion")]
    out = {"workload": "%s dt=%dms T=%d, x0 + 0.05 U(-1,1), %d rollouts, maxiter %d" % (rel, dt, d.T, B, maxiter),
           "solved": B - len(unsolved), "unsolved": len(unsolved), "first_explosion_by_branch": branches,
           "explosions_accepting_a_cost_increase": int(sum(e["cost_increase_accepted"] for e in ex)),
           "explosion_iteration_min_median_max": [int(min(e["iteration"] for e in ex)), float(np.median([e["iteration"] for e in ex])),
                                                  int(max(e["iteration"] for e in ex))] if ex else None,
           "explosion_steplength_histogram": {str(a): int(sum(e["steplength"] == a for e in ex)) for a in sorted(set(e["steplength"] for e in ex))},
           "examples": [dict(rollout=i, **rows[i]) for i in range(B) if not rows[i]["solved"]][:6]}
    # single-option variants
    variants = {"baseline": {},
                "no_ascent_steps (th_acceptnegstep = 0: a step must reduce the cost)": {"th_acceptnegstep": 0.0},
                "ascent_threshold_1 (th_acceptnegstep = 1)": {"th_acceptnegstep": 1.0},
                "U1 stop = expected reduction |d0 + d1/2|": {"stop_criteria": 1},
                "U1 stop = sum |Qu|^2 (upstream DDP)": {"stop_criteria": 2},
                "U1 gap norm = Linf": {"gap_norm": 1},
                "U2 terminal cost not scaled by dt": {"terminal_dt_scaling": 0},
                "U3 smooth-sat d^4": {"smoothsat_power": 4},
                "U4 th_gaptol = 1e-9": {"th_gaptol": 1e-9},
                "th_acceptstep = 0.01 (easier descent acceptance)": {"th_acceptstep": 0.01},
                "reg_init = 1e-3": {"reg_init": 1e-3}}
    res = {}
    for label, changes in variants.items():
        prm = ob.default_params()
        for k, v in changes.items():
            setattr(prm, k, v)
        r = ob.solve_batch(d, x0s, maxiter, nthreads=workers, params=prm, want_traj=False)
        ok = np.array([solved(int(s), float(c)) for s, c in zip(r["status"], r["cost"])])
        res[label] = {"solved": int(ok.sum()), "mean_iterations_of_solved": float((r["iter"][ok] + 1).mean()) if ok.any() else None,
                      "median_cost_of_solved": float(np.median(r["cost"][ok])) if ok.any() else None}
    out["variants"] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--maxiter", type=int, default=100)
    ap.add_argument("--configs", default="hover,eagle_catch")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_divergence_study.json"))
    args = ap.parse_args()
    workers = min(os.cpu_count() or 1, 32)
    out = {"tool": "tools/divergence_study.py", "oracle": "oracle/liboracle.so (CPU restatement; the GPU solver follows the same rules)"}
    for name in args.configs.split(","):
        out[name] = study(name, args.batch, args.maxiter, workers)
        print(name, json.dumps({k: v for k, v in out[name].items() if k != "examples"}, indent=1))
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
