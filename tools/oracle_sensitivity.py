#!/usr/bin/env python3
"""Oracle-vs-oracle' sensitivity study (CPU only; test infrastructure, never part of the product path).

Question: when the GPU solver and the oracle disagree on a perturbed eagle_catch / hover rollout,
is that a defect of the GPU path or the conditioning of the algorithm on that problem?  This tool answers it without a
GPU: it solves the SAME batch of perturbed rollouts with variants of the oracle that differ only in rounding --

  base   oracle/liboracle.so as shipped (g++ -O3 -ffp-contract=off)
  fma    the same sources built with -ffp-contract=fast (x86 FMA contraction, i.e. "another correct compiler")
  ulp    base library, every component of x0 moved to the adjacent double (random direction, quaternion renormalised)

-- and reports, per pair, how many rollouts end with the same iteration count and within 1e-4 on xs/us (the north-star
tolerance), next to the committed GPU-vs-oracle figures.  If base-vs-fma and base-vs-ulp disagree as often as
GPU-vs-base, the disagreement is a property of the problem, not of the device code.

Usage: python tools/oracle_sensitivity.py [--batch 256] [--configs eagle_catch,hover,displacement] [--out profiles/r02_oracle_sensitivity.json]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import empc_loader  # noqa: E402
import oracle_binding as ob  # noqa: E402

empc = empc_loader.load()
T = empc.T
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

CONFIGS = {"displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
           "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
           "hover": ("hexacopter370/trajectories/hover.yaml", 40),
           "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13)}
VARIANT_DIR = os.path.join(ROOT, "oracle", "_variants")  # git-ignored build products
BASE_FLAGS = "-O3 -march=x86-64-v3 -fPIC -std=c++17 -fopenmp -Wall -Wno-unused-variable -Wno-unused-but-set-variable".split()


def build_variant(name, extra):
    os.makedirs(VARIANT_DIR, exist_ok=True)
    out = os.path.join(VARIANT_DIR, "liboracle_%s.so" % name)
    src = os.path.join(ROOT, "oracle", "oracle_api.cpp")
    deps = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++"] + BASE_FLAGS + extra + ["-shared", "-o", out, src])
    L = C.CDLL(out)
    L.oracle_solve_batch.restype = C.c_double
    L.oracle_solve_batch.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams), C.c_int, _dp, C.c_int, C.c_int,
                                     _dp, _dp, _dp, _dp, _ip, _ip]
    return L


def solve_with(L, desc, prm, x0s, maxiter, nthreads):
    x0s = np.ascontiguousarray(x0s, dtype=np.float64)
    B = x0s.shape[0]
    xs = np.zeros((B, desc.T + 1, desc.nx))
    us = np.zeros((B, desc.T, desc.nu))
    usq = np.zeros((B, desc.T, desc.nu))
    cost = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_dp)
    L.oracle_solve_batch(C.byref(desc), C.byref(prm), B, p(x0s), int(maxiter), int(nthreads), p(xs), p(us), p(usq), p(cost),
                         iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip))
    return dict(xs=xs, us=us, cost=cost, iter=iters, status=status)


def solved(r):
    return ((r["status"] & 1) != 0) & ((r["status"] & (2 | 4)) == 0) & np.isfinite(r["cost"]) & (np.abs(r["cost"]) < 1e6)


def compare(a, b, tol=1e-4):
    B = a["iter"].shape[0]
    same = a["iter"] == b["iter"]
    ex = np.abs(a["xs"] - b["xs"]).reshape(B, -1).max(axis=1)
    eu = np.abs(a["us"] - b["us"]).reshape(B, -1).max(axis=1)
    ec = np.abs(a["cost"] - b["cost"]) / (1 + np.abs(a["cost"]))
    ok = same & (ex < tol) & (eu < tol)
    sa, sb = solved(a), solved(b)
    both = sa & sb
    q = lambda v, m: [float(x) for x in np.quantile(v[m], [0.5, 0.9, 1.0])] if m.any() else None
    return {"rollouts": int(B), "iterations_equal": int(same.sum()), "within_1e-4_and_iterations_equal": int(ok.sum()),
            "solved_by_a": int(sa.sum()), "solved_by_both": int(both.sum()),
            "solved_by_a_and_within_tolerance": int((sa & ok).sum()),
            "solved_by_both_other_iteration_count": int((both & ~same).sum()),
            "xs_err_q50_q90_max_where_iterations_differ_both_solved": q(ex, both & ~same),
            "cost_rel_err_q50_q90_max_where_iterations_differ_both_solved": q(ec, both & ~same),
            "xs_err_q50_q90_max_where_iterations_equal": q(ex, same),
            "cost_rel_err_q50_q90_max_solved_by_both": q(ec, both)}


def bind_solver_api(L):
    L.oracle_solver_create.restype = C.c_void_p
    L.oracle_solver_create.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams)]
    L.oracle_solver_destroy.argtypes = [C.c_void_p]
    L.oracle_solver_set_x0.argtypes = [C.c_void_p, _dp]
    L.oracle_solver_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int]
    L.oracle_solver_trace.argtypes = [C.c_void_p, _dp, C.c_int]


def trace_of(L, desc, prm, x0, maxiter):
    h = C.c_void_p(L.oracle_solver_create(C.byref(desc), C.byref(prm)))
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    L.oracle_solver_set_x0(h, x0.ctypes.data_as(_dp))
    L.oracle_solver_solve(h, None, None, int(maxiter), 0)
    n = L.oracle_solver_trace(h, None, 0)
    tr = np.zeros((n, 12))
    L.oracle_solver_trace(h, tr.ctypes.data_as(_dp), n)
    L.oracle_solver_destroy(h)
    return tr


def first_divergence(ta, tb, rtol=1e-6):
    """index of the first iteration record at which two traces differ: another step length / feasibility / pass, or a
    cost that differs by more than rtol relative; len(shorter) when one is a prefix of the other"""
    n = min(len(ta), len(tb))
    for i in range(n):
        a, b = ta[i], tb[i]
        if a[0] != b[0] or a[1] != b[1] or a[5] != b[5] or a[6] != b[6] or abs(a[2] - b[2]) > rtol * (1 + abs(a[2])):
            return i
    return n


def _trace_job(args):
    name, b, maxiter = args
    rel, dt = CONFIGS[name]
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(rel))
    pb = tr.createProblem(dt, True, "IntegratedActionModelEuler")
    d = pb.desc
    base = build_variant("base", ["-ffp-contract=off"])
    fma = build_variant("fma", ["-ffp-contract=fast"])
    bind_solver_api(base)
    bind_solver_api(fma)
    x0s = empc.perturbed_x0s(pb.x0, b + 1, nq=d.model.nq)
    prm = ob.default_params()
    prm13 = ob.default_params()
    prm13.th_gaptol = 1e-13
    t_base = trace_of(base, d, prm, x0s[b], maxiter)
    t_fma = trace_of(fma, d, prm13, x0s[b], maxiter)
    t_ulp = trace_of(base, d, prm, nudge_ulp(x0s)[b], maxiter)
    return b, len(t_base), first_divergence(t_base, t_fma), first_divergence(t_base, t_ulp)


def trace_study(name, n, maxiter, procs):
    """first iteration at which the oracle and its rounding variants part ways, per rollout"""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        rows = pool.map(_trace_job, [(name, b, maxiter) for b in range(n)])
    rows = np.array(rows)
    q = lambda v: [int(x) for x in np.quantile(v, [0.0, 0.05, 0.25, 0.5])]
    return {"rollouts": int(n), "iterations_base_q0_q5_q25_q50": q(rows[:, 1]),
            "first_divergent_iteration_base_vs_fma_q0_q5_q25_q50": q(rows[:, 2]),
            "first_divergent_iteration_base_vs_ulp_q0_q5_q25_q50": q(rows[:, 3]),
            "rollouts_identical_path_base_vs_fma": int((rows[:, 2] >= rows[:, 1]).sum()),
            "rollouts_identical_path_base_vs_ulp": int((rows[:, 3] >= rows[:, 1]).sum())}


def nudge_ulp(x0s, seed=7):
    rng = np.random.default_rng(seed)
    out = x0s.copy()
    up = rng.integers(0, 2, size=out.shape).astype(bool)
    out = np.where(up, np.nextafter(out, np.inf), np.nextafter(out, -np.inf))
    out[:, 3:7] /= np.linalg.norm(out[:, 3:7], axis=1, keepdims=True)
    return out


def options_study(out_path):
    """The option branches whose GPU tests use a relaxed criterion (tests/test_gpu_box_solvers.py,
    tests/test_gpu_contact_options.py): the oracle against its -ffp-contract=fast build on exactly the inputs of those tests --
    first divergent iteration record, relative difference of the FIRST iteration's cost, final xs difference."""
    import pathlib
    import tempfile
    from conftest import contact_variant
    base = build_variant("base", ["-ffp-contract=off"])
    fma = build_variant("fma", ["-ffp-contract=fast"])
    bind_solver_api(base)
    bind_solver_api(fma)

    def study(d, prm, x0s, maxiter):
        fd, c0, it_a, it_b = [], [], [], []
        for b in range(x0s.shape[0]):
            ta, tb = trace_of(base, d, prm, x0s[b], maxiter), trace_of(fma, d, prm, x0s[b], maxiter)
            fd.append(first_divergence(ta, tb, rtol=1e-5))
            c0.append(float(abs(ta[0][2] - tb[0][2]) / (1 + abs(ta[0][2]))))
            it_a.append(len(ta))
            it_b.append(len(tb))
        ra, rb = solve_with(base, d, prm, x0s, maxiter, 4), solve_with(fma, d, prm, x0s, maxiter, 4)
        ex = np.abs(ra["xs"] - rb["xs"]).reshape(x0s.shape[0], -1).max(axis=1)
        return {"first_divergent_record": [int(v) for v in fd], "first_iteration_cost_rel_diff": c0,
                "iteration_records_base": it_a, "iteration_records_fma": it_b, "xs_max_abs_diff": [float(v) for v in ex]}

    res = {"note": "oracle (-ffp-contract=off) vs the same sources with -ffp-contract=fast on the inputs of the GPU tests of the "
                   "option branches; first_divergent_record uses rtol 1e-5 on the cost like tests/parity_criteria.py"}
    box = {}
    for name, dt in [("hover", 40), ("displacement", 80), ("eagle_catch", 32), ("push_slide", 13)]:
        tr = empc.Trajectory()
        tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
        pb = tr.createProblem(dt, False, "IntegratedActionModelEuler")
        d = pb.desc
        x0s = empc.perturbed_x0s(pb.x0, 8, nq=d.model.nq)
        for st, sname in ((1, "SolverBoxFDDP"), (2, "SolverBoxDDP")):
            prm = ob.default_params()
            prm.solver_type = st
            box["%s/%s" % (name, sname)] = study(d, prm, x0s, 30)
    res["box_solvers_cold_start_B8_maxiter30"] = box
    con = {}
    with tempfile.TemporaryDirectory() as td:
        for contact, gains in [("ContactModel3D", (0.0, 0.0)), ("ContactModel3D", (9.0, 4.0)), ("ContactModel6D", (0.0, 0.0)),
                               ("ContactModel6D", (11.0, 5.0))]:
            _, pb = contact_variant(empc, pathlib.Path(td), contact, gains)
            d = pb.desc
            x0s = empc.perturbed_x0s(pb.x0, 4, nq=d.model.nq, amplitude=0.02)
            x0s[0] = pb.x0
            con["%s gains %s" % (contact, list(gains))] = study(d, ob.default_params(), x0s, 100)
    res["eagle_catch_contact_options_B4_maxiter100"] = con
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    for k in ("box_solvers_cold_start_B8_maxiter30", "eagle_catch_contact_options_B4_maxiter100"):
        for name, v in res[k].items():
            print(name, "first divergent record", v["first_divergent_record"], "| first-iteration cost diff max %.1e" %
                  max(v["first_iteration_cost_rel_diff"]), "| xs diff max %.1e" % max(v["xs_max_abs_diff"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--options", action="store_true",
                    help="study the option branches (box solvers, contact options) instead; writes profiles/r02_oracle_sensitivity_options.json")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--configs", default="eagle_catch,hover,displacement")
    ap.add_argument("--maxiter", type=int, default=100)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--trace-rollouts", type=int, default=64, help="rollouts of the per-iteration trace comparison (0 = skip)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_oracle_sensitivity.json"))
    args = ap.parse_args()
    ob.build_oracle()
    if args.options:
        options_study(os.path.join(ROOT, "profiles", "r02_oracle_sensitivity_options.json"))
        return
    base = build_variant("base", ["-ffp-contract=off"])
    fma = build_variant("fma", ["-ffp-contract=fast"])
    res = {"note": "CPU oracle against rounding-only variants of itself on identical perturbed rollouts "
                   "(benchmark/utils/utils.hpp:15-27 recipe, seed as in bench.py); tolerance 1e-4 on xs/us",
           "variants": {"base": "g++ -O3 -march=x86-64-v3 -ffp-contract=off (the shipped oracle)",
                        "fma": "same sources, -ffp-contract=fast (FMA contraction)",
                        "ulp": "base library, each component of x0 moved to the adjacent double",
                        "gaptol13": "base library with th_gaptol = 1e-13 (the device's feasibility tolerance)"}}
    for name in args.configs.split(","):
        rel, dt = CONFIGS[name]
        tr = empc.Trajectory()
        tr.autoSetup(empc.yaml_path(rel))
        pb = tr.createProblem(dt, True, "IntegratedActionModelEuler")
        d = pb.desc
        x0s = empc.perturbed_x0s(pb.x0, args.batch, nq=d.model.nq)
        prm = ob.default_params()
        prm13 = ob.default_params()
        prm13.th_gaptol = 1e-13
        t0 = time.time()
        r_base = solve_with(base, d, prm, x0s, args.maxiter, args.threads)
        r_g13 = solve_with(base, d, prm13, x0s, args.maxiter, args.threads)
        r_fma = solve_with(fma, d, prm13, x0s, args.maxiter, args.threads)  # FMA breaks diff(x, x) == 0 exactly: needs the 1e-13 gap tolerance
        r_ulp = solve_with(base, d, prm, nudge_ulp(x0s), args.maxiter, args.threads)
        res[name] = {"knots": d.T, "iteration_range_base": [int(r_base["iter"].min()), int(r_base["iter"].max())],
                     "mean_iterations_base": float(r_base["iter"].mean() + 1),
                     "base_vs_gaptol13": compare(r_base, r_g13), "base_vs_fma": compare(r_base, r_fma),
                     "base_vs_ulp": compare(r_base, r_ulp), "seconds": round(time.time() - t0, 1)}
        if args.trace_rollouts > 0 and name != "displacement":
            res[name]["trace_study"] = trace_study(name, min(args.trace_rollouts, args.batch), args.maxiter, args.threads)
        print(name, json.dumps(res[name], indent=1), flush=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
