"""Statistics of a free-running batch against the oracle (perturbed eagle_catch, perturbed hover).  Test infrastructure:
used by tests/ and by bench.py's parity leg, never by the product path.

These numbers are REPORTED, not asserted: the free-running iteration paths of the contact problem are rounding-sensitive
(profiles/r02_oracle_sensitivity.json: the oracle against its own FMA build ends with the same iteration count and within
1e-4 on 178 of 239 rollouts), so no threshold on an agreement rate says anything about the device code.  The parity claim
is made step by step by tests/stepwise.py (teacher-forced in both directions + the same minimiser from a common restart).
What IS asserted from here: both sides solve the same problem -- the oracle's cost at the GPU's final point equals the
GPU's cost (1e-9) and the GPU's xs is the rollout of its us under the oracle's dynamics (defect 1e-8)."""
import numpy as np

EARLY_K = 3  # (only used to report how many rollouts share their first records with the oracle)
TOL = 1e-4


def solved(status, cost):
    return ((status & 1) != 0) & ((status & (2 | 4)) == 0) & np.isfinite(cost) & (np.abs(cost) < 1e6)


def first_divergence(ta, tb, rtol=1e-5):
    n = min(len(ta), len(tb))
    for i in range(n):
        a, b = ta[i], tb[i]
        if a[0] != b[0] or a[1] != b[1] or a[5] != b[5] or a[6] != b[6] or abs(a[2] - b[2]) > rtol * (1 + abs(a[2])):
            return i
    return n


def batch_statistics(gpu, ref, tol=TOL):
    """criteria B and C on whole-batch results (dicts with xs, us, cost, iter, status)"""
    B = ref["iter"].shape[0]
    same = gpu["iter"] == ref["iter"]
    ex = np.abs(gpu["xs"] - ref["xs"]).reshape(B, -1).max(axis=1)
    eu = np.abs(gpu["us"] - ref["us"]).reshape(B, -1).max(axis=1)
    ec = np.abs(gpu["cost"] - ref["cost"]) / (1 + np.abs(ref["cost"]))
    ok = same & (ex < tol) & (eu < tol)
    so, sg = solved(ref["status"], ref["cost"]), solved(gpu["status"], gpu["cost"])
    both = so & sg
    return {"rollouts": int(B), "iterations_equal": int(same.sum()), "within_tolerance_and_iterations_equal": int(ok.sum()),
            "solved_by_oracle": int(so.sum()), "solved_by_gpu": int(sg.sum()), "solved_by_both": int(both.sum()),
            "solved_by_oracle_and_within_tolerance": int((so & ok).sum()),
            "agreement_rate_among_oracle_solved": float((so & ok).sum() / max(so.sum(), 1)),
            "cost_rel_err_median_solved_by_both": float(np.median(ec[both])) if both.any() else None,
            "xs_err_max_where_within_tolerance": float(ex[ok].max()) if ok.any() else None,
            "tolerance": tol}


def sample_checks(ob, d, x0s, gpu, gpu_traces, sample, final_smooth, th_stop, maxiter=100):
    """criteria A, D, E on a sample of rollouts: the oracle follows each one alone (single thread)"""
    first_div, early_ok, cost_err, defect, dec_gpu, dec_ref = [], [], [], [], [], []
    for b in sample:
        o = ob.OracleSolver(d)
        o.set_x0(x0s[b])
        o.solve(None, None, maxiter)
        tr_ref, r = o.trace(), o.result()
        tr_gpu = gpu_traces[b]
        fd = first_divergence(tr_gpu, tr_ref)
        first_div.append(fd)
        early_ok.append(fd >= min(EARLY_K, len(tr_ref), len(tr_gpu)))
        g_conv = bool(solved(np.array([gpu["status"][b]]), np.array([gpu["cost"][b]]))[0])
        if g_conv:
            # D: the oracle's view of the GPU's final point
            o2 = ob.OracleSolver(d)
            o2.set_x0(x0s[b])
            o2.set_smooth(final_smooth)
            c, fs, _ = o2.phase_calcdiff(gpu["xs"][b], gpu["us"][b])
            cost_err.append(abs(c - gpu["cost"][b]) / (1 + abs(c)))
            defect.append(float(np.abs(fs).max()))
            # E: expected reduction of one more full step from there
            ok, _, _, _, _, dgdq = o2.phase_backward(1e-9)
            if ok:
                dec_gpu.append(abs(dgdq[0] + 0.5 * dgdq[1]))
        if bool(solved(np.array([r["status"]]), np.array([r["cost"]]))[0]):
            o3 = ob.OracleSolver(d)
            o3.set_x0(x0s[b])
            o3.set_smooth(final_smooth)
            o3.phase_calcdiff(r["xs"], r["us"])
            ok, _, _, _, _, dgdq = o3.phase_backward(1e-9)
            if ok:
                dec_ref.append(abs(dgdq[0] + 0.5 * dgdq[1]))
    bound_e = max(10 * th_stop, 3 * max(dec_ref)) if dec_ref else 10 * th_stop
    return {"sample": [int(b) for b in sample], "early_path_ok": int(sum(early_ok)), "early_k": EARLY_K,
            "first_divergent_iteration_min_median": [int(min(first_div)), float(np.median(first_div))],
            "converged_on_gpu_in_sample": len(cost_err),
            "oracle_cost_at_gpu_point_rel_err_max": float(max(cost_err)) if cost_err else None,
            "oracle_dynamics_defect_at_gpu_point_max": float(max(defect)) if defect else None,
            "expected_reduction_next_step_gpu_max": float(max(dec_gpu)) if dec_gpu else None,
            "expected_reduction_next_step_oracle_max": float(max(dec_ref)) if dec_ref else None,
            "expected_reduction_bound": float(bound_e)}


# ---- the parity contract of the north star, in the one form that survives a rounding-level change of either side ----------
# north_star: "matching reference trajectory within 1e-4 on xs/us".  A plain solve of the contact problem stops on a flat stretch
# (stop rule |delta cost| < 1e-3) some 0.09 in xs away from the minimiser; where on that stretch it ends is decided by rounding:
# the oracle against its own FMA-contracted build is 1.6e-4 apart on the unperturbed eagle_catch rollout, the GPU 4e-5 ... 7e-5
# (profiles/r05_margin_profile_cpu.json).  The contract therefore reads
#   (1) same minimiser: restarted from the device's final point with the stop threshold at `tight`, device and oracle end within
#       TOL = 1e-4 of each other in xs and squashed controls (measured 5e-13),
#   (2) plain solve: cost within COST_RTOL = 1e-5 relative of the oracle's,
#   (3) plain solve: identical iteration count,
# all three ASSERTED; the plain-solve xs / us distance is REPORTED with a TRIPWIRE of 2e-4 and the oracle-vs-FMA yardstick beside it.
COST_RTOL = 1e-5
PLAIN_TRIPWIRE = 2e-4


def north_star_contract(empc, ob, sw, problem, gpu_xs, gpu_us, gpu_cost, gpu_iter, x0=None, maxiter=100, tight=1e-9, backend=None):
    """the three-part check on ONE rollout (device results of a plain solve from `x0`, default the file's state); returns a dict
    with every number and `passed`; raises nothing (callers assert on `passed` / `failures`)"""
    d = problem.desc
    x0 = np.asarray(problem.x0 if x0 is None else x0, dtype=np.float64)
    ref = ob.solve_batch(d, np.array([x0]), maxiter, nthreads=1)
    fma = ob.solve_batch(d, np.array([x0]), maxiter, nthreads=1, variant="fma")
    ex = float(np.abs(gpu_xs - ref["xs"][0]).max())
    eu = float(np.abs(gpu_us - ref["us"][0]).max())
    crel = float(abs(gpu_cost - ref["cost"][0]) / (1.0 + abs(ref["cost"][0])))
    prm = ob.default_params()
    make = backend if backend is not None else (lambda n, p2: sw.GpuBackend(empc, problem, p2, n))
    same = sw.same_minimum(make, d, prm, np.array([x0]), np.array([gpu_xs]), np.array([gpu_us]), tight=tight)[0]
    out = {"tolerance": TOL, "cost_rtol": COST_RTOL, "plain_tripwire": PLAIN_TRIPWIRE,
           "restart_xs_err": float(same["xs_err"]), "restart_us_squashed_err": float(same["usq_err"]),
           "restart_iterations_device_oracle": [int(same["iters_device"]), int(same["iters_oracle"])], "restart_moved": float(same["moved"]),
           "plain_cost_rel_err": crel, "plain_iterations_device_oracle": [int(gpu_iter), int(ref["iter"][0])],
           "plain_xs_err_reported": ex, "plain_us_err_reported": eu,
           "yardstick_oracle_vs_its_fma_build": {"xs": float(np.abs(fma["xs"] - ref["xs"]).max()), "us": float(np.abs(fma["us"] - ref["us"]).max()),
                                                 "iterations": [int(ref["iter"][0]), int(fma["iter"][0])]}}
    failures = []
    if not (out["restart_xs_err"] < TOL and out["restart_us_squashed_err"] < TOL):
        failures.append("different minimisers from a common restart")
    if not crel <= COST_RTOL:
        failures.append("plain-solve cost differs by %.2e relative" % crel)
    if int(gpu_iter) != int(ref["iter"][0]):
        failures.append("plain-solve iteration counts differ")
    if not (ex < PLAIN_TRIPWIRE and eu < PLAIN_TRIPWIRE):
        failures.append("plain-solve xs / us beyond the 2e-4 tripwire (%.2e / %.2e)" % (ex, eu))
    out["failures"] = failures
    out["passed"] = not failures
    return out
