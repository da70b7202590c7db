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
