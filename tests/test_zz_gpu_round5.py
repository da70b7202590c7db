"""GPU tests written in round 5 while no GPU was reachable: they have NOT run on hardware yet, so they sort last in the suite
(the driver runs `pytest -x`: a first-run surprise here must not hide the results of the tests with a history).  They run the
shipped, hardware-verified kernels only.  What is already known to hold is asserted; quantities nobody has measured yet are
recorded under gpurun_out/parity/ and reported as warnings on the first run (no bound before a measurement), and
become assertions in the commit that carries the measured value."""
import os

import numpy as np
import pytest

from test_gpu_baked import runtime_solver

pytestmark = pytest.mark.gpu


def test_unperturbed_eagle_catch_margin_profile(empc, problems):
    """The north-star rollout (eagle_catch from its YAML state, the one __graft_entry__.smoke() holds to 1e-4 against the oracle):
    where its distance to the oracle lives, per knot, for both kernel families; the families must agree to 2e-5 (VERDICT r04 item
    6: the margin shrank 4.2e-5 -> 6.6e-5 when the baked family changed the contraction; a regression shows here before smoke()
    trips).  The per-knot profile goes to gpurun_out/parity/r05_margin_profile.json (kept as profiles/r05_margin_profile.json)."""
    import json
    import oracle_binding as ob
    _, problem = problems["eagle_catch"]
    d = problem.desc
    a = empc.SolverSbFDDP(problem, batch=1)
    b = runtime_solver(empc, problem, 1)
    assert a.kernel_family.startswith("baked") and b.kernel_family == "runtime model"
    a.solve([], [], 100)
    b.solve([], [], 100)
    ref = {}
    for variant in (None, "fma"):
        o = ob.OracleSolver(d, ob.default_params(), variant=variant)
        o.solve(None, None, 100)
        ref[variant or "plain"] = o.result()
    r = ref["plain"]
    xa, xb = a.xs_batch[0], b.xs_batch[0]
    ua, ub = a.us_batch[0], b.us_batch[0]
    rep = {"iterations": {"baked": int(a.iter_batch[0]), "runtime": int(b.iter_batch[0]), "oracle": int(r["iter"]), "oracle_fma": int(ref["fma"]["iter"])},
           "cost": {"baked": float(a.cost_batch[0]), "runtime": float(b.cost_batch[0]), "oracle": float(r["cost"]), "oracle_fma": float(ref["fma"]["cost"])},
           "max_abs_err_xs": {"baked_vs_oracle": float(np.abs(xa - r["xs"]).max()), "runtime_vs_oracle": float(np.abs(xb - r["xs"]).max()),
                              "baked_vs_runtime": float(np.abs(xa - xb).max()), "oracle_fma_vs_oracle": float(np.abs(ref["fma"]["xs"] - r["xs"]).max())},
           "max_abs_err_us": {"baked_vs_oracle": float(np.abs(ua - r["us"]).max()), "runtime_vs_oracle": float(np.abs(ub - r["us"]).max()),
                              "baked_vs_runtime": float(np.abs(ua - ub).max()), "oracle_fma_vs_oracle": float(np.abs(ref["fma"]["us"] - r["us"]).max())},
           "per_knot_max_abs_err_xs": {"baked_vs_oracle": np.abs(xa - r["xs"]).max(axis=1).tolist(), "runtime_vs_oracle": np.abs(xb - r["xs"]).max(axis=1).tolist(),
                                       "oracle_fma_vs_oracle": np.abs(ref["fma"]["xs"] - r["xs"]).max(axis=1).tolist()},
           "per_knot_max_abs_err_us": {"baked_vs_oracle": np.abs(ua - r["us"]).max(axis=1).tolist(), "runtime_vs_oracle": np.abs(ub - r["us"]).max(axis=1).tolist(),
                                       "oracle_fma_vs_oracle": np.abs(ref["fma"]["us"] - r["us"]).max(axis=1).tolist()},
           "per_state_component_max_abs_err_baked_vs_oracle": np.abs(xa - r["xs"]).max(axis=0).tolist(),
           "per_control_component_max_abs_err_baked_vs_oracle": np.abs(ua - r["us"]).max(axis=0).tolist()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    json.dump(rep, open(os.path.join(out, "r05_margin_profile.json"), "w"), indent=1)
    print({k: rep[k] for k in ("iterations", "cost", "max_abs_err_xs", "max_abs_err_us")})
    # hard: what smoke() and the golden vector already hold for the shipped (baked) family
    assert rep["iterations"]["baked"] == rep["iterations"]["oracle"]
    # (plain-solve end points: tripwire only -- the oracle's own FMA build lies 1.6e-4 from it on this rollout; the asserted form
    #  of the north-star bound is the common restart, tests/parity_criteria.py north_star_contract, run by smoke() and
    #  tests/test_gpu_eagle_catch.py)
    assert rep["max_abs_err_xs"]["baked_vs_oracle"] < 2e-4 and rep["max_abs_err_us"]["baked_vs_oracle"] < 2e-4
    # a flat stretch, not drift: the costs agree to 1e-5 relative (the oracle against its own FMA build: 2e-6) while the end points
    # differ by up to 1e-4 (profiles/r05_margin_profile_cpu.json)
    assert abs(rep["cost"]["baked"] - rep["cost"]["oracle"]) < 1e-5 * (1 + abs(rep["cost"]["oracle"]))
    # not yet measured on hardware (VERDICT r04 item 6 asks for 2e-5 between the families): recorded and reported on the first run,
    # asserted from the commit that carries the measured value
    unmeasured = {"families_same_iterations": rep["iterations"]["baked"] == rep["iterations"]["runtime"],
                  "families_xs_within_2e-5": rep["max_abs_err_xs"]["baked_vs_runtime"] <= 2e-5,
                  "families_us_within_2e-5": rep["max_abs_err_us"]["baked_vs_runtime"] <= 2e-5}
    rep["first_run_checks"] = unmeasured
    json.dump(rep, open(os.path.join(out, "r05_margin_profile.json"), "w"), indent=1)
    if not all(unmeasured.values()):
        import warnings
        warnings.warn("first-run checks not met (recorded, not asserted yet): %s" % unmeasured)


@pytest.mark.parametrize("name", ["displacement", "eagle_catch"])
def test_families_take_the_same_exits_on_poisoned_inputs(empc, problems, name):
    """The baked units are compiled with -fno-honor-nans; the solver's divergence handling must not depend on it: initial states
    carrying NaN, inf and 1e200 (and healthy neighbours in the same batch) give the same status, iteration count and result rows
    from the baked and the runtime-model kernels -- NaN for NaN, bit for bit elsewhere on the rows that blew up at once."""
    _, problem = problems[name]
    d = problem.desc
    B = 8
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq, seed=21)
    x0s[1, 0] = np.nan
    x0s[2, d.model.nq + 1] = np.inf
    x0s[3, 2] = 1e200
    x0s[4, d.model.nq:] = 1e6      # absurd velocities: finite garbage for a few knots, then overflow
    x0s[5, 7 if d.nx > 13 else 0] = -1e300
    a = empc.SolverSbFDDP(problem, batch=B)
    b = runtime_solver(empc, problem, B)
    assert a.kernel_family.startswith("baked") and b.kernel_family == "runtime model"
    a.solve([], [], 30, x0s=x0s)
    b.solve([], [], 30, x0s=x0s)
    poisoned = [1, 2, 3, 4, 5]
    for i in (1, 2):  # hard: a NaN / inf initial state never reports convergence, in either family
        assert (a.status_batch[i] & 1) == 0 and (b.status_batch[i] & 1) == 0, (i, a.status_batch[i], b.status_batch[i])
    # identical exits of the two families: never measured on hardware -- recorded and reported on the first run, asserted from the
    # commit that carries the measurement (a difference here is the finding ADVICE r04 warns about, not a flaky test)
    same = {"status": bool(np.array_equal(a.status_batch[poisoned], b.status_batch[poisoned])),
            "iterations": bool(np.array_equal(a.iter_batch[poisoned], b.iter_batch[poisoned])),
            "finite_pattern": all(bool(np.array_equal(np.isfinite(a.xs_batch[i]), np.isfinite(b.xs_batch[i]))) for i in poisoned)}
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    json.dump({"workload": name, "baked_status": a.status_batch.tolist(), "runtime_status": b.status_batch.tolist(),
               "baked_iterations": a.iter_batch.tolist(), "runtime_iterations": b.iter_batch.tolist(), "same": same},
              open(os.path.join(out, "r05_poisoned_inputs_%s.json" % name), "w"), indent=1)
    if not all(same.values()):
        import warnings
        warnings.warn("first-run check not met (recorded, not asserted yet): families differ on poisoned inputs: %s" % same)
    # the healthy rollouts of the same batch are untouched by their neighbours
    clean = empc.SolverSbFDDP(problem, batch=B)
    x0c = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq, seed=21)
    clean.solve([], [], 30, x0s=x0c)
    for i in (0, 6, 7):
        assert np.array_equal(clean.xs_batch[i], a.xs_batch[i]) and clean.iter_batch[i] == a.iter_batch[i]


@pytest.mark.parametrize("workload", ["hover_gentle", "contact6d_baumgarte_gentle"])
def test_gentle_workloads_leave_nothing_waived(empc, tmp_path, workload):
    """GPU edition of tests/test_teacher_forced_emulator.py::test_gentle_workloads_leave_nothing_waived: step-wise parity on
    workloads whose iterates do not explode, so that (nearly) every iteration carries the numerical assertions -- the loose
    waiver bounds of the perturbed hover (0.70) and of ContactModel6D (0.35) in tests/test_gpu_teacher_forced.py are no longer the
    only step-wise evidence for those kernels."""
    import oracle_binding as ob
    import stepwise as sw
    from conftest import contact_variant
    from test_gpu_teacher_forced import check, factory, save
    if workload == "hover_gentle":
        t = empc.Trajectory()
        t.autoSetup(empc.yaml_path("hexacopter370/trajectories/hover.yaml"))
        problem = t.createProblem(40, True, "IntegratedActionModelEuler")
        rollouts, kw, floor = 16, dict(tape_every=43, tol_tape=1e-8, tight=1e-12), 250
    else:
        _, problem = contact_variant(empc, tmp_path, "ContactModel6D", (11.0, 5.0))
        rollouts, kw, floor = 6, dict(tape_every=41), 400
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, rollouts, nq=d.model.nq, amplitude=0.002)
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, **kw)
    try:
        check(rep, max_waived=0.10, min_asserted=floor, max_exploded=3)  # (emulator: 0 exploded)
    finally:
        save("gentle_" + workload, rep)
