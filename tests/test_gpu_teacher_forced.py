"""Teacher-forced parity on the GPU, through the C ABI (empc_solver_set_states / empc_sweep_batch / empc_select_batch):
the decisive form of the north-star parity claim on the workloads whose free-running iteration paths are
rounding-sensitive (perturbed eagle_catch, the ContactModel6D variant, the box solvers' cold starts, RK4 nodes, two iris
files).  Driver and argument: tests/stepwise.py.  For every rollout of the sample

  1. the GPU reproduces EVERY iteration of the oracle's own path from the oracle's iterate -- tape 1e-9, gains 1e-6, the
     cost of every step length 1e-9 (or 10x the distance between the oracle's own two builds on that trial, where a
     near-unstable rollout amplifies rounding), and the accepted step, regularisation, feasibility and stop decision EXACTLY;
  2. the oracle reproduces every iteration of the GPU's own free-running path from the GPU's iterate;
  3. restarted from the GPU's final point with the convergence threshold at 1e-9 both reach the same minimiser: the
     north-star bound 1e-4 on xs / us is asserted there (measured: 1e-11).

1 + 2: wherever the two free-running paths part ways, each side's decision is the other's on the same inputs, so the paths
differ by accumulated rounding, not by a different rule.  Reference: src/sbfddp.cpp:192-393 on the shipped YAMLs."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import stepwise as sw
from conftest import CONFIGS, contact_variant

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")


def factory(empc, problem, prm, cls=None):
    return lambda n, p2: sw.GpuBackend(empc, problem, p2 if p2 is not None else prm, n, cls)


def save(name, rep):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "r05_stepwise_%s.json" % name), "w") as f:
            json.dump(rep, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
        print("stepwise %s: waived %.3f (free run %.3f), accepted chaotic trials %d of which prefix-checked %d" % (
            name, rep.get("waived_fraction", -1), rep["free_run"].get("waived_fraction", -1), rep.get("accepted_trials_chaotic", 0),
            rep.get("accepted_chaotic_prefix_checked", 0)))
    except OSError:
        pass


def waived_fraction(rep):
    """Share of the teacher-forced iterations that carry NO numerical assertion on the step they took: decisions excused as
    chaotic / tied / direction ties, iterates skipped as exploded, and accepted chaotic trials for which not even a knot prefix
    could be compared.  An accepted chaotic trial whose candidate was compared knot by knot up to the split of the oracle's own
    variants (tests/stepwise.py, PREFIX_SPLIT) counts as checked."""
    waived = (rep.get("decisions_excused_chaotic", 0) + rep.get("decisions_excused_tied", 0) + rep.get("direction_ties_excused", 0) +
              rep.get("iterates_skipped_exploded", 0) + rep.get("accepted_chaotic_no_prefix", 0) +
              rep.get("accepted_chaotic_variants_disagree_on_step", 0))
    return waived / max(rep["decisions_checked"], 1)


def check(rep, max_waived=0.10, min_asserted=0, *, max_exploded):
    """every pair went through the comparison, nothing unexplained in the reverse direction, and the waivers stay a bounded
    minority (a collapse of the harness into 'everything excused' fails here); `min_asserted`: an absolute floor on the
    iterations that DID carry the numerical assertions (a loose relative bound alone lets a test pass on a handful)"""
    assert rep["decisions_checked"] == rep["pairs"] > 0
    asserted = rep["decisions_checked"] * (1.0 - waived_fraction(rep))
    rep["decisions_asserted"] = asserted
    assert asserted >= min_asserted, (asserted, min_asserted)
    # (the measured count of the last hardware run + ~20 %; mandatory: exploded iterates are compared for nothing but finiteness)
    assert rep.get("iterates_skipped_exploded", 0) <= max_exploded, (rep.get("iterates_skipped_exploded"), max_exploded)
    assert rep["free_run"]["unexplained"] == 0
    rep["waived_fraction"] = waived_fraction(rep)
    fr = rep["free_run"]
    rep["free_run"]["waived_fraction"] = (fr["excused_chaotic_or_tied"] + fr["skipped_exploded_iterates"]) / max(fr["device_iterations"], 1)
    assert rep["waived_fraction"] <= max_waived, (rep["waived_fraction"], max_waived)


def test_eagle_catch_perturbed_64(empc, problems):
    """the north-star workload: 64 perturbed rollouts, every iterate (~3 500 iterations)"""
    _, problem = problems["eagle_catch"]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 64, nq=d.model.nq)
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, chunk=1024, tape_every=11)
    check(rep, max_waived=0.08, min_asserted=3000, max_exploded=120)  # (measured r04: 3 513 iterates, 93 skipped as exploded, waived 0.027)
    save("eagle_catch_64", rep)
    assert rep["pairs"] > 2000 and rep["tapes_checked"] > 250


def test_select_alone_on_gpu(empc, problems):
    """empc_select_batch: the decision stage fed with the oracle's numbers returns the oracle's decision bit for bit"""
    _, problem = problems["eagle_catch"]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 6, nq=d.model.nq)
    paths = sw.oracle_paths(d, prm, x0s)
    pairs = [(b, i) for b in range(len(x0s)) for i in range(len(paths[b]["iterates"]))]
    n = sw.select_in_isolation(lambda k: sw.GpuBackend(empc, problem, prm, k), d, prm, x0s, paths, pairs)
    assert n == len(pairs) > 100


@pytest.mark.parametrize("contact,gains", [("ContactModel6D", (0.0, 0.0)), ("ContactModel6D", (11.0, 5.0)), ("ContactModel3D", (9.0, 4.0))])
def test_contact_options(empc, tmp_path, contact, gains):
    _, problem = contact_variant(empc, tmp_path, contact, gains)
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 6, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, tape_every=41)
    try:
        # (six contact rows on a 9-dof arm are poorly conditioned: a quarter of the ContactModel6D iterations carry trials the
        #  oracle's own builds disagree on -- measured 0.26 / 0.16 / 0.00)
        # bounds = the fractions measured on hardware in round 4 (profiles/r04_stepwise/) + 0.05
        bound, blown = {("ContactModel6D", 0.0): (0.32, 125), ("ContactModel6D", 11.0): (0.21, 110), ("ContactModel3D", 9.0): (0.05, 10)}[(contact, gains[0])]
        check(rep, max_waived=bound, max_exploded=blown)
    finally:
        save("contact_%s_%g" % (contact, gains[0]), rep)


@pytest.mark.parametrize("name,dt,solver_type", [("hover", 40, 1), ("hover", 40, 2), ("eagle_catch", 32, 1), ("eagle_catch", 32, 2), ("displacement", 80, 2)])
def test_box_solvers(empc, name, dt, solver_type):
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = tr.createProblem(dt, False, "IntegratedActionModelEuler")
    d = problem.desc
    prm = ob.default_params()
    prm.solver_type = solver_type
    cls = {1: empc.SolverBoxFDDP, 2: empc.SolverBoxDDP}[solver_type]
    x0s = empc.perturbed_x0s(problem.x0, 8, nq=d.model.nq)
    rep = sw.stepwise_parity(factory(empc, problem, prm, cls), d, prm, x0s, maxiter=30, tape_every=29, do_same_minimum=False)
    try:
        # (cold starts of the box solvers: a third of the accepted steps are rollouts the oracle's own builds differ on by more
        #  than 1e-4 and, more often than not, accept different step lengths in its own variants -- only a minority has a
        #  comparable knot prefix; measured 0.36 / 0.31 / 0.11 / 0.28 / 0.00)
        # bounds = measured on hardware in round 4 (profiles/r04_stepwise/r04_stepwise_box_*.json) + 0.05
        bound, blown = {("hover", 1): (0.41, 50), ("hover", 2): (0.36, 25), ("eagle_catch", 1): (0.17, 10), ("eagle_catch", 2): (0.34, 10),
                        ("displacement", 2): (0.05, 5)}[(name, solver_type)]
        check(rep, max_waived=bound, max_exploded=blown)
    finally:
        save("box_%s_%d" % (name, solver_type), rep)


@pytest.mark.parametrize("name", ["eagle_catch", "displacement"])
def test_rk4_nodes(empc, problems, name):
    tr, _ = problems[name]
    problem = tr.createProblem(CONFIGS[name][1], True, "IntegratedActionModelRK4")
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 4, nq=d.model.nq, amplitude=0.02)
    # (tape at 1e-8: the RK4 node's Lu is a sum of four stage terms on Hessians of 1e9; measured 5e-9)
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, tape_every=31, tol_tape=1e-8)
    try:
        check(rep, max_waived=0.05, max_exploded=5)  # (measured r04: 0.000 on 333 / 64 iterations)
    finally:
        save("rk4_" + name, rep)


@pytest.mark.parametrize("rel", ["iris/trajectories/loop.yaml", "iris_px4/trajectories/hover.yaml", "hexacopter370/trajectories/hover.yaml"])
def test_long_running_shipped_files(empc, rel):
    """the shipped files whose free-running paths are rounding-sensitive from their own initial state (35 vs 199 iterations
    between the oracle's two builds on iris/loop), and the perturbed hover of BASELINE configs[0]"""
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(rel))
    try:
        problem = t.createProblem()
    except empc.EmpcError:
        problem = t.createProblem(40, True, "IntegratedActionModelEuler")
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 4, nq=d.model.nq, amplitude=0.05 if "hexacopter370" in rel else 0.0)
    # tape at 1e-8: perturbed hovers pass through iterates that have blown up (states of 1e3, costs of 1e9) where Fx loses a
    # digit; common restart at 1e-12: iris_px4/hover's valley is flat (cost equal to 1e-9 still leaves xs 4e-4 apart)
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, tape_every=43, do_same_minimum="hover" in rel,
                             tol_tape=1e-8, tight=1e-12)
    try:
        # (perturbed hovers: most rollouts explode in their first iteration on both sides -- LABNOTES.md, divergence study -- and
        #  iterate on at costs of 1e13: those iterates are skipped as exploded)
        # (r04: 358 iterates on the hover, 216 of them skipped: at least 100 must carry the assertions; the same kernels on a hover
        #  whose iterates do not explode: tests/test_zz_gpu_round5.py::test_gentle_workloads_leave_nothing_waived)
        # bounds = measured on hardware in round 4 + 0.05: hover 0.603 (216 of 358 iterates exploded), iris loop 0.056, iris_px4 0.000
        bound, blown, floor = {"hexacopter370/trajectories/hover.yaml": (0.66, 260, 100), "iris/trajectories/loop.yaml": (0.11, 10, 100),
                               "iris_px4/trajectories/hover.yaml": (0.05, 5, 30)}[rel]
        check(rep, max_waived=bound, min_asserted=floor, max_exploded=blown)
    finally:
        save(rel.replace("/", "_").replace(".yaml", ""), rep)
