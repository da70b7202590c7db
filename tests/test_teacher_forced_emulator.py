"""Teacher-forced parity of the kernel bodies on the CPU lane emulator (tests/stepwise.py): every iteration of the oracle's
paths reproduced by the device code from the oracle's iterate, every iteration of the device's free-running paths
reproduced by the oracle from the device's iterate, and the same minimiser from a common restart.  The GPU edition
(tests/test_gpu_teacher_forced.py) runs the same driver through the C ABI on larger batches.
Reference: the loop bodies of SolverSbFDDP::solveFDDP / solveDDP, src/sbfddp.cpp:241-311, 329-389."""
import numpy as np
import pytest

import oracle_binding as ob
import stepwise as sw


@pytest.fixture(scope="module")
def emu():
    return sw.load_emulator()


@pytest.mark.parametrize("name,rollouts", [("displacement", 2), ("eagle_catch", 3)])
def test_stepwise_parity_on_the_emulator(empc, problems, emu, name, rollouts):
    _, problem = problems[name]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, rollouts, nq=d.model.nq)
    rep = sw.stepwise_parity(lambda n, p2: sw.EmuBackend(emu, d, p2 if p2 is not None else prm, n), d, prm, x0s, chunk=64,
                             tape_every=11, tight_maxiter=200)
    assert rep["decisions_checked"] == rep["pairs"] > 0
    assert rep["free_run"]["unexplained"] == 0
    assert rep["same_minimum"]["xs_err_max"] < 1e-8  # (far inside the north-star bound: the restart is well conditioned)
    print(name, {k: v for k, v in rep.items() if k != "max_rel"}, rep["max_rel"])


def test_select_alone_follows_the_oracle(empc, problems, emu):
    """The line-search decision in isolation (select_decide_state): fed with the ORACLE's trial costs and gap terms of real
    iterates of the contact problem, it accepts the step the oracle accepted and leaves the scalars the oracle's trace holds
    -- exactly, since no device arithmetic precedes it (src/sbfddp.cpp:260-311)."""
    _, problem = problems["eagle_catch"]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 2, nq=d.model.nq)
    paths = sw.oracle_paths(d, prm, x0s)
    pairs = [(b, i) for b in range(2) for i in range(len(paths[b]["iterates"]))]
    checked = sw.select_in_isolation(lambda n: sw.EmuBackend(emu, d, prm, n), d, prm, x0s, paths, pairs)
    assert checked == len(pairs)


@pytest.mark.parametrize("workload", ["hover_gentle", "contact6d_baumgarte_gentle"])
def test_gentle_workloads_leave_nothing_waived(empc, emu, tmp_path, workload):
    """Workloads on which NO iterate explodes (the step-wise tests of the perturbed hover and of ContactModel6D waive
    up to 70 % / 35 % of their iterations as exploded or chaotic): hexacopter370 hover from states perturbed by 0.002 and the
    ContactModel6D variant of eagle_catch with Baumgarte gains (11, 5) -- every iteration of both sides' paths carries the
    numerical assertions (measured here: waived 0.000 on 167 and 238 iterations).  The GPU edition with larger batches:
    tests/test_zz_gpu_round5.py."""
    from conftest import contact_variant
    from test_gpu_teacher_forced import check
    if workload == "hover_gentle":
        t = empc.Trajectory()
        t.autoSetup(empc.yaml_path("hexacopter370/trajectories/hover.yaml"))
        problem = t.createProblem(40, True, "IntegratedActionModelEuler")
        rollouts, kw, floor = 4, dict(tape_every=43, tol_tape=1e-8, tight=1e-12, do_same_minimum=False), 60
    else:
        _, problem = contact_variant(empc, tmp_path, "ContactModel6D", (11.0, 5.0))
        rollouts, kw, floor = 2, dict(tape_every=41, do_same_minimum=False), 150
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, rollouts, nq=d.model.nq, amplitude=0.002)
    rep = sw.stepwise_parity(lambda n, p2: sw.EmuBackend(emu, d, p2 if p2 is not None else prm, n), d, prm, x0s, chunk=64, **kw)
    check(rep, max_waived=0.05, min_asserted=floor, max_exploded=0)
    print(workload, "pairs", rep["pairs"], "waived", rep["waived_fraction"], "asserted", rep["decisions_asserted"])


def test_exploded_iterate_with_different_direction_outcome_is_set_aside(empc, problems, emu):
    """Regression of the harness (round-5 soak, seed 53): a rollout whose iterates explode (cost 4e17, joint angles of 8e7 rad)
    reaches an iterate on which the oracle's computeDirection gives up at every regularisation while the device's succeeds at
    1e3 -- tapes equal to 1e-14, the ill-conditioned recursion differs.  Such an iterate is counted as exploded, not asserted
    on; everything else of the rollout still goes through the comparison."""
    _, problem = problems["eagle_catch"]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 8, nq=d.model.nq, seed=53)[4:5]
    paths = sw.oracle_paths(d, prm, x0s)
    assert len(paths[0]["iterates"]) > 114
    # (the twelve iterates around the one in question: the whole path costs 100 s of oracle variants on the CPU)
    paths[0]["iterates"] = paths[0]["iterates"][106:118]
    rep = sw.teacher_forced(lambda n: sw.EmuBackend(emu, d, prm, n), d, prm, x0s, paths, chunk=64, tape_every=5)
    assert rep["decisions_checked"] == rep["pairs"] == len(paths[0]["iterates"]) >= 9
    assert rep.get("iterates_skipped_exploded", 0) >= 1
