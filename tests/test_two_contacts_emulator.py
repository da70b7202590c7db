"""Several contacts per stage: src/stage.cpp:38-48 adds EVERY name of a stage's `contacts` list to one
ContactModelMultiple, whose rows crocoddyl stacks in the order of its name-sorted map into one KKT system.  No shipped YAML lists
more than one; the factory accepts any number.  Here: eagle_catch with a second ContactModel3D ("elbow", on link 2) in its grasp
stage -- six stacked rows --
  * the C++ oracle against the NumPy second restatement (dense stacked KKT, complex-step derivatives) at 1e-10, Euler and RK4 nodes,
    Baumgarte gains, the friction cone reading the force of the contact on ITS frame;
  * the kernel bodies of the CT_PAIR3 instantiation (csrc/empc_dev_model.hpp contact_forward_pair3, the pair paths of
    empc_linearize2.hpp / empc_rollout6.hpp / empc_prep.hpp) on the CPU lane emulator against the oracle: tape 1e-9, gains 1e-6,
    trial rollouts 1e-7, the per-lane nominal evaluation 1e-9 -- the tolerances of the single-contact classes.
The GPU edition is tests/test_zz_gpu_two_contacts.py (opt-in: these kernels have never run on hardware)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
from conftest import arm5_two_contact_variant, two_contact_variant
from test_emulator_parity import emu, emu_baked, families_equal, kernel_bodies, rel  # noqa: F401  (emu, emu_baked: fixtures that build / bind the emulators)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
KEYS = ("xnext", "cost", "acc", "lam", "u_squash", "Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu")

# Tape tolerance: the parity contract's 1e-9 (tests/test_gpu_parity.py).  The single-contact emulator tests hold 1e-11; the six-row
# KKT system of a random candidate is worse conditioned (measured: 1.4e-11 on Fx of the first grasp knot).
TAPE_TOL = 1e-9

# (gains of "end_effector", gains of "elbow", friction cone on the second contact's frame)
OPTIONS = [((0.0, 0.0), (0.0, 0.0), False), ((3.0, 1.5), (2.0, 0.7), True)]


def gain_yardstick(problem):
    """Tolerances of the backward pass for kernel_bodies: the usual 1e-6 / 1e-8, widened to three times what the oracle's OWN
    FMA-contracted build (liboracle_fma.so: the same source, another correct compiler) differs from the oracle by on the same
    candidate.  With the friction cone on the elbow contact the barrier is inactive on this candidate, Luu is 1e2 beside a Vxx of
    1e11, and Quu^-1 turns the tape's last bits into 2.4e-5 of K between the two oracle builds (measured; the emulator: 2.6e-5)."""
    from test_emulator_parity import candidate
    d = problem.desc
    res = []
    for variant in (None, "fma"):
        o = ob.OracleSolver(d, variant=variant)
        xs, us = candidate(d, 3)
        o.set_smooth(0.1)
        o.phase_calcdiff(xs, us)
        ok, K, k, Vx, _, _ = o.phase_backward(1e-9)
        assert ok
        res.append((K, k, Vx))
    yk = max(rel(res[0][0], res[1][0]), rel(res[0][1], res[1][1]))
    yv = rel(res[0][2], res[1][2])
    return dict(gain_tol=max(1e-6, 3 * yk), vx_tol=max(1e-8, 3 * yv))


def random_node(problem, d, rng):
    x = np.array(problem.x0)
    x[:3] += rng.normal(size=3) * 0.2
    q = x[3:7] + rng.normal(size=4) * 0.3
    x[3:7] = q / np.linalg.norm(q)
    x[7:] += rng.normal(size=d.nx - 7) * 0.4
    u = rng.uniform(1, 8, size=d.nu)
    u[d.n_rotors:] = rng.normal(size=d.nu - d.n_rotors) * 0.3
    return x, u


@pytest.mark.parametrize("integrator", ["IntegratedActionModelEuler", "IntegratedActionModelRK4"])
@pytest.mark.parametrize("gains,gains2,cone2", OPTIONS)
def test_oracle_two_contacts_vs_second_restatement(empc, tmp_path, integrator, gains, gains2, cone2):
    import numpy_restatement as nr
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", gains, gains2, integrator=integrator, cone_on_second=cone2)
    d = problem.desc
    prm = ob.default_params()
    smooth = 0.07
    P = nr.Problem(d, prm)
    sets = nr.cost_sets_of(d, prm, smooth)
    o = ob.OracleSolver(d, prm)
    o.set_smooth(smooth)
    knots = [t for t in range(d.T + 1) if len(sets[d.knot_set[t]]["contacts"]) == 2]
    assert len(knots) >= 3
    # the stacking order is the name-sorted map's: "elbow" (link 2) before "end_effector" (gripper)
    f0, f1 = (c["frame"] for c in sets[d.knot_set[knots[0]]]["contacts"])
    assert f0 != f1
    rng = np.random.default_rng(5)
    for t in knots[:2]:
        x, u = random_node(problem, d, rng)
        mine = nr.node(P, sets[d.knot_set[t]], x, u, smooth)
        ref = o.node_calc(t, x, u)
        assert np.abs(ref["lam"][:6]).min() > 1e-6, "both contacts must carry a force"
        for key in KEYS:
            got = ref[key][:6] if key == "lam" else ref[key]
            want = mine[key][:6] if key == "lam" else mine[key]
            assert rel(np.ravel(got), np.ravel(np.asarray(want, dtype=float))) < 1e-10, (t, key)


def test_oracle_refuses_nothing_it_cannot_solve(empc, tmp_path):
    """6D + 3D on the 9-dof arm: Jc M^-1 Jc^T is singular (nine rows on a tree with nine degrees of freedom, one body pinned twice);
    the second restatement's dense solve says so -- the reason the kernels take two ContactModel3D only."""
    import numpy_restatement as nr
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel6D")
    d = problem.desc
    prm = ob.default_params()
    P = nr.Problem(d, prm)
    sets = nr.cost_sets_of(d, prm, 0.07)
    t = [t for t in range(d.T + 1) if len(sets[d.knot_set[t]]["contacts"]) == 2][0]
    x, u = random_node(problem, d, np.random.default_rng(1))
    with pytest.raises(np.linalg.LinAlgError):
        nr.node(P, sets[d.knot_set[t]], x, u, 0.07)


@pytest.mark.parametrize("gains,gains2,cone2", OPTIONS)
def test_two_contact_kernel_bodies_vs_oracle(empc, emu, tmp_path, gains, gains2, cone2):
    """linearize (six-row body on the grasp knots, three-row / lean bodies elsewhere), backward, role-split rollout of the CT_PAIR3
    instantiation against the oracle"""
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", gains, gains2, cone_on_second=cone2)
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6, tape_tol=TAPE_TOL, **gain_yardstick(problem))


def test_two_contact_per_lane_forms_vs_oracle(empc, emu, tmp_path):
    """the per-lane rollout (k_rollout, > 16 step lengths) and the nominal node evaluation (calc kernel, RK4 stage kernel) of
    the same instantiation"""
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", (3.0, 1.5), (2.0, 0.7), cone_on_second=True)
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 1, tape_tol=TAPE_TOL, **gain_yardstick(problem))
    d = problem.desc
    prm = ob.default_params()
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    o = ob.OracleSolver(d, prm)
    o.set_smooth(0.07)
    rng = np.random.default_rng(9)
    pair_knots = [44, 45]  # grasp stage (asserted to hold two contacts by test_oracle_two_contacts_vs_second_restatement)
    for t in pair_knots[:2] + [10]:
        x, u = random_node(problem, d, rng)
        r = o.node_calc(t, x, u, diff=False)
        xn, acc, cost, usq, lam = np.zeros(d.nx), np.zeros(d.model.nv), np.zeros(1), np.zeros(d.nu), np.zeros(6)
        emu.emu_node_nominal(e, t, ob.P(x), ob.P(u), 0.07, ob.P(xn), ob.P(acc), ob.P(cost), ob.P(usq), ob.P(lam))
        assert np.abs(xn - r["xnext"]).max() < 1e-9 and np.abs(acc - r["acc"]).max() < 1e-9 * (1 + np.abs(r["acc"]).max())
        assert abs(cost[0] - r["cost"]) < 1e-9 * (1 + abs(r["cost"]))
        assert np.abs(lam - r["lam"][:6]).max() < 1e-9 * (1 + np.abs(r["lam"]).max())
    emu.emu_destroy(e)


def test_two_contact_rk4_kernel_bodies_vs_oracle(empc, emu, tmp_path):
    """IntegratedActionModelRK4 nodes over the two-contact differential model: stage kernel, raw records of the stage batch
    through the six-row linearize body, chain rule, four-stage rollout"""
    from test_emulator_parity import kernel_bodies_rk4
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", integrator="IntegratedActionModelRK4")
    # candidate seed 4: on seed 3 (the other tests') the first grasp knot sits next to a singular Jc M^-1 Jc^T -- Fx entries of 9e7
    # in the oracle, and the same absolute noise (1e-4) on Fu entries of order 1: conditioning of the candidate, not of a kernel
    kernel_bodies_rk4(emu, problem, seed=4)


def test_two_contact_stepwise_parity_on_the_emulator(empc, tmp_path):
    """The step-wise argument (tests/stepwise.py) on a two-contact problem that converges (second contact on link 1, no gains, the
    arm bent in the initial state -- the file's stretched arm is a singular configuration of ANY two point contacts, see
    conftest.two_contact_variant; 27 iterations from there): every iteration of the oracle's paths reproduced by the CT_PAIR3 kernel bodies from the
    oracle's iterate (tape 1e-9, gains 1e-6, trial costs 1e-9, every decision exactly), every iteration of the device's own paths
    reproduced by the oracle, the same minimiser from a common restart."""
    import stepwise as sw
    from test_gpu_teacher_forced import check
    emu_sw = sw.load_emulator()
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", link2="flying_arm_3__link_1", bent=(0.4, -0.7, 0.5))
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 2, nq=d.model.nq, amplitude=0.002)
    rep = sw.stepwise_parity(lambda n, p2: sw.EmuBackend(emu_sw, d, p2 if p2 is not None else prm, n), d, prm, x0s, chunk=64,
                             tape_every=13, tight=1e-6, tight_maxiter=300)
    check(rep, max_waived=0.10, min_asserted=80, max_exploded=0)  # (measured: 103 pairs, waived 0.02-0.04, 99-101 asserted)
    assert rep["same_minimum"]["converged_on_oracle"] >= 1 and rep["same_minimum"]["xs_err_max"] < 1e-6  # (measured 2.0e-8)
    print("two contacts: pairs", rep["pairs"], "waived", rep["waived_fraction"], "asserted", rep["decisions_asserted"], rep["same_minimum"], rep["max_rel"])


def test_two_contact_kernel_bodies_on_the_11_dof_class(empc, emu, tmp_path):
    """the (6, 6) robot class (hextilt_flying_arm_5): 64-lane linearize units, six-row body on the knots of the appended stage"""
    _, problem = arm5_two_contact_variant(empc, tmp_path, (2.0, 1.0), (0.0, 3.0))
    d = problem.desc
    o = ob.OracleSolver(d)
    o.set_smooth(0.07)
    x, u = random_node(problem, d, np.random.default_rng(2))
    r = o.node_calc(d.T - 3, x, u, diff=False)
    assert np.abs(r["lam"][:6]).min() > 1e-6, "both contacts must carry a force"
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6, tape_tol=TAPE_TOL, **gain_yardstick(problem))


def test_factory_answer_for_two_contact_problems(empc, tmp_path, monkeypatch):
    """empc_solver_supported: a two-ContactModel3D stage is refused with the reason until the opt-in is set (the six-row kernels
    have never run on hardware) and accepted with it, on both arm classes; a 6D + 3D pair is refused whatever the switch says,
    with the reason (six rows are what the kernels and the robots have room for)."""
    _, p3 = two_contact_variant(empc, tmp_path, "ContactModel3D")
    _, p5 = arm5_two_contact_variant(empc, tmp_path)
    _, p6 = two_contact_variant(empc, tmp_path, "ContactModel6D")
    monkeypatch.delenv("EMPC_EXPERIMENTAL_CONTACT", raising=False)
    for p in (p3, p5):
        assert not empc.solver_supported(p) and "EMPC_EXPERIMENTAL_CONTACT" in empc.last_error() and "two contacts" in empc.last_error()
    monkeypatch.setenv("EMPC_EXPERIMENTAL_CONTACT", "1")
    for p in (p3, p5):
        assert empc.solver_supported(p), empc.last_error()
    assert not empc.solver_supported(p6) and "two ContactModel3D only" in empc.last_error()
    # the single-contact problem of the same file still takes the shipped three-row instantiation, switch or no switch
    from conftest import contact_variant
    _, p1 = contact_variant(empc, tmp_path)
    monkeypatch.delenv("EMPC_EXPERIMENTAL_CONTACT", raising=False)
    assert empc.solver_supported(p1), empc.last_error()


def test_both_contact_points_stand_still_positions_only(empc, tmp_path):
    """The constraint itself, checked through POSITIONS only (no velocity / acceleration recursion, no Jacobian): move the robot
    along q(s) = q (+) (v s + a s^2 / 2) with the oracle's contact acceleration a and take the second difference of the world
    position of both contact points -- with zero gains ContactModel3D asks for zero classical acceleration of the point, i.e. a
    world-frame p'' = 0.  With the acceleration of the ONE-contact problem (end effector only) the elbow point accelerates."""
    import numpy_restatement as nr
    from conftest import contact_variant
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D")
    _, single = contact_variant(empc, tmp_path)
    d = problem.desc
    prm = ob.default_params()
    P = nr.Problem(d, prm)
    sets = nr.cost_sets_of(d, prm, 0.07)
    nq, nv, t = d.model.nq, d.model.nv, 45
    x, u = random_node(problem, d, np.random.default_rng(3))

    def second_difference(a, h=1e-4):
        def points(s):
            xq = np.real(nr.state_integrate(nq, x, np.concatenate([x[nq:] * s + 0.5 * a * s * s, np.zeros(nv)])))
            R, p, _, vel, acc0 = nr.kinematics(P.md, nr._c(xq[:nq]), np.zeros(nv), np.zeros(nv), gravity=False)
            return [np.real(nr.frame_kin(P.md, ct["frame"], R, p, vel, acc0)[1]) for ct in sets[d.knot_set[t]]["contacts"]]
        pp, p0, pm = points(h), points(0.0), points(-h)
        return [np.abs(pp[k] - 2 * p0[k] + pm[k]).max() / h ** 2 for k in range(2)]

    o = ob.OracleSolver(d, prm)
    o.set_smooth(0.07)
    both = second_difference(o.node_calc(t, x, u, diff=False)["acc"])
    o1 = ob.OracleSolver(single.desc, prm)
    o1.set_smooth(0.07)
    one = second_difference(o1.node_calc(t, x, u, diff=False)["acc"])
    print("p'' of (elbow, end effector): two contacts", both, "one contact", one)
    assert max(both) < 1e-4            # (measured 1.0e-6: truncation of the difference quotient)
    assert one[1] < 1e-3 and one[0] > 1.0  # (measured 1.5e-4 and 49.7) the end effector is held in both problems, the elbow only in the two-contact one


def test_three_frames_need_a_build_with_three_captures(empc, emu, tmp_path):
    """Two contact frames AND a frame cost on a third link: three distinct frames in one stage.  The product's kernels capture two
    (NCAP, every shipped stage names at most two) and the factory refuses the problem with that reason; a library built with
    -DEMPC_NCAP=3 takes it.  On the emulator of such a build (EMU_MACROS="EMPC_NCAP=3"): phase parity against the oracle."""
    import os
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", (0.0, 2.0), (0.0, 2.0), extra_frame_cost="hexacopter370__base_link")
    if "EMPC_NCAP=3" not in os.environ.get("EMU_MACROS", "").split():
        os.environ["EMPC_EXPERIMENTAL_CONTACT"] = "1"
        try:
            assert not empc.solver_supported(problem) and "more distinct frames" in empc.last_error()
        finally:
            del os.environ["EMPC_EXPERIMENTAL_CONTACT"]
        pytest.skip("three capture slots: run with EMU_MACROS=EMPC_NCAP=3 (tools/variant_verdicts.py does)")
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6, tape_tol=TAPE_TOL, **gain_yardstick(problem))
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 1, tape_tol=TAPE_TOL, **gain_yardstick(problem))


def test_host_mirror_lists_both_contacts(empc, tmp_path):
    """Stage::autoSetup (src/stage.cpp:38-48) through the host mirror: both names of the stage's `contacts` list in its
    ContactModelMultiple, in the name-sorted order crocoddyl's map would stack them, with their factory types; the flat descriptor
    handed to the kernels carries the same two entries (frames of two different links)."""
    tr, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", (3.0, 1.5), (2.0, 0.7))
    grasp = [s for s in tr.stages if s.name == "grasp"][0]
    assert grasp.n_contacts == 2 and [c["name"] for c in grasp.contacts] == ["elbow", "end_effector"]
    assert grasp.contact_types == {"elbow": "ContactModel3D", "end_effector": "ContactModel3D"}
    d = problem.desc
    sets = [d.sets[k] for k in range(d.n_sets) if d.sets[k].ncontacts == 2]
    assert len(sets) == 1 and d.has_contact == 1
    c0, c1 = sets[0].contacts[0], sets[0].contacts[1]
    assert (c0.name.decode(), c1.name.decode()) == ("elbow", "end_effector") and c0.frame != c1.frame
    assert tuple(c0.gains) == (2.0, 0.7) and tuple(c1.gains) == (3.0, 1.5) and tuple(c0.ref_p) == (0.1, -0.05, 0.2)


def test_third_algorithm_arbitration_on_an_ill_conditioned_node(empc, tmp_path):
    """What the step-wise harness's arbitration rests on (tests/stepwise.py third_algorithm_distance): with the arm 1e-4 rad from
    stretched, Jc M^-1 Jc^T of the two point contacts is 1e-8 from singular.  There the NumPy restatement -- a third algorithm --
    sits far from the oracle in the digits where the oracle's FMA build (the SAME algorithm, the harness's usual yardstick) still
    agrees with it; with the arm bent all three agree to 1e-13."""
    import stepwise as sw
    _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", link2="flying_arm_3__link_1")
    d = problem.desc
    prm = ob.default_params()
    o, ofma = ob.OracleSolver(d, prm), ob.OracleSolver(d, prm, variant="fma")
    for s_ in (o, ofma):
        s_.set_smooth(0.1)
    t, dist = 45, {}
    for tag, arm in (("nearly stretched", (1e-4, -2e-4, 1.5e-4)), ("bent", (0.4, -0.7, 0.5))):
        x = np.zeros(d.nx)
        x[6] = 1.0
        x[7:10] = arm
        x[d.model.nq:] = 0.05
        u = np.full(d.nu, 4.0)
        u[d.n_rotors:] = 0.1
        it = dict(smooth=0.1, xs=np.tile(x, (d.T + 1, 1)), us=np.tile(u, (d.T, 1)))
        ref, rf = o.node_calc(t, x, u), ofma.node_calc(t, x, u)
        dist[tag] = (sw.third_algorithm_distance(d, prm, it, t, "Fu", ref["Fu"]), rel(np.ravel(rf["Fu"]), np.ravel(ref["Fu"])))
    print("Fu: (NumPy restatement, FMA build) vs oracle:", dist)
    assert dist["nearly stretched"][0] > 100 * dist["nearly stretched"][1] and dist["nearly stretched"][0] > 1e-11  # (measured 1.1e-10 and 1e-14)
    assert dist["bent"][0] < 1e-12 and dist["bent"][1] < 1e-12


@pytest.mark.parametrize("robot", ["arm3", "arm5"])
def test_pair_bodies_over_baked_tables_equal_the_runtime_family(empc, emu, emu_baked, tmp_path, robot):
    """The two-contact kernel bodies instantiated over the baked tables of the two shipped arm robots against the runtime-model
    instantiation, bit for bit on the CPU.  The LIBRARY does not carry baked pair units: built (round 6), their k_calc showed two
    far hits of the static hazard scan behind round 4's GPU memory fault (tools/isa_exec_copy_scan.py: a register copy as the last
    instruction in front of an EXEC restore, read 800 instructions later -- most likely the phi copies of the one-contact /
    two-contact branch, but not provable without hardware), which the runtime-model pair units do not; two-contact problems of the
    shipped robots therefore run the runtime-model family (empc_solver.hip find_table)."""
    if robot == "arm3":
        _, problem = two_contact_variant(empc, tmp_path, "ContactModel3D", (3.0, 1.5), (2.0, 0.7), cone_on_second=True)
    else:
        _, problem = arm5_two_contact_variant(empc, tmp_path, (2.0, 1.0), (0.0, 3.0))
    families_equal(emu, emu_baked, problem)


FIX2 = os.path.join(ROOT, "tests", "golden", "second_restatement_two_contacts")
NAMES2 = sorted(f[:-4] for f in os.listdir(FIX2) if f.endswith(".npz"))


def fixture_problem(empc, g, tmp_path):
    import ast
    from conftest import arm5_two_contact_variant as arm5
    meta = ast.literal_eval(str(g["meta"]))
    integrator = str(g["integrator"])
    if meta["robot"] == "arm3":
        return two_contact_variant(empc, tmp_path, "ContactModel3D", tuple(meta["gains"]), tuple(meta["gains2"]), integrator=integrator,
                                   cone_on_second=meta["cone_on_second"])[1]
    return arm5(empc, tmp_path, tuple(meta["gains"]), tuple(meta["gains2"]))[1]


@pytest.mark.parametrize("name", NAMES2)
def test_oracle_and_kernel_bodies_match_the_two_contact_fixtures(empc, emu, tmp_path, name):
    """tests/golden/second_restatement_two_contacts/*.npz: inputs and NumPy-restatement outputs (data, generated by
    tests/golden/make_second_restatement.py two_contacts) of one node per distinct cost set of the two-contact problems -- the C++
    oracle at 1e-10, the linearize kernel bodies on the lane emulator at 1e-9 (1e-8 on RK4 nodes), as the single-contact fixtures
    are held by tests/test_second_restatement.py and tests/test_gpu_second_restatement.py"""
    g = np.load(os.path.join(FIX2, name + ".npz"))
    problem = fixture_problem(empc, g, tmp_path)
    d = problem.desc
    prm = ob.default_params()
    o = ob.OracleSolver(d, prm)
    o.set_smooth(float(g["smooth"]))
    for i, t in enumerate(g["knots"]):
        term = int(t) == d.T
        r = o.node_calc(int(t), g["xs"][i], None if term else g["us"][i])
        for key in KEYS:
            if term and key in ("Fu", "Lu", "Lxu", "Luu"):
                continue
            got = r[key][:6] if key == "lam" else r[key]
            assert rel(np.ravel(got), np.ravel(g[key][i])) < 1e-10, ("oracle", name, int(t), key)
    # the kernel bodies: the fixture's nodes planted into a trajectory, one linearize pass
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    xs = np.tile(np.array(problem.x0), (d.T + 1, 1))
    xs[:, 7:7 + 3] += 0.3  # (off the stretched arm: the nodes that are NOT compared must not be singular either)
    us = np.full((d.T, d.nu), 4.0)
    us[:, d.n_rotors:] = 0.0
    for i, t in enumerate(g["knots"]):
        xs[int(t)] = g["xs"][i]
        if int(t) < d.T:
            us[int(t)] = g["us"][i]
    emu.emu_set_linearize_version(2)
    emu.emu_set_warmstart(e, ob.P(np.ascontiguousarray(xs)), ob.P(np.ascontiguousarray(us)))
    emu.emu_phase_setup(e, float(g["smooth"]), 0, 1e-9, 0)
    tape = np.zeros((d.T + 1, emu.emu_rec(e)))
    acc = np.zeros((d.T + 1, d.model.nv))
    emu.emu_phase_linearize(e, ob.P(tape), ob.P(acc))
    n, m = d.ndx, d.nu
    nm = n + m
    tol = 1e-8 if "rk4" in name else 1e-9
    for i, t in enumerate(g["knots"]):
        r_ = tape[int(t)]
        A, HX, o2 = r_[:n * nm].reshape(n, nm), r_[n * nm:2 * n * nm].reshape(n, nm), 2 * n * nm
        got = {"Fx": A[:, :n], "Fu": A[:, n:], "Lxx": HX[:, :n], "Lxu": HX[:, n:], "Luu": r_[o2:o2 + m * m].reshape(m, m),
               "Lx": r_[o2 + m * m:o2 + m * m + n], "Lu": r_[o2 + m * m + n:o2 + m * m + n + m]}
        for key in got:
            if int(t) == d.T and key in ("Fx", "Fu", "Lxu", "Luu", "Lu"):
                continue
            assert rel(np.ravel(got[key]), np.ravel(g[key][i])) < tol, ("kernel bodies", name, int(t), key, rel(np.ravel(got[key]), np.ravel(g[key][i])))
    emu.emu_destroy(e)


def test_third_algorithm_rollout_and_condition_number(empc, problems, tmp_path):
    """the two other arbiters of tests/stepwise.py: the NumPy restatement's forward pass (gap-aware on an infeasible iterate, plain on a
    feasible one) lands on the oracle's trial costs to 1e-11 on well-conditioned iterates, with the oracle's gains; and
    constraint_condition() tells a bent arm (cond ~ 50) from a nearly stretched one (1e8) and a knot without contacts (1)"""
    import stepwise as sw
    _, problem = problems["displacement"]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 1, nq=d.model.nq, amplitude=0.002, seed=5)
    iterates = sw.oracle_paths(d, prm, x0s, maxiter=12)[0]["iterates"]
    checked = 0
    for i in (3, 8):
        it = iterates[i]
        ddp = it["phase"] == sw.T.PHASE_DDP
        o = ob.OracleSolver(d, prm)
        o.set_x0(x0s[0])
        p = o.iter_probe(it["xs"], it["us"], it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"])
        K, k, _ = o.last_gains()
        for a_ in (1, 3):
            if not p["ok"][a_]:
                continue
            c3 = sw.third_algorithm_trial_cost(d, prm, it, x0s[0], K, k, 2.0 ** -a_, ddp, bool(p["is_feasible"]))
            assert abs(c3 - p["cost_try"][a_]) <= 1e-11 * (1 + abs(p["cost_try"][a_])), (i, a_, c3, p["cost_try"][a_])
            checked += 1
    assert checked >= 3
    _, pair = two_contact_variant(empc, tmp_path, "ContactModel3D", link2="flying_arm_3__link_1")
    dp = pair.desc
    conds = {}
    for tag, arm in (("bent", (0.4, -0.7, 0.5)), ("nearly stretched", (1e-4, -2e-4, 1.5e-4))):
        x = np.zeros(dp.nx)
        x[6] = 1.0
        x[7:10] = arm
        it = dict(smooth=0.1, xs=np.tile(x, (dp.T + 1, 1)), us=np.zeros((dp.T, dp.nu)))
        conds[tag] = (sw.constraint_condition(dp, prm, it, 45), sw.constraint_condition(dp, prm, it, 10))
    print(conds)
    assert conds["bent"][0] < 1e3 < 1e7 < conds["nearly stretched"][0] and conds["bent"][1] == 1.0
