"""The drop-in surface of the Python mirror, asserted from a table of names and behaviours.

What a user of the reference's bindings relies on, by class (bindings/python/eagle_mpc/trajectory.hpp:38-63, sbfddp.hpp,
mpc-base.hpp, utils/{path,simulator,tools}.py): attribute and method names, constructor argument order, what `solve` returns
and leaves behind, how often a callback fires, and that a controller loop over the simulator produces the plant the oracle's
RK4 node produces.  No statement of the reference's example scripts is kept here; `examples/python/*.py` of this repository are
the runnable counterparts (tests/test_examples.py)."""
import inspect

import numpy as np
import pytest

ARM3 = "hexacopter370_flying_arm_3"

# class -> names the reference's bindings export for it
SURFACE = {
    "Trajectory": ["autoSetup", "createProblem", "removeStage", "stages", "robot_model", "robot_model_path", "platform_params",
                   "squash", "initial_state", "duration"],
    "PlatformParams": [],  # instance attributes: cf, cm, n_rotors, tau_f, max_thrust, min_thrust, base_link_name, max_torque,
                           # min_torque, u_lb, u_ub, rotors_pose (test_platform_params_and_stage_views)
    "SolverSbFDDP": ["solve", "setCallbacks", "getCallbacks", "xs", "us", "us_squash", "iter", "cost", "stop", "problem",
                     "convergence_init"],
    "CarrotMpc": ["updateProblem", "createProblem", "solver", "problem", "robot_model", "robot_model_path", "platform_params", "squash", "iters"],
    "RailMpc": ["updateProblem", "createProblem", "solver", "problem", "robot_model", "robot_model_path", "platform_params", "squash", "iters"],
    "WeightedMpc": ["updateProblem", "createProblem", "solver", "problem", "robot_model", "robot_model_path", "platform_params", "squash", "iters"],
}


def arm3_trajectory(empc, name="displacement"):
    t = empc.Trajectory()
    t.autoSetup(empc.YAML_DIR + "/" + ARM3 + "/trajectories/" + name + ".yaml")
    return t


@pytest.mark.parametrize("cls", sorted(SURFACE))
def test_exported_names(empc, cls):
    """CPU: every name of the table is an attribute of the mirror class (properties included)"""
    c = getattr(empc, cls)
    missing = [n for n in SURFACE[cls] if not hasattr(c, n) and n not in ("xs", "us", "us_squash", "iter", "cost", "stop", "problem",
                                                                            "solver", "robot_model", "platform_params", "iters")]
    assert not missing, (cls, missing)


def test_constructor_signatures(empc):
    """CPU: argument order of the constructors and of solve / createProblem"""
    def names(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]
    assert names(empc.SolverSbFDDP.__init__)[:2] == ["problem", "squashing_model"]
    assert names(empc.SolverSbFDDP.solve)[:5] == ["init_xs", "init_us", "maxiter", "is_feasible", "regInit"]
    assert names(empc.Trajectory.createProblem) == ["dt", "squash", "integration_method"]
    assert names(empc.CarrotMpc.__init__)[:4] == ["trajectory", "state_ref", "dt_ref", "yaml_path"]
    assert names(empc.RailMpc.__init__)[:3] == ["state_ref", "dt_ref", "yaml_path"]
    assert names(empc.WeightedMpc.__init__)[:3] == ["trajectory", "dt_ref", "yaml_path"]


def test_trajectory_members(empc):
    """CPU: stages / duration / robot_model_path / removeStage / squash / platform_params / robot_model of a loaded file"""
    t = arm3_trajectory(empc)
    st = t.stages
    assert len(st) == t.n_stages >= 2 and all(isinstance(s, empc.StageInfo) for s in st)
    assert sum(s.duration for s in st) == t.duration and st[0].t_ini == 0
    assert [s.t_ini for s in st] == list(np.cumsum([0] + [s.duration for s in st[:-1]]))
    assert all(len(s.costs) == s.n_costs and all(c["name"] for c in s.costs) for s in st)
    path = t.robot_model_path
    assert path.endswith(".urdf") and ARM3 in path
    import os
    assert os.path.isfile(path)
    sq = t.squash
    assert isinstance(sq, empc.SquashingModelSmoothSat) and sq.ns == t.nu == 9 and sq.smooth == 0.1
    _, lb, ub = t.platform()
    assert np.array_equal(sq.u_lb, lb) and np.array_equal(sq.s_ub, ub)
    pp, rm = t.platform_params, t.robot_model
    assert pp.n_rotors == 6 and pp.tau_f.shape == (6, 6) and pp.u_lb.shape == (9,) and pp.max_thrust == pp.u_ub[0]
    assert (rm.nq, rm.nv) == (10, 9) and isinstance(rm.name, str) and rm.name


def test_platform_params_and_stage_views(empc):
    """CPU: MultiCopterBaseParams as the reference's bindings expose it (cf, cm, thrust limits, base link, rotor poses whose
    thrust axes and arms rebuild tau_f: src/multicopter-base-params.cpp:67-78) and the Stage views (cost_types, contacts,
    contact_types, is_terminal)"""
    t = arm3_trajectory(empc, "eagle_catch")
    pp = t.platform_params
    assert pp.cf == pytest.approx(4.138394792004922e-06) and pp.cm == pytest.approx(6.991478005829954e-08)
    assert pp.max_thrust == 20.6991 and pp.min_thrust == 0.0 and pp.base_link_name == "hexacopter370__base_link"
    assert pp.max_prop_speed == pytest.approx(np.sqrt(pp.max_thrust / pp.cf)) and pp.min_prop_speed == 0.0
    assert len(pp.rotors_pose) == pp.n_rotors == 6
    assert np.array_equal(pp.max_torque, pp.u_ub[6:]) and np.array_equal(pp.min_torque, pp.u_lb[6:]) and pp.max_torque.shape == (3,)
    for i, rp in enumerate(pp.rotors_pose):
        assert abs(np.linalg.det(rp.rotation) - 1.0) < 1e-12 and rp.spin_direction in (-1, 1)
        axis = rp.rotation[:, 2]  # thrust along the rotor frame's z axis
        assert np.allclose(pp.tau_f[:3, i], axis, atol=1e-12)
        assert np.allclose(pp.tau_f[3:, i], np.cross(rp.translation, axis) + rp.spin_direction * pp.cm / pp.cf * axis, atol=1e-12)
    stages = t.stages
    grasp = [s for s in stages if s.n_contacts][0]
    assert grasp.contacts == [{"name": grasp.contacts[0]["name"], "type": "ContactModel3D"}]
    assert grasp.contact_types == {grasp.contacts[0]["name"]: "ContactModel3D"}
    assert all(s.is_terminal is False for s in stages)
    known = {"CostModelState", "CostModelControl", "CostModelFramePlacement", "CostModelFrameRotation", "CostModelFrameVelocity",
             "CostModelFrameTranslation", "CostModelContactFrictionCone"}
    for s in stages:
        assert set(s.cost_types) == {c["name"] for c in s.costs} and set(s.cost_types.values()) <= known
    assert "CostModelContactFrictionCone" in grasp.cost_types.values()


def test_controller_members_without_a_gpu(empc):
    """CPU: what MpcAbstract's bindings expose besides the solver (bindings/python/eagle_mpc/mpc-base.hpp:39-72): knots, dt, iters,
    squash, robot_model_path, createProblem -- no GPU needed until `.solver` is touched"""
    t = arm3_trajectory(empc)
    yaml = empc.YAML_DIR + "/" + ARM3 + "/mpc/mpc.yaml"
    ref = np.tile(t.initial_state, (40, 1))
    for mpc in (empc.CarrotMpc(t, ref, 20, yaml), empc.WeightedMpc(arm3_trajectory(empc), 20, yaml), empc.RailMpc(ref, 20, yaml)):
        assert mpc.knots == mpc.problem.T + 1 and mpc.dt > 0 and mpc.iters >= 1
        sq = mpc.squash
        assert isinstance(sq, empc.SquashingModelSmoothSat) and sq.ns == mpc.nu and np.array_equal(sq.u_ub, mpc.platform_params.u_ub)
        assert mpc.createProblem() is mpc.problem
        path = mpc.robot_model_path
        assert path is None if isinstance(mpc, empc.RailMpc) else path.endswith(".urdf")


def test_remove_stage(empc):
    """CPU: removeStage erases exactly that stage, leaves the others' durations and start times alone (as the reference
    does, src/trajectory.cpp:145-150) and the problem built afterwards has that many knots fewer"""
    t = arm3_trajectory(empc)
    before = t.stages
    T0 = t.createProblem(20, True, "IntegratedActionModelEuler").T
    victim = 1
    t.removeStage(victim)
    after = t.stages
    assert [s.name for s in after] == [s.name for i, s in enumerate(before) if i != victim]
    assert [(s.duration, s.t_ini) for s in after] == [(s.duration, s.t_ini) for i, s in enumerate(before) if i != victim]
    assert t.duration == sum(s.duration for s in before)  # duration_ is what autoSetup summed
    T1 = t.createProblem(20, True, "IntegratedActionModelEuler").T
    assert T0 - T1 == before[victim].duration // 20
    with pytest.raises(IndexError):
        t.removeStage(len(after))
    with pytest.raises(IndexError):
        t.removeStage(-1)


def test_solver_constructor_checks(empc):
    """CPU: the constructor validates its arguments before any GPU work"""
    t = arm3_trajectory(empc)
    problem = t.createProblem(80, True, "IntegratedActionModelEuler")
    with pytest.raises(TypeError, match="squashing model"):
        empc.SolverSbFDDP(problem, 4)  # rounds 1-2 took the batch here; the reference takes the squashing model
    other = empc.Trajectory()
    other.autoSetup(empc.yaml_path("hexacopter370/trajectories/hover.yaml"))
    with pytest.raises(empc.EmpcError, match="does not belong"):
        empc.SolverSbFDDP(problem, other.squash)
    if empc.device_count() == 0:
        with pytest.raises(empc.EmpcError, match="no HIP device"):
            empc.SolverSbFDDP(problem, t.squash)


def test_utils_package_layout(empc):
    """CPU: the reference's utils modules exist under the same names (bindings/python/eagle_mpc/utils/{path,simulator,tools}.py)"""
    from eagle_mpc_amd.utils.path import EAGLE_MPC_YAML_DIR
    from eagle_mpc_amd.utils.simulator import AerialSimulator
    from eagle_mpc_amd.utils.tools import saveLogfile
    assert EAGLE_MPC_YAML_DIR == empc.YAML_DIR and callable(saveLogfile) and AerialSimulator is empc.utils.AerialSimulator
    t = arm3_trajectory(empc)
    with pytest.raises(ValueError, match="controller"):
        AerialSimulator(t.robot_model, t.platform_params, 2, t.initial_state)  # a trajectory has no solver to host the plant


def test_callback_verbose_header_cadence(empc):
    """CPU: CallbackVerbose reprints its header every 10 iterations, like crocoddyl's and like the C++ mirror"""
    import io
    out = io.StringIO()
    cb = empc.CallbackVerbose(out)

    for i in range(25):
        # one trace record: phase, iter, cost, stop, xreg, step length, feasible, dV, dVexp, gap norm, d0, d1
        cb(empc.IterationRecord([0, i, 1.0, 0.1, 1e-9, 1.0, 1, 0.0, 0.0, 0.0, 0.0, -0.5], None))
    assert len(cb.lines) == 25
    headers = [l for l in out.getvalue().splitlines() if l.lstrip().startswith("iter")]
    assert len(headers) == 3  # before iterations 0, 10 and 20


@pytest.mark.gpu
def test_solve_leaves_the_reference_attributes(empc, capsys):
    """GPU: SolverSbFDDP(problem, trajectory.squash) + setCallbacks + solve([], [], maxiter): returns True, fires one callback
    per iteration in order, leaves xs (T + 1 states), us / us_squash (T controls), iter, cost, stop; getCallbacks returns the list;
    SolverBoxFDDP takes the problem alone"""
    t = arm3_trajectory(empc)
    problem = t.createProblem(20, True, "IntegratedActionModelEuler")
    solver = empc.SolverSbFDDP(problem, t.squash)
    cb = empc.CallbackVerbose()
    solver.setCallbacks([cb])
    assert solver.getCallbacks() == [cb] and solver.problem is problem
    assert solver.solve([], [], maxiter=100) is True
    T = problem.T
    assert len(solver.xs) == T + 1 and len(solver.us) == T and len(solver.us_squash) == T
    assert all(np.shape(x) == (t.nx,) for x in solver.xs) and all(np.shape(u) == (t.nu,) for u in solver.us_squash)
    assert len(cb.lines) >= solver.iter + 1
    assert abs(float(cb.lines[-1].split()[1]) - solver.cost) < 1e-4 * (1 + abs(solver.cost))
    out = capsys.readouterr().out
    assert "iter" in out and "cost" in out
    lb, ub = t.squash.u_lb, t.squash.u_ub
    assert all(np.all(u >= lb - 1e-12) and np.all(u <= ub + 1e-12) for u in solver.us_squash)
    box = empc.SolverBoxFDDP(t.createProblem(20, False, "IntegratedActionModelEuler"))
    assert box.solve([], [], maxiter=3) is True


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["carrot", "rail", "weighted"])
def test_controller_loop_over_the_simulator(empc, kind):
    """GPU: each controller class, built from its reference constructor arguments, runs a closed loop with
    utils.simulator.AerialSimulator; every plant step equals the oracle's RK4 node of the free dynamics (1e-10)"""
    import oracle_binding as ob
    from eagle_mpc_amd.utils.simulator import AerialSimulator
    dt, dt_sim, steps = 20, 2, 40
    t = arm3_trajectory(empc)
    plan = empc.SolverSbFDDP(t.createProblem(dt, True, "IntegratedActionModelEuler"), t.squash)
    plan.solve([], [], maxiter=400)
    yaml = empc.YAML_DIR + "/" + ARM3 + "/mpc/mpc.yaml"
    mpc = {"carrot": lambda: empc.CarrotMpc(t, plan.xs, dt, yaml), "rail": lambda: empc.RailMpc(plan.xs, dt, yaml),
           "weighted": lambda: empc.WeightedMpc(t, dt, yaml)}[kind]()
    for name in SURFACE["CarrotMpc"]:
        assert hasattr(mpc, name), name
    H = mpc.problem.T
    mpc.updateProblem(0)
    mpc.solver.solve(plan.xs[:H + 1], plan.us[:H])
    mpc.solver.convergence_init = 1e-3
    sim = AerialSimulator(mpc.robot_model, mpc.platform_params, dt_sim, plan.xs[0])
    for k in range(steps):
        mpc.problem.x0 = sim.states[-1]
        mpc.updateProblem(k * dt_sim)
        mpc.solver.solve(mpc.solver.xs, mpc.solver.us, mpc.iters)
        sim.simulateStep(np.copy(mpc.solver.us_squash[0]))
    assert len(sim.states) == steps + 1 and len(sim.controls) == steps
    xs = np.array(sim.states)
    assert np.isfinite(xs).all() and np.abs(xs[-1][:3] - xs[0][:3]).max() < 0.5  # 80 ms of flight: still near the start
    d = mpc.problem.desc
    for k in (0, 17, steps - 1):
        xn = ob.plant_rk4(d, sim.states[k], sim.controls[k], dt_sim / 1000.0)
        assert np.abs(xn - sim.states[k + 1]).max() < 1e-10
