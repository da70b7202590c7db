"""The drop-in surface of the mirrors: the statement sequence of the reference's
examples/python/trajectory.py:16-25 runs unchanged against the Python mirror -- `SolverSbFDDP(problem, trajectory.squash)`,
`solver.setCallbacks([...])`, `solver.solve([], [], maxiter=100)` -- and the C++ mirror has the same constructor, get_squash()
and setCallbacks (include/eagle_mpc/sbfddp.hpp:39-40, include/eagle_mpc/trajectory.hpp:68)."""
import numpy as np
import pytest


def test_squash_model_and_constructor_checks(empc):
    """CPU: the surface exists and validates its arguments before any GPU work"""
    trajectory = empc.Trajectory()
    trajectory.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
    sq = trajectory.squash
    assert isinstance(sq, empc.SquashingModelSmoothSat) and sq.ns == trajectory.nu == 9
    _, lb, ub = trajectory.platform()
    assert np.array_equal(sq.u_lb, lb) and np.array_equal(sq.s_ub, ub) and sq.smooth == 0.1
    problem = trajectory.createProblem(80, True, "IntegratedActionModelEuler")
    with pytest.raises(TypeError):
        empc.SolverSbFDDP(problem, 4)  # a positional batch is not accepted any more: the second argument is the squashing model
    other = empc.Trajectory()
    other.autoSetup(empc.yaml_path("hexacopter370/trajectories/hover.yaml"))
    with pytest.raises(empc.EmpcError, match="does not belong"):
        empc.SolverSbFDDP(problem, other.squash)
    if empc.device_count() == 0:
        with pytest.raises(empc.EmpcError, match="no HIP device"):
            empc.SolverSbFDDP(problem, trajectory.squash)


@pytest.mark.gpu
def test_reference_example_sequence(empc, capsys):
    """examples/python/trajectory.py:16-25 of the reference, statement by statement (eagle_mpc -> the mirror module, the
    crocoddyl callback -> the mirror's CallbackVerbose)"""
    eagle_mpc = empc
    dt = 20  # ms
    useSquash = True
    robotName = 'hexacopter370_flying_arm_3'
    trajectoryName = 'displacement'

    trajectory = eagle_mpc.Trajectory()
    trajectory.autoSetup(eagle_mpc.YAML_DIR + "/" + robotName + "/trajectories/" + trajectoryName + ".yaml")
    problem = trajectory.createProblem(dt, useSquash, "IntegratedActionModelEuler")

    if useSquash:
        solver = eagle_mpc.SolverSbFDDP(problem, trajectory.squash)
    else:
        solver = eagle_mpc.SolverBoxFDDP(problem)

    cb = eagle_mpc.CallbackVerbose()
    solver.setCallbacks([cb])
    solver.solve([], [], maxiter=100)
    # one callback invocation per DDP iteration, in order, with the final cost in the last one
    assert len(cb.lines) >= solver.iter + 1
    out = capsys.readouterr().out
    assert "iter" in out and "cost" in out
    assert solver.getCallbacks() == [cb]
    assert solver.problem is problem
    assert abs(float(cb.lines[-1].split()[1]) - solver.cost) < 1e-4 * (1 + abs(solver.cost))


def test_utils_package_layout(empc):
    """CPU: the reference's utils modules exist under the same names (bindings/python/eagle_mpc/utils/{path,simulator,tools}.py)"""
    from eagle_mpc_amd.utils.path import EAGLE_MPC_YAML_DIR
    from eagle_mpc_amd.utils.simulator import AerialSimulator
    from eagle_mpc_amd.utils.tools import saveLogfile
    assert EAGLE_MPC_YAML_DIR == empc.YAML_DIR and callable(saveLogfile) and AerialSimulator is empc.utils.AerialSimulator
    trajectory = empc.Trajectory()
    trajectory.autoSetup(EAGLE_MPC_YAML_DIR + "/hexacopter370_flying_arm_3/trajectories/displacement.yaml")
    pp, rm = trajectory.platform_params, trajectory.robot_model
    assert pp.n_rotors == 6 and pp.tau_f.shape == (6, 6) and pp.u_lb.shape == (9,) and pp.max_thrust == pp.u_ub[0]
    assert (rm.nq, rm.nv) == (10, 9) and isinstance(rm.name, str) and rm.name
    with pytest.raises(ValueError, match="controller"):
        AerialSimulator(rm, pp, 2, trajectory.initial_state)  # a trajectory has no solver to host the plant


@pytest.mark.gpu
@pytest.mark.parametrize("mpcName", ["carrot", "rail", "weighted"])
def test_reference_mpc_example_sequence(empc, mpcName):
    """examples/python/mpc.py:19-62 of the reference, statement by statement (eagle_mpc -> the mirror module, crocoddyl's
    callback -> the mirror's; the closed loop shortened to 40 plant steps), then the plant against the oracle's RK4 node."""
    import oracle_binding as ob
    eagle_mpc = empc
    from eagle_mpc_amd.utils.path import EAGLE_MPC_YAML_DIR
    from eagle_mpc_amd.utils.simulator import AerialSimulator
    dt = 20  # ms
    useSquash = True
    robotName = 'hexacopter370_flying_arm_3'
    trajectoryName = 'displacement'

    trajectory = eagle_mpc.Trajectory()
    trajectory.autoSetup(EAGLE_MPC_YAML_DIR + "/" + robotName + "/trajectories/" + trajectoryName + ".yaml")
    problem = trajectory.createProblem(dt, useSquash, "IntegratedActionModelEuler")

    if useSquash:
        solver = eagle_mpc.SolverSbFDDP(problem, trajectory.squash)
    else:
        solver = eagle_mpc.SolverBoxFDDP(problem)

    solver.setCallbacks([eagle_mpc.CallbackVerbose()])
    solver.solve([], [], maxiter=400)

    mpcPath = EAGLE_MPC_YAML_DIR + "/" + robotName + "/mpc/mpc.yaml"
    if mpcName == 'rail':
        mpcController = eagle_mpc.RailMpc(solver.xs, dt, mpcPath)
    elif mpcName == 'weighted':
        mpcController = eagle_mpc.WeightedMpc(trajectory, dt, mpcPath)
    else:
        mpcController = eagle_mpc.CarrotMpc(trajectory, solver.xs, dt, mpcPath)

    mpcController.updateProblem(0)
    mpcController.solver.solve(solver.xs[:mpcController.problem.T + 1], solver.us[:mpcController.problem.T])
    mpcController.solver.convergence_init = 1e-3

    dtSimulator = 2
    simulator = AerialSimulator(mpcController.robot_model, mpcController.platform_params, dtSimulator, solver.xs[0])
    t = 0
    for i in range(0, 40):
        mpcController.problem.x0 = simulator.states[-1]
        mpcController.updateProblem(int(t))
        mpcController.solver.solve(mpcController.solver.xs, mpcController.solver.us, mpcController.iters)
        control = np.copy(mpcController.solver.us_squash[0])
        simulator.simulateStep(control)
        t += dtSimulator
    assert len(simulator.states) == 41 and len(simulator.controls) == 40
    xs = np.array(simulator.states)
    assert np.isfinite(xs).all() and np.abs(xs[-1][:3] - xs[0][:3]).max() < 0.5  # 80 ms of flight: still near the start
    # the plant steps are the oracle's RK4 node of the free dynamics with these controls
    d = mpcController.problem.desc
    for k in (0, 17, 39):
        xn = ob.plant_rk4(d, simulator.states[k], simulator.controls[k], dtSimulator / 1000.0)
        assert np.abs(xn - simulator.states[k + 1]).max() < 1e-10
