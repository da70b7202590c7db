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
