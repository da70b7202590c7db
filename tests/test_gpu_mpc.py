"""Closed-loop Carrot MPC on the GPU against the CPU oracle (BASELINE.json configs[4]: receding horizon, RK4 plant).

The loop is the one of examples/python/mpc.py:30-62: updateProblem(t) -> solve(previous xs, us, iters) -> plant step with
us_squash[0] -> t += dt_simulator.  On the GPU the B plants, their MPC solves and the warm starts stay resident on the
device; the oracle runs the same loop one plant at a time on the CPU.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu

ARM3_TRAJ = "hexacopter370_flying_arm_3/trajectories/displacement.yaml"
ARM3_MPC = "hexacopter370_flying_arm_3/mpc/mpc.yaml"


def random_states(d, B, seed):
    rng = np.random.default_rng(seed)
    nq, nv = d.model.nq, d.model.nv
    x = np.zeros((B, d.nx))
    x[:, :3] = rng.uniform(-1, 1, (B, 3))
    q = rng.normal(size=(B, 4))
    x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x[:, 7:nq] = rng.uniform(-1, 1, (B, nq - 7))
    x[:, nq:] = rng.uniform(-0.5, 0.5, (B, nv))
    u = np.concatenate([rng.uniform(1, 6, (B, d.n_rotors)), rng.uniform(-2e-3, 2e-3, (B, d.nu - d.n_rotors))], axis=1)
    return x, u


@pytest.mark.parametrize("name", ["hover", "displacement", "push_slide"])
def test_plant_rk4_parity(empc, problems, name):
    _, prob = problems[name]
    d = prob.desc
    B = 16
    x, u = random_states(d, B, 11)
    solver = empc.SolverSbFDDP(prob, batch=B)
    solver.plant_states = x
    assert np.array_equal(solver.plant_states, x)
    solver.plant_step(2, controls=u, substeps=3)
    got = solver.plant_states
    exp = ob.plant_rk4(d, x, u, 0.002, substeps=3)
    assert np.abs(got - exp).max() < 1e-11
    # a plant step before any solve needs explicit controls
    s2 = empc.SolverSbFDDP(prob, batch=1)
    with pytest.raises(empc.EmpcError, match="no solve has run"):
        s2.plant_step(2)


def planned_trajectory(empc, dt_traj=80, squashed=False):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    prob = traj.createProblem(dt_traj, True, "IntegratedActionModelEuler")
    tsolver = empc.SolverSbFDDP(prob, batch=1)
    tsolver.solve([], [], 100)
    return traj, np.array(tsolver.xs), np.array(tsolver.us_squash if squashed else tsolver.us)


def closed_loop(empc, mpc, xs_ref, us_ref, nq, B=4, n_steps=6, dt_sim=2, tol=1e-6, sample=None, solver_type=0):
    """examples/python/mpc.py:30-62 on the GPU (B plants at once) and on the oracle (one plant at a time; `sample` =
    the plants the oracle follows, default all of them)."""
    sample = list(range(B)) if sample is None else list(sample)
    T_ = mpc.problem.T
    x_plants = empc.perturbed_x0s(xs_ref[0], B, nq=nq, amplitude=0.02)
    d = mpc.problem.desc
    oprm = ob.default_params()
    oprm.solver_type = solver_type

    mpc.updateProblem(0)
    solver = mpc.solver
    assert solver.SOLVER_TYPE == solver_type
    solver.plant_states = x_plants
    # first solve: warm start = the head of the planned trajectory (mpc.py:38)
    solver.solve(xs_ref[:T_ + 1], us_ref[:T_], 100, x0s="plant")
    solver.convergence_init = 1e-3                                   # mpc.py:39
    o_xs, o_us, o_x = {}, {}, x_plants.copy()
    for b in sample:
        s = ob.OracleSolver(d, oprm)
        s.set_x0(o_x[b])
        s.solve(xs_ref[:T_ + 1], us_ref[:T_], 100)
        r = s.result()
        o_xs[b] = r["xs"]
        o_us[b] = r["us"]
    assert np.abs(solver.xs_batch[sample] - np.array([o_xs[b] for b in sample])).max() < tol
    assert np.abs(solver.us_batch[sample] - np.array([o_us[b] for b in sample])).max() < tol

    t = 0
    worst = 0.0
    for step in range(n_steps):
        mpc.updateProblem(t)
        solver.solve("previous", "previous", mpc.iters, x0s="plant")
        usq = solver.us_squash_batch[:, 0]
        solver.plant_step(dt_sim)
        gx = solver.plant_states
        for b in sample:
            s = ob.OracleSolver(d, oprm)
            ob.orc().oracle_solver_set_convergence_init(s.h, C.c_double(1e-3))
            s.set_x0(o_x[b])
            s.solve(o_xs[b], o_us[b], mpc.iters)
            r = s.result()
            o_xs[b], o_us[b] = r["xs"], r["us"]
            assert np.abs(usq[b] - r["us_squash"][0]).max() < tol, (step, b)
            o_x[b] = ob.plant_rk4(d, o_x[b], r["us_squash"][0], dt_sim / 1000.0)[0]
        worst = max(worst, np.abs(gx[sample] - o_x[sample]).max())
        t += dt_sim
    assert worst < tol
    # the plants moved and stayed bounded
    assert np.isfinite(gx).all() and np.abs(gx - x_plants).max() > 1e-5
    return worst


def test_carrot_50_knots_batch_256(empc):
    """BASELINE.json configs[4] at its stated size: 50-knot receding horizon, 256 perturbed plants, RK4 plant at 2 ms.
    The oracle follows a sample of the plants through every controller cycle (1e-6 on controls and plant states); all
    256 stay finite, and a plant's closed loop does not depend on its neighbours in the batch (bitwise re-run of a subset)."""
    import os
    traj, xs_ref, us_ref = planned_trajectory(empc)
    yaml50 = os.path.join(os.path.dirname(empc.YAML_DIR), "mpc", "carrot_50knots.yaml")
    B, sample = 256, [0, 1, 100, 255]
    mpc = empc.CarrotMpc(traj, xs_ref, 80, yaml50, batch=B)
    assert mpc.knots == 50
    nq = traj.nx - traj.ndx // 2
    closed_loop(empc, mpc, xs_ref, us_ref, nq=nq, B=B, n_steps=8, sample=sample)
    full_states = mpc.solver.plant_states
    assert np.isfinite(full_states).all() and np.isfinite(mpc.solver.us_squash_batch).all()
    # batch independence: the same four plants alone
    traj2, _, _ = planned_trajectory(empc)
    mpc2 = empc.CarrotMpc(traj2, xs_ref, 80, yaml50, batch=len(sample))
    x_all = empc.perturbed_x0s(xs_ref[0], B, nq=nq, amplitude=0.02)
    T_ = mpc2.problem.T
    mpc2.updateProblem(0)
    s2 = mpc2.solver
    s2.plant_states = np.ascontiguousarray(x_all[sample])
    s2.solve(xs_ref[:T_ + 1], us_ref[:T_], 100, x0s="plant")
    s2.convergence_init = 1e-3
    t = 0
    for _ in range(8):
        mpc2.updateProblem(t)
        s2.solve("previous", "previous", mpc2.iters, x0s="plant")
        s2.plant_step(2)
        t += 2
    assert np.array_equal(s2.plant_states, full_states[sample])


def test_closed_loop_matches_oracle(empc):
    traj, xs_ref, us_ref = planned_trajectory(empc)
    mpc = empc.CarrotMpc(traj, xs_ref, 80, empc.yaml_path(ARM3_MPC), batch=4)
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2)


def test_rail_closed_loop_matches_oracle(empc):
    """RailMpc (src/mpc-controllers/rail-mpc.cpp): every knot tracks the planned state at its own time."""
    traj, xs_ref, us_ref = planned_trajectory(empc)
    mpc = empc.RailMpc(xs_ref, 80, empc.yaml_path(ARM3_MPC), batch=4)
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2)


def test_weighted_closed_loop_matches_oracle(empc):
    """WeightedMpc (src/mpc-controllers/weighted-mpc.cpp): task costs of the active stage, exponentially weighted; the
    cost sets carry an operational frame, so this also runs the full linearize body inside the MPC loop."""
    traj, xs_ref, us_ref = planned_trajectory(empc)
    mpc = empc.WeightedMpc(traj, 80, empc.yaml_path(ARM3_MPC), batch=4)
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2)


def test_carrot_closed_loop_with_rk4_nodes(empc, tmp_path):
    """`integration_method: IntegratedActionModelRK4` in the controller YAML (src/mpc-base.cpp:41-43,
    src/mpc-controllers/carrot-mpc.cpp:216-217): the receding-horizon problem is built from RK4 nodes; closed loop on the
    GPU against the oracle."""
    src = open(empc.yaml_path(ARM3_MPC)).read()
    assert "IntegratedActionModelEuler" in src
    f = tmp_path / "mpc_rk4.yaml"
    f.write_text(src.replace("IntegratedActionModelEuler", "IntegratedActionModelRK4"))
    traj, xs_ref, us_ref = planned_trajectory(empc)
    mpc = empc.CarrotMpc(traj, xs_ref, 80, str(f), batch=2)
    assert mpc.problem.desc.integrator == 1
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2, B=2, n_steps=4, tol=1e-5)


@pytest.mark.parametrize("solver_name,solver_type", [("SolverBoxFDDP", 1), ("SolverBoxDDP", 2)])
def test_carrot_closed_loop_with_box_solver(empc, tmp_path, solver_name, solver_type):
    """`solver: SolverBoxFDDP | SolverBoxDDP` in the controller YAML (src/mpc-base.cpp:50,
    src/mpc-controllers/carrot-mpc.cpp:188-193,232-242): the controller builds its nodes on the plain actuation and hands
    them to crocoddyl's box solver; here the same kernels with the box-QP backward pass, closed loop against the oracle."""
    src = open(empc.yaml_path(ARM3_MPC)).read()
    assert "SolverSbFDDP" in src
    f = tmp_path / "mpc_box.yaml"
    f.write_text(src.replace("SolverSbFDDP", solver_name))
    traj, xs_ref, us_ref = planned_trajectory(empc, squashed=True)  # rotor thrusts inside their limits
    mpc = empc.CarrotMpc(traj, xs_ref, 80, str(f), batch=2)
    assert mpc.solver_type == solver_name and not mpc.problem.desc.use_squash
    closed_loop(empc, mpc, xs_ref, us_ref, nq=traj.nx - traj.ndx // 2, B=2, n_steps=4, tol=1e-5, solver_type=solver_type)
