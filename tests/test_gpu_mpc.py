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


def test_closed_loop_matches_oracle(empc):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    dt_traj = 80
    prob = traj.createProblem(dt_traj, True, "IntegratedActionModelEuler")
    tsolver = empc.SolverSbFDDP(prob, batch=1)
    tsolver.solve([], [], 100)
    xs_ref = np.array(tsolver.xs)
    us_ref = np.array(tsolver.us)

    B, n_steps, dt_sim = 4, 6, 2
    mpc = empc.CarrotMpc(traj, xs_ref, dt_traj, empc.yaml_path(ARM3_MPC), batch=B)
    T_ = mpc.problem.T
    nq = traj.nx - traj.ndx // 2
    x_plants = empc.perturbed_x0s(xs_ref[0], B, nq=nq, amplitude=0.02)
    d = mpc.problem.desc

    mpc.updateProblem(0)
    solver = mpc.solver
    solver.plant_states = x_plants
    # first solve: warm start = the head of the planned trajectory (mpc.py:38)
    solver.solve(xs_ref[:T_ + 1], us_ref[:T_], 100, x0s="plant")
    solver.convergence_init = 1e-3                                   # mpc.py:39
    o_xs, o_us, o_x = [], [], x_plants.copy()
    for b in range(B):
        s = ob.OracleSolver(d)
        s.set_x0(o_x[b])
        s.solve(xs_ref[:T_ + 1], us_ref[:T_], 100)
        r = s.result()
        o_xs.append(r["xs"])
        o_us.append(r["us"])
    assert np.abs(solver.xs_batch - np.array(o_xs)).max() < 1e-6
    assert np.abs(solver.us_batch - np.array(o_us)).max() < 1e-6

    t = 0
    worst = 0.0
    for step in range(n_steps):
        mpc.updateProblem(t)
        solver.solve("previous", "previous", mpc.iters, x0s="plant")
        usq = solver.us_squash_batch[:, 0]
        solver.plant_step(dt_sim)
        gx = solver.plant_states
        for b in range(B):
            s = ob.OracleSolver(d)
            ob.orc().oracle_solver_set_convergence_init(s.h, C.c_double(1e-3))
            s.set_x0(o_x[b])
            s.solve(o_xs[b], o_us[b], mpc.iters)
            r = s.result()
            o_xs[b], o_us[b] = r["xs"], r["us"]
            assert np.abs(usq[b] - r["us_squash"][0]).max() < 1e-6, (step, b)
            o_x[b] = ob.plant_rk4(d, o_x[b], r["us_squash"][0], dt_sim / 1000.0)[0]
        worst = max(worst, np.abs(gx - o_x).max())
        t += dt_sim
    assert worst < 1e-6
    # the plants moved and stayed bounded
    assert np.isfinite(gx).all() and np.abs(gx - x_plants).max() > 1e-5
