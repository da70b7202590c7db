"""CarrotMpc / MpcAbstract host logic and the CPU restatement of the RK4 plant (no GPU needed).

Expected values are derived by hand from the reference's rules:
  src/mpc-base.cpp:40-60 (controller parameters), src/mpc-controllers/carrot-mpc.cpp:15-50 (t_stages),
  :250-296 (cost table of every knot), :298-362 (update rules), :384-403 (state reference, integer alpha),
  bindings/python/eagle_mpc/utils/simulator.py:8-29 (plant).
"""
import bisect
import math

import numpy as np
import pytest

import oracle_binding as ob

ARM3_TRAJ = "hexacopter370_flying_arm_3/trajectories/displacement.yaml"
ARM3_MPC = "hexacopter370_flying_arm_3/mpc/mpc.yaml"


def make_reference(traj, n=101, dt_ref=80):
    """A synthetic 'solved trajectory': distinct states so that every lookup is identifiable."""
    x0 = traj.initial_state
    ref = np.tile(x0, (n, 1))
    ref[:, 0] = np.linspace(0.0, 2.0, n)          # x position
    ref[:, 7] = 0.01 * np.arange(n)               # first arm joint
    ref[:, traj.nx - 1] = 0.001 * np.arange(n)    # last joint velocity
    return ref, dt_ref


@pytest.fixture(scope="module")
def carrot(empc):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    ref, dt_ref = make_reference(traj)
    return traj, ref, dt_ref, empc.CarrotMpc(traj, ref, dt_ref, empc.yaml_path(ARM3_MPC))


def cost_table(desc, knot):
    st = desc.sets[desc.knot_set[knot]]
    return {st.costs[i].name.decode(): st.costs[i] for i in range(st.ncosts)}


def test_controller_parameters(carrot):
    traj, ref, dt_ref, m = carrot
    assert (m.knots, m.iters, m.dt) == (30, 2, 30)            # mpc.yaml: knots 30, iters 2, dt 30
    d = m.problem.desc
    assert d.T == 29 and d.n_sets == 30                       # running = first knots-1 models, terminal = last
    assert [d.knot_set[i] for i in range(30)] == list(range(30))  # one private action model per knot
    assert d.use_squash == 1 and d.has_contact == 0 and d.integrator == 0
    assert d.dt == pytest.approx(0.03)
    x0 = np.array([d.x0[i] for i in range(d.nx)])
    zero = np.zeros(d.nx)
    zero[6] = 1.0
    assert np.array_equal(x0, zero)                           # problem x0 = robot_state->zero()
    assert (d.nx, d.ndx, d.nu) == (traj.nx, traj.ndx, traj.nu)


def test_t_stages_floor_durations_to_dt(carrot):
    traj, ref, dt_ref, m = carrot
    # stage durations 2000,0,2000,0,2000,0,2000,0 -> zero-length way-points last one controller dt (30 ms)
    assert m.t_stages == [0, 2000, 2030, 4030, 4060, 6060, 6090, 8090, 8120]


def test_cost_table_of_a_knot(carrot, empc):
    traj, ref, dt_ref, m = carrot
    d = m.problem.desc
    c = cost_table(d, 3)
    assert sorted(c) == ["carrot_state", "carrot_tail", "control_reg", "state_limits", "state_reg"]
    T = empc.T
    assert c["state_reg"].weight == 1e-2 and c["state_reg"].activation == 1 and c["state_reg"].active == 1
    assert c["control_reg"].weight == 1e-1 and c["control_reg"].type == 1
    assert [c["control_reg"].act_w[i] for i in range(d.nu)] == [1, 1, 1, 1, 1, 1, 10, 10, 10]
    assert c["state_limits"].weight == 10 and c["state_limits"].activation == 3
    assert [c["state_limits"].ub[i] for i in range(d.ndx)] == [0] * 6 + [1.9] * 3 + [0] * 6 + [3] * 3
    assert [c["state_limits"].lb[i] for i in range(d.ndx)] == [0] * 6 + [-1.9] * 3 + [0] * 6 + [-3] * 3
    assert [c["state_limits"].act_w[i] for i in range(d.ndx)] == [0] * 6 + [1] * 3 + [0] * 6 + [1] * 3
    assert c["carrot_state"].weight == 1000 and c["carrot_state"].activation == 0
    assert c["carrot_tail"].weight == 1 and c["carrot_tail"].activation == 1
    assert [c["carrot_tail"].act_w[i] for i in range(d.ndx)] == [1000] * 3 + [1] * 3 + [10] * 3 + [1] * 9


def expected_reference(ref, dt_ref, time, nq):
    """carrot-mpc.cpp:384-403 with the std::size_t division: the reference is the sample at or before `time`."""
    t_ref = [dt_ref * i for i in range(len(ref))]
    idx = bisect.bisect_right(t_ref, time)
    if idx >= len(ref):
        x = np.zeros(ref.shape[1])
        x[6] = 1.0
        x[:nq] = ref[-1, :nq]
        return x
    assert (time - t_ref[idx - 1]) // (t_ref[idx] - t_ref[idx - 1]) == 0
    return ref[idx - 1].copy()


def expected_update(traj_stages, t_stages, knots, dt, ref, dt_ref, nq, t, state):
    """state: per knot dict(carrot_active, tail_active, carrot_ref, tail_ref) mutated like the reference does."""
    for i in range(knots):
        node_time = t + i * dt
        idx_stage = bisect.bisect_right(t_stages, node_time) - 1
        s = state[i]
        if idx_stage < len(traj_stages):
            if (not traj_stages[idx_stage]["is_transition"]) or i == knots - 1:
                s["carrot_active"] = 1
                s["carrot_ref"] = expected_reference(ref, dt_ref, node_time, nq)
            else:
                s["carrot_active"] = 0
        else:
            s["carrot_active"] = 0
            s["tail_active"] = 1
            s["tail_ref"] = expected_reference(ref, dt_ref, node_time, nq)


def test_update_problem_rules(empc):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    ref, dt_ref = make_reference(traj)
    m = empc.CarrotMpc(traj, ref, dt_ref, empc.yaml_path(ARM3_MPC))
    stages = [traj.stage_info(i) for i in range(traj.n_stages)]
    zero = np.zeros(traj.nx)
    zero[6] = 1.0
    state = [dict(carrot_active=0, tail_active=0, carrot_ref=zero.copy(), tail_ref=zero.copy()) for _ in range(m.knots)]
    nq = traj.nx - traj.ndx // 2
    # times crossing way-points (2000-2030 ms), the end of the trajectory (8120 ms) and beyond the reference samples
    for t in [0, 2, 1130, 1160, 1990, 2000, 2029, 2030, 3999, 7250, 7260, 8000, 8119, 8120, 9000]:
        m.updateProblem(t)
        expected_update(stages, m.t_stages, m.knots, m.dt, ref, dt_ref, nq, t, state)
        d = m.problem.desc
        for i in range(m.knots):
            c = cost_table(d, i)
            assert c["carrot_state"].active == state[i]["carrot_active"], (t, i)
            assert c["carrot_tail"].active == state[i]["tail_active"], (t, i)
            got = np.array([c["carrot_state"].ref[k] for k in range(traj.nx)])
            assert np.array_equal(got, state[i]["carrot_ref"]), (t, i)
            got = np.array([c["carrot_tail"].ref[k] for k in range(traj.nx)])
            assert np.array_equal(got, state[i]["tail_ref"]), (t, i)
            # the regularisers are never touched
            assert c["state_reg"].active == 1 and c["control_reg"].active == 1 and c["state_limits"].active == 1
    # at t = 0 only the last knot carries the carrot while inside a transition stage
    m2 = empc.CarrotMpc(traj, ref, dt_ref, empc.yaml_path(ARM3_MPC))
    m2.updateProblem(0)
    act = [cost_table(m2.problem.desc, i)["carrot_state"].active for i in range(m2.knots)]
    assert act == [0] * 29 + [1]
    # a horizon straddling the first way-point [2000, 2030): exactly the knot inside it and the last one are active
    m2.updateProblem(1700)
    act = [cost_table(m2.problem.desc, i)["carrot_state"].active for i in range(m2.knots)]
    assert [i for i, a in enumerate(act) if a] == [10, 29]   # 1700 + 10*30 = 2000


def test_state_reference_integer_alpha(carrot):
    traj, ref, dt_ref, m = carrot
    nq = traj.nx - traj.ndx // 2
    for t in [0, 1, 79, 80, 81, 159, 160, 7999, 8000, 8001, 20000]:
        assert np.array_equal(m.computeStateReference(t), expected_reference(ref, dt_ref, t, nq)), t
    # no interpolation: 79 ms still returns sample 0, not a blend (size_t division at carrot-mpc.cpp:390-391)
    assert np.array_equal(m.computeStateReference(79), ref[0])
    # past the last sample: zero velocities, configuration of the last sample
    tail = m.computeStateReference(8000)
    assert np.array_equal(tail[:nq], ref[-1, :nq]) and not tail[nq:].any()


def test_contact_trajectory_is_rejected(empc):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml"))
    ref = np.tile(traj.initial_state, (4, 1))
    with pytest.raises(empc.EmpcError, match="Carrot with contact has not been implemented"):
        empc.CarrotMpc(traj, ref, 10, empc.yaml_path(ARM3_MPC))


def test_bad_arguments(empc, tmp_path):
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    ref = np.tile(traj.initial_state, (4, 1))
    with pytest.raises(empc.EmpcError):
        empc.CarrotMpc(traj, ref, 10, str(tmp_path / "missing.yaml"))
    # missing mandatory key (knots): the ParamsServer exception propagates like in the reference
    text = open(empc.yaml_path(ARM3_MPC)).read().replace("  knots: 30\n", "")
    text = text.replace('follow: "hexacopter370_flying_arm_3/platform/hexacopter370.yaml"',
                        'follow: "%s"' % empc.yaml_path("hexacopter370_flying_arm_3/platform/hexacopter370.yaml"))
    bad = tmp_path / "mpc.yaml"
    bad.write_text(text)
    with pytest.raises(empc.EmpcError, match="knots"):
        empc.CarrotMpc(traj, ref, 10, str(bad))
    # wrong weight-vector size surfaces at cost construction
    text2 = open(empc.yaml_path(ARM3_MPC)).read().replace(
        "carrot_control_reg_act_weights: [1, 1, 1, 1, 1, 1, 10, 10, 10]", "carrot_control_reg_act_weights: [1, 1, 1]")
    text2 = text2.replace('follow: "hexacopter370_flying_arm_3/platform/hexacopter370.yaml"',
                          'follow: "%s"' % empc.yaml_path("hexacopter370_flying_arm_3/platform/hexacopter370.yaml"))
    bad2 = tmp_path / "mpc2.yaml"
    bad2.write_text(text2)
    with pytest.raises(empc.EmpcError, match="dimension"):
        empc.CarrotMpc(traj, ref, 10, str(bad2))
    m = empc.CarrotMpc(traj, ref, 10, empc.yaml_path(ARM3_MPC))
    with pytest.raises(empc.EmpcError):
        m.updateProblem(-1)


# ---- plant (oracle restatement) ---------------------------------------------------------------------------

def test_plant_free_fall_known_answer(problems):
    """Zero thrust, zero torques, level attitude at rest: every body falls with g; RK4 is exact for a parabola."""
    _, prob = problems["displacement"]
    d = prob.desc
    x = np.zeros(d.nx)
    x[6] = 1.0
    x[2] = 5.0
    dt = 0.05
    xn = ob.plant_rk4(d, x, np.zeros(d.nu), dt)[0]
    nq = d.model.nq
    exp = x.copy()
    exp[2] = 5.0 - 0.5 * 9.81 * dt * dt
    exp[nq + 2] = -9.81 * dt
    assert np.allclose(xn, exp, atol=1e-12)


def test_plant_rk4_convergence_order(problems):
    _, prob = problems["displacement"]
    d = prob.desc
    rng = np.random.default_rng(3)
    x = np.zeros(d.nx)
    x[:3] = rng.uniform(-1, 1, 3)
    q = rng.normal(size=4)
    x[3:7] = q / np.linalg.norm(q)
    x[7:d.model.nq] = rng.uniform(-1, 1, d.model.nq - 7)
    x[d.model.nq:] = rng.uniform(-1, 1, d.model.nv)
    u = np.concatenate([rng.uniform(2, 6, d.n_rotors), rng.uniform(-2e-3, 2e-3, d.nu - d.n_rotors)])  # light arm links
    h = 0.008
    fine = ob.plant_rk4(d, x, u, h / 16, substeps=16)[0]
    e1 = np.abs(ob.plant_rk4(d, x, u, h)[0] - fine).max()
    e2 = np.abs(ob.plant_rk4(d, x, u, h / 2, substeps=2)[0] - fine).max()
    assert e1 < 1e-3 and e2 < e1 / 10.0       # 4th order: halving the step divides the error by ~16
    # hover thrust holds altitude: total thrust = m g, level, at rest (arm torques = gravity torques are zero when the
    # arm hangs along gravity is not guaranteed, so only the vertical base acceleration is checked to first order)
    mass = sum(d.model.mass[i] for i in range(d.model.nbodies))
    x0 = np.zeros(d.nx)
    x0[6] = 1.0
    uh = np.concatenate([np.full(d.n_rotors, mass * 9.81 / d.n_rotors), np.zeros(d.nu - d.n_rotors)])
    xn = ob.plant_rk4(d, x0, uh, 1e-3)[0]
    # momentum balance: total vertical momentum change = (thrust - m g) dt = 0
    assert abs(xn[d.model.nq + 2]) < 5e-3


def test_oracle_tracks_carrot(empc):
    """Sanity of the whole MPC construction on the CPU oracle: from the reference's first state the receding-horizon
    solve keeps the cost finite and pulls the last knot towards its carrot."""
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path(ARM3_TRAJ))
    ref, dt_ref = make_reference(traj)
    ref[:, 7] = 0.0
    ref[:, traj.nx - 1] = 0.0
    m = empc.CarrotMpc(traj, ref, dt_ref, empc.yaml_path(ARM3_MPC))
    m.updateProblem(0)
    m.problem.x0 = ref[0]
    d = m.problem.desc
    s = ob.OracleSolver(d)
    s.solve(None, None, 20, False)
    r = s.result()
    assert math.isfinite(r["cost"])
    target = m.computeStateReference(29 * 30)
    assert abs(r["xs"][-1][0] - target[0]) < abs(ref[0][0] - target[0])
