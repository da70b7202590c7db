"""RailMpc / WeightedMpc host logic (no GPU needed).

Expected values are derived by hand from the reference's rules:
  src/mpc-controllers/rail-mpc.cpp:14-60 (parameters), :128-149 (cost table), :151-174 (update), :176-200 (state
  reference: integer alpha inside the samples, yaw-only hover beyond them);
  src/mpc-controllers/weighted-mpc.cpp:57-69 (transition stages merged), :145-168 (cost table = every stage's costs,
  inactive), :170-199 (active stage, never skipping one between two knots), :203-243 (activation and weights).
"""
import bisect
import math

import numpy as np
import pytest

import oracle_binding as ob

ARM3_TRAJ = "hexacopter370_flying_arm_3/trajectories/displacement.yaml"
ARM3_MPC = "hexacopter370_flying_arm_3/mpc/mpc.yaml"


def make_reference(nx, x0, n=101):
    ref = np.tile(x0, (n, 1))
    ref[:, 0] = np.linspace(0.0, 2.0, n)
    ref[:, 7] = 0.01 * np.arange(n)
    ref[:, nx - 1] = 0.001 * np.arange(n)
    return ref


def cost_table(desc, knot):
    st = desc.sets[desc.knot_set[knot]]
    return {st.costs[i].name.decode(): st.costs[i] for i in range(st.ncosts)}


@pytest.fixture(scope="module")
def traj(empc):
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(ARM3_TRAJ))
    return t


# ---- RailMpc ----------------------------------------------------------------------------------------------

def test_rail_parameters_and_costs(empc, traj):
    ref = make_reference(traj.nx, traj.initial_state)
    m = empc.RailMpc(ref, 80, empc.yaml_path(ARM3_MPC))
    assert (m.knots, m.iters, m.dt) == (30, 2, 30)
    assert (m.nx, m.ndx, m.nu) == (traj.nx, traj.ndx, traj.nu)
    d = m.problem.desc
    assert d.T == 29 and d.n_sets == 30 and [d.knot_set[i] for i in range(30)] == list(range(30))
    assert d.use_squash == 1 and d.has_contact == 0 and d.dt == pytest.approx(0.03)
    zero = np.zeros(d.nx)
    zero[6] = 1.0
    assert np.array_equal(np.array([d.x0[i] for i in range(d.nx)]), zero)
    T = empc.T
    for knot in (0, 17, 29):
        c = cost_table(d, knot)
        assert sorted(c) == ["control", "rail_state"]
        assert c["rail_state"].weight == 100 and c["rail_state"].active == 1      # rail_weight
        assert c["rail_state"].type == T.COST_STATE and c["rail_state"].activation == T.ACT_WEIGHTED_QUAD
        assert [c["rail_state"].act_w[i] for i in range(d.ndx)] == [80] * 3 + [1] * 3 + [80] * 3 + [1] * 9
        assert c["control"].weight == 1e-2 and c["control"].active == 1           # rail_control_weight
        assert c["control"].type == T.COST_CONTROL and c["control"].activation == T.ACT_QUAD
        assert c["control"].nr == d.nu


def expected_rail_reference(ref, dt_ref, t, nq):
    t_ref = [dt_ref * i for i in range(len(ref))]
    idx = bisect.bisect_right(t_ref, t)
    if idx >= len(ref):
        x = np.zeros(ref.shape[1])
        x[:nq] = ref[-1, :nq]
        n = math.sqrt(ref[-1, 6] ** 2 + ref[-1, 5] ** 2)
        x[5], x[6] = ref[-1, 5] / n, ref[-1, 6] / n
        return x
    return ref[idx - 1].copy()


def test_rail_state_reference(empc, traj):
    nq = traj.nx - traj.ndx // 2
    ref = make_reference(traj.nx, traj.initial_state)
    # give the last sample a tilted attitude so that the yaw-only rule is visible
    q = np.array([0.1, -0.2, 0.3, 0.9])
    ref[-1, 3:7] = q / np.linalg.norm(q)
    m = empc.RailMpc(ref, 80, empc.yaml_path(ARM3_MPC))
    for t in [0, 1, 79, 80, 81, 7999, 8000, 8001, 30000]:
        assert np.allclose(m.computeStateReference(t), expected_rail_reference(ref, 80, t, nq), rtol=0, atol=1e-16), t
    assert np.array_equal(m.computeStateReference(79), ref[0])         # size_t quotient: no interpolation
    tail = m.computeStateReference(9000)
    assert not tail[nq:].any()                                          # hover: zero velocities
    assert np.array_equal(tail[3:5], ref[-1, 3:5])                      # qx, qy are NOT reset (reference quirk)
    assert tail[5] ** 2 + tail[6] ** 2 == pytest.approx(1.0)            # (qz, qw) renormalised on their own
    assert tail[5] / tail[6] == pytest.approx(ref[-1, 5] / ref[-1, 6])  # yaw kept


def test_rail_update_problem(empc, traj):
    nq = traj.nx - traj.ndx // 2
    ref = make_reference(traj.nx, traj.initial_state)
    m = empc.RailMpc(ref, 80, empc.yaml_path(ARM3_MPC))
    for t in [0, 50, 1234, 7500, 7990, 9000]:
        m.updateProblem(t)
        d = m.problem.desc
        for i in range(m.knots):
            c = cost_table(d, i)["rail_state"]
            got = np.array([c.ref[k] for k in range(traj.nx)])
            assert np.allclose(got, expected_rail_reference(ref, 80, t + i * m.dt, nq), rtol=0, atol=1e-16), (t, i)
            assert c.active == 1 and c.weight == 100


def test_rail_defaults_and_errors(empc, traj, tmp_path):
    ref = make_reference(traj.nx, traj.initial_state, n=5)
    text = open(empc.yaml_path(ARM3_MPC)).read()
    text = text.replace('follow: "hexacopter370_flying_arm_3/platform/hexacopter370.yaml"',
                        'follow: "%s"' % empc.yaml_path("hexacopter370_flying_arm_3/platform/hexacopter370.yaml"))
    lines = [ln for ln in text.splitlines() if not ln.strip().startswith("rail_")]
    p = tmp_path / "mpc_defaults.yaml"
    p.write_text("\n".join(lines) + "\n")
    m = empc.RailMpc(ref, 80, str(p))
    c = cost_table(m.problem.desc, 0)
    assert c["rail_state"].weight == 10 and c["control"].weight == 1e-1     # rail-mpc.cpp:30,56
    assert [c["rail_state"].act_w[i] for i in range(m.ndx)] == [1.0] * m.ndx
    bad = tmp_path / "mpc_bad.yaml"
    bad.write_text(text.replace("rail_activation_weights: [80, 80, 80, 1, 1, 1, 80, 80, 80, 1, 1, 1, 1, 1, 1, 1, 1, 1]",
                                "rail_activation_weights: [80, 80, 80]"))
    with pytest.raises(empc.EmpcError, match="dimension"):
        empc.RailMpc(ref, 80, str(bad))
    with pytest.raises(empc.EmpcError):
        empc.RailMpc(ref[:, :5], 80, empc.yaml_path(ARM3_MPC))             # wrong nx
    with pytest.raises(empc.EmpcError):
        m.updateProblem(-1)


def test_oracle_follows_the_rail(empc, traj):
    """The rail problem solved by the CPU oracle tracks a slow straight-line reference."""
    ref = np.tile(traj.initial_state, (101, 1))
    ref[:, 0] += np.linspace(0.0, 0.5, 101)
    m = empc.RailMpc(ref, 80, empc.yaml_path(ARM3_MPC))
    m.updateProblem(0)
    m.problem.x0 = ref[0]
    s = ob.OracleSolver(m.problem.desc)
    s.solve(None, None, 30, False)
    r = s.result()
    assert math.isfinite(r["cost"])
    target = m.computeStateReference(29 * 30)
    assert abs(r["xs"][-1][0] - target[0]) < 0.5 * abs(ref[0][0] - target[0])


# ---- WeightedMpc ------------------------------------------------------------------------------------------

@pytest.fixture()
def weighted(empc):
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(ARM3_TRAJ))
    before = [t.stage_info(i) for i in range(t.n_stages)]
    return t, before, empc.WeightedMpc(t, 80, empc.yaml_path(ARM3_MPC))


def test_weighted_merges_transition_stages(weighted):
    t, before, m = weighted
    # nav_wp1 (2000, transition) + wp_1 (0) -> wp_1 lasting 2000 ms from t = 0, and so on
    assert [s["name"] for s in before] == ["nav_wp1", "wp_1", "nav_wp2", "wp_2", "nav_wp3", "wp_3", "nav_wp4", "wp_4"]
    after = [t.stage_info(i) for i in range(t.n_stages)]
    assert [s["name"] for s in after] == [s["name"] for s in before if not s["is_transition"]]
    assert [s["duration"] for s in after] == [2000] * len(after)
    assert [s["t_ini"] for s in after] == [2000 * i for i in range(len(after))]
    assert m.t_stages == [2000 * i for i in range(len(after))]
    assert (m.knots, m.iters, m.dt) == (30, 2, 30)


def test_weighted_cost_table(weighted, empc):
    t, before, m = weighted
    d = m.problem.desc
    names = sorted(s["name"] + "/" + c["name"] for s in [t.stage_info(i) for i in range(t.n_stages)] for c in s["costs"])
    assert len(names) == 3 + 3 + 3 + 6
    for knot in (0, 29):
        c = cost_table(d, knot)
        assert sorted(c) == names                   # every stage's task costs in every knot
        assert all(v.active == 0 for v in c.values())   # added inactive (weighted-mpc.cpp:162-163)
    assert d.model.nframes >= 1                         # the base-link frame of the placement / motion costs


def expected_weighted(stages, t_stages, duration, knots, dt, alpha, beta, t):
    """active stage and weight factor per knot (weighted-mpc.cpp:170-199, 230-243)"""
    out = []
    last = bisect.bisect_right(t_stages, t) - 1
    for i in range(knots):
        nt = t + i * dt
        idx = bisect.bisect_right(t_stages, nt) - 1
        if idx == last + 2:
            idx -= 1
        wt = 0.0 if nt > duration else (nt - (stages[idx]["t_ini"] + stages[idx]["duration"])) / 1000.0
        out.append((idx, math.exp(alpha * wt) * beta))
        last = idx
    return out


def test_weighted_update_rules(weighted):
    t, before, m = weighted
    stages = [t.stage_info(i) for i in range(t.n_stages)]
    base = {s["name"]: {c["name"]: c["weight"] for c in s["costs"]} for s in stages}
    duration = sum(s["duration"] for s in before)
    for now in [0, 30, 1500, 1990, 2000, 3900, duration - 100, duration, duration + 500]:
        m.updateProblem(now)
        d = m.problem.desc
        exp = expected_weighted(stages, m.t_stages, duration, m.knots, m.dt, 3.0, 0.01, now)
        for i in range(m.knots):
            idx, factor = exp[i]
            name = stages[idx]["name"]
            for full, c in cost_table(d, i).items():
                st, cn = full.split("/")
                if st == name:
                    assert c.active == 1, (now, i, full)
                    if cn.startswith("reg") or cn.startswith("limits"):   # "/reg*", "/limits*" keep the stage weight
                        assert c.weight == base[st][cn], (now, i, full)
                    else:
                        assert c.weight == pytest.approx(base[st][cn] * factor, rel=1e-15), (now, i, full)
                else:
                    assert c.active == 0, (now, i, full)
    # at the end of a stage the factor is beta, one second earlier beta * exp(-alpha)
    m.updateProblem(2000 - 29 * 30)
    last = cost_table(m.problem.desc, 29)
    first = cost_table(m.problem.desc, 0)
    w0 = base["wp_1"]["state_arm"]
    assert last["wp_2/state_arm"].active == 1 and last["wp_1/state_arm"].active == 0   # node time 2000 -> stage 2
    assert first["wp_1/state_arm"].weight == pytest.approx(w0 * 0.01 * math.exp(3.0 * (-870 / 1000.0)))


def test_weighted_rejects_contacts_and_bad_input(empc):
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml"))
    with pytest.raises(empc.EmpcError, match="Weighted with contact has not been implemented"):
        empc.WeightedMpc(t, 10, empc.yaml_path(ARM3_MPC))
    t2 = empc.Trajectory()
    t2.autoSetup(empc.yaml_path(ARM3_TRAJ))
    with pytest.raises(empc.EmpcError):
        empc.WeightedMpc(t2, 10, "/nonexistent/mpc.yaml")


def test_oracle_solves_weighted_problem(weighted):
    t, before, m = weighted
    m.updateProblem(1500)
    m.problem.x0 = t.initial_state
    s = ob.OracleSolver(m.problem.desc)
    s.solve(None, None, 10, False)
    r = s.result()
    assert math.isfinite(r["cost"]) and np.isfinite(r["xs"]).all()
