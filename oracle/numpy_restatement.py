"""ORACLE, second restatement (test infrastructure only -- never imported by the product path, bench.py's timed region or
the package; used by tests/test_second_restatement.py and tests/golden/make_second_restatement.py).

Purpose: oracle/liboracle.so and the HIP kernels were written
by one author from one specification.  This file restates the per-node arithmetic a THIRD time, in
NumPy, along routes chosen to share as little as possible with the C++ oracle:

  quantity                         C++ oracle (oracle/*.hpp)                          here
  forward dynamics (free)          RNEA bias + CRBA + Cholesky                        articulated-body algorithm (Featherstone ABA)
  forward dynamics (contact)       Schur complement on M^-1 via two Cholesky's        dense KKT matrix [[M, J^T], [J, 0]], np.linalg.solve,
                                                                                      M from unit-acceleration RNEA columns
  every first derivative           forward-mode dual numbers through the recursions   complex-step differentiation of the VALUE code
  (Fx, Fu, Lx, Lu, Rx, Ru)         + hand-written Lie-group Jacobians (Jexp6, Jlog6)  (h = 1e-30: exact to rounding, no Jacobian formulas at all)
  SE(3) log                        closed-form V^-1 coefficients                      np.linalg.solve(V(w), p)
  backward pass                    hand-written loops + own Cholesky                  np.linalg (cholesky, solve)

What it restates (reference call sites; the arithmetic itself lives in the un-vendored Crocoddyl fork / Pinocchio):
  node(x, u)      IntegratedActionModelEuler / RK4 (src/factory/int-action.cpp:26-31) over DifferentialActionModel
                  {Free,Contact}FwdDynamics (src/factory/diff-action.cpp:31,34) with ActuationSquashingModel
                  (src/trajectory.cpp:47-52), CostModelSum of the residual / activation types of src/factory/cost.cpp:38-168,
                  src/factory/activation.cpp:35-96, ContactModel3D / 6D (src/factory/contacts.cpp:49-79)  -- SURVEY A.3-A.7
  backward_pass   crocoddyl SolverDDP::backwardPass + computeGains (called at src/sbfddp.cpp:244,332) -- SURVEY A.2

Conventions: q = [p, quat xyzw, theta], v = [v_lin, omega (body frame), theta_dot], spatial order [lin; ang].
Everything is written for complex arguments (no abs / conj / real-only branches on values that carry the imaginary seed).
"""
import numpy as np

H = 1e-30  # complex step


def _c(a):
    return np.asarray(a, dtype=complex)


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=complex)


def quat_to_R(q):
    x, y, z, w = q
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]], dtype=complex)


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], dtype=complex)


def quat_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]], dtype=complex)


def _small(t):
    return abs(complex(t).real) < 1e-4 and abs(complex(t).imag) < 1e-4


def quat_exp3(w):
    w = _c(w)
    t2 = w @ w
    t = np.sqrt(t2)
    if _small(t):
        a = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0       # sin(t/2)/t
        c = 1.0 - t2 / 8.0 + t2 * t2 / 384.0         # cos(t/2)
    else:
        a = np.sin(t / 2) / t
        c = np.cos(t / 2)
    return np.array([a * w[0], a * w[1], a * w[2], c], dtype=complex)


def quat_log3(q):
    q = _c(q)
    q = q / np.sqrt(q @ q)
    if q[3].real < 0:
        q = -q
    v = q[:3]
    n2 = v @ v
    n = np.sqrt(n2)
    if _small(n):
        f = 2.0 / q[3] * (1.0 - n2 / (3.0 * q[3] * q[3]))  # 2 atan(n/w)/n, series
    else:
        half = np.arctan(n / q[3]) if q[3].real > n.real else (np.pi / 2 - np.arctan(q[3] / n))
        f = 2.0 * half / n
    return f * v


def R_to_quat(R):
    """rotation matrix -> unit quaternion (xyzw); branch on real parts only"""
    R = _c(R)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr.real > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0].real > R[1, 1].real and R[0, 0].real > R[2, 2].real:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1].real > R[2, 2].real:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    return np.array(q, dtype=complex)


def V_matrix(w):
    """SE(3): exp6([v; w]) has translation V(w) v"""
    w = _c(w)
    t2 = w @ w
    t = np.sqrt(t2)
    if _small(t):
        B = 0.5 - t2 / 24.0 + t2 * t2 / 720.0
        C = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0
    else:
        B = (1 - np.cos(t)) / t2
        C = (t - np.sin(t)) / (t2 * t)
    W = skew(w)
    return np.eye(3) + B * W + C * (W @ W)


def log6(R, p):
    w = quat_log3(R_to_quat(R))
    v = np.linalg.solve(V_matrix(w), _c(p))
    return np.concatenate([v, w])


def log6_quat(q, p):
    w = quat_log3(q)
    v = np.linalg.solve(V_matrix(w), _c(p))
    return np.concatenate([v, w])


class Model:
    def __init__(self, desc):
        m = desc.model
        self.nb, self.nq, self.nv = m.nbodies, m.nq, m.nv
        self.parent = [m.parent[b] for b in range(self.nb)]
        self.jR = [np.array(m.jplace_R[b][:]).reshape(3, 3) for b in range(self.nb)]
        self.jp = [np.array(m.jplace_p[b][:]) for b in range(self.nb)]
        self.axis = [np.array(m.axis[b][:]) for b in range(self.nb)]
        self.mass = [m.mass[b] for b in range(self.nb)]
        self.com = [np.array(m.com[b][:]) for b in range(self.nb)]
        self.inertia = [np.array(m.inertia[b][:]).reshape(3, 3) for b in range(self.nb)]
        self.frame_body = [m.frame_body[f] for f in range(m.nframes)]
        self.fR = [np.array(m.frame_R[f][:]).reshape(3, 3) for f in range(m.nframes)]
        self.fp = [np.array(m.frame_p[f][:]) for f in range(m.nframes)]
        self.g = np.array(m.gravity[:])
        # spatial inertia about the body origin, order [lin; ang]
        self.I6 = []
        for b in range(self.nb):
            C = skew(self.com[b]).real
            I = np.zeros((6, 6))
            I[:3, :3] = self.mass[b] * np.eye(3)
            I[:3, 3:] = -self.mass[b] * C
            I[3:, :3] = self.mass[b] * C
            I[3:, 3:] = self.inertia[b] - self.mass[b] * C @ C
            self.I6.append(I)

    def S(self, b):
        s = np.zeros(6)
        s[3:] = self.axis[b]
        return s


def crm(v):
    """motion cross product matrix, order [lin; ang]: v x m"""
    M = np.zeros((6, 6), dtype=complex)
    M[:3, :3] = skew(v[3:])
    M[:3, 3:] = skew(v[:3])
    M[3:, 3:] = skew(v[3:])
    return M


def crf(v):
    return -crm(v).T


def kinematics(md, q, v=None, a=None, gravity=False):
    """placements (R, p world), parent->child motion transforms X, body-frame spatial velocities / accelerations"""
    q = _c(q)
    nb = md.nb
    R, p, X = [None] * nb, [None] * nb, [None] * nb
    R[0] = quat_to_R(q[3:7])
    p[0] = q[:3]
    vel, acc = [None] * nb, [None] * nb
    if v is not None:
        v = _c(v)
        vel[0] = v[:6]
        a = np.zeros(md.nv, dtype=complex) if a is None else _c(a)
        acc[0] = a[:6].copy()
        if gravity:
            acc[0][:3] = acc[0][:3] + R[0].T @ (-md.g)
    for b in range(1, nb):
        th = q[7 + b - 1]
        ax = md.axis[b]
        K = skew(ax)
        Rj = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        XR = md.jR[b] @ Rj          # child -> parent rotation
        E = XR.T
        par = md.parent[b]
        R[b] = R[par] @ XR
        p[b] = p[par] + R[par] @ md.jp[b]
        Xb = np.zeros((6, 6), dtype=complex)
        Xb[:3, :3] = E
        Xb[:3, 3:] = -E @ skew(md.jp[b])
        Xb[3:, 3:] = E
        X[b] = Xb
        if v is not None:
            Sq = md.S(b) * v[6 + b - 1]
            vel[b] = Xb @ vel[par] + Sq
            acc[b] = Xb @ acc[par] + md.S(b) * a[6 + b - 1] + crm(vel[b]) @ Sq
    return R, p, X, vel, acc


def rnea(md, q, v, a, fext=None, gravity=True):
    R, p, X, vel, acc = kinematics(md, q, v, a, gravity)
    f = [None] * md.nb
    for b in range(md.nb):
        f[b] = md.I6[b] @ acc[b] + crf(vel[b]) @ (md.I6[b] @ vel[b])
        if fext is not None and fext[b] is not None:
            f[b] = f[b] - fext[b]
    tau = np.zeros(md.nv, dtype=complex)
    for b in range(md.nb - 1, 0, -1):
        tau[6 + b - 1] = md.S(b) @ f[b]
        f[md.parent[b]] = f[md.parent[b]] + X[b].T @ f[b]
    tau[:6] = f[0]
    return tau


def aba(md, q, v, tau, fext=None):
    """articulated-body algorithm for the free-flyer + revolute chain: generalized acceleration"""
    R, p, X, vel, _ = kinematics(md, q, v, None, False)
    nb = md.nb
    tau = _c(tau)
    IA = [md.I6[b].astype(complex) for b in range(nb)]
    pA = [None] * nb
    c = [None] * nb
    for b in range(nb):
        pA[b] = crf(vel[b]) @ (md.I6[b] @ vel[b])
        if fext is not None and fext[b] is not None:
            pA[b] = pA[b] - fext[b]
        if b > 0:
            c[b] = crm(vel[b]) @ (md.S(b) * _c(v)[6 + b - 1])
    U, D, u = [None] * nb, [None] * nb, [None] * nb
    for b in range(nb - 1, 0, -1):
        S = md.S(b)
        U[b] = IA[b] @ S
        D[b] = S @ U[b]
        u[b] = tau[6 + b - 1] - S @ pA[b]
        Ia = IA[b] - np.outer(U[b], U[b]) / D[b]
        pa = pA[b] + Ia @ c[b] + U[b] * (u[b] / D[b])
        par = md.parent[b]
        IA[par] = IA[par] + X[b].T @ Ia @ X[b]
        pA[par] = pA[par] + X[b].T @ pa
    acc = [None] * nb
    a0 = np.linalg.solve(IA[0], tau[:6] - pA[0])   # includes the fictitious -g of the base
    acc[0] = a0
    qdd = np.zeros(md.nv, dtype=complex)
    ag = np.zeros(6, dtype=complex)
    ag[:3] = R[0].T @ (-md.g)
    qdd[:6] = a0 - ag
    for b in range(1, nb):
        ab = X[b] @ acc[md.parent[b]] + c[b]
        qdd[6 + b - 1] = (u[b] - U[b] @ ab) / D[b]
        acc[b] = ab + md.S(b) * qdd[6 + b - 1]
    return qdd


def mass_matrix(md, q):
    nv = md.nv
    z = np.zeros(nv)
    h0 = rnea(md, q, z, z, gravity=False)
    M = np.zeros((nv, nv), dtype=complex)
    for j in range(nv):
        e = np.zeros(nv)
        e[j] = 1.0
        M[:, j] = rnea(md, q, z, e, gravity=False) - h0
    return M


def frame_kin(md, f, R, p, vel, acc):
    b = md.frame_body[f]
    Rf = R[b] @ md.fR[f]
    pf = p[b] + R[b] @ md.fp[f]
    Xf = np.zeros((6, 6), dtype=complex)
    Xf[:3, :3] = md.fR[f].T
    Xf[:3, 3:] = -md.fR[f].T @ skew(md.fp[f])
    Xf[3:, 3:] = md.fR[f].T
    vf = Xf @ vel[b] if vel is not None else None
    af = Xf @ acc[b] if acc is not None and acc[b] is not None else None
    return Rf, pf, vf, af, Xf


# ---- state manifold ---------------------------------------------------------------------------------------------------
def state_integrate(nq, x, dx):
    x, dx = _c(x), _c(dx)
    nv = len(dx) // 2
    R = quat_to_R(x[3:7])
    out = x.copy()
    out[:3] = x[:3] + R @ (V_matrix(dx[3:6]) @ dx[:3])
    qn = quat_mul(x[3:7], quat_exp3(dx[3:6]))
    out[3:7] = qn / np.sqrt(qn @ qn)
    out[7:nq] = x[7:nq] + dx[6:nv]
    out[nq:] = x[nq:] + dx[nv:]
    return out


def state_diff(nq, x0, x1):
    x0, x1 = _c(x0), _c(x1)
    nv = len(x0) - nq
    R0 = quat_to_R(x0[3:7])
    qd = quat_mul(quat_conj(x0[3:7]), x1[3:7])
    d6 = log6_quat(qd, R0.T @ (x1[:3] - x0[:3]))
    return np.concatenate([d6, x1[7:nq] - x0[7:nq], x1[nq:] - x0[nq:]])


# ---- one node ---------------------------------------------------------------------------------------------------------
class Problem:
    def __init__(self, desc, prm, sets=None):
        self.d, self.prm = desc, prm
        self.md = Model(desc)
        self.nq, self.nv, self.nx, self.ndx, self.nu = desc.model.nq, desc.model.nv, desc.nx, desc.ndx, desc.nu
        self.nrot = desc.n_rotors
        self.tau_f = np.array(desc.tau_f[:6 * self.nrot]).reshape(6, self.nrot)
        self.lb = np.array(desc.u_lb[:self.nu])
        self.ub = np.array(desc.u_ub[:self.nu])
        self.dt = desc.dt
        self.sets = sets  # list of cost sets as plain dicts (see cost_sets_of)


def cost_sets_of(desc, prm, smooth):
    """cost / contact tables as plain Python data, with the solver's barrier cost (src/sbfddp.cpp:169-190, 464-477) inserted
    in name order into every set a running knot uses"""
    nu = desc.nu
    running = set(desc.knot_set[t] for t in range(desc.T))
    out = []
    for si in range(desc.n_sets):
        s = desc.sets[si]
        costs = []
        for i in range(s.ncosts):
            c = s.costs[i]
            costs.append(dict(name=c.name.decode(), type=c.type, activation=c.activation, active=c.active, frame=c.frame, nr=c.nr,
                              weight=c.weight, ref=np.array(c.ref[:]), act_w=np.array(c.act_w[:]), lb=np.array(c.lb[:]),
                              ub=np.array(c.ub[:])))
        if prm.solver_type == 0 and si in running and not any(c["name"] == "barrier" for c in costs):
            w = 1.0 / (smooth * (np.array(desc.u_ub[:nu]) - np.array(desc.u_lb[:nu]))) ** 2
            costs.append(dict(name="barrier", type=1, activation=3, active=1, frame=-1, nr=nu, weight=prm.barrier_weight,
                              ref=np.zeros(29), act_w=np.concatenate([w, np.ones(28 - nu)]),
                              lb=np.concatenate([np.array(desc.u_lb[:nu]), np.zeros(28 - nu)]),
                              ub=np.concatenate([np.array(desc.u_ub[:nu]), np.zeros(28 - nu)])))
            costs.sort(key=lambda c: c["name"])
        contacts = []
        for i in range(s.ncontacts):
            ct = s.contacts[i]
            contacts.append(dict(type=ct.type, frame=ct.frame, ref_p=np.array(ct.ref_p[:]), ref_R=np.array(ct.ref_R[:]).reshape(3, 3),
                                 gains=np.array(ct.gains[:])))
        out.append(dict(costs=costs, contacts=contacts))
    return out


def squash(P, s, smooth):
    d = smooth * (P.ub - P.lb)
    a = d ** 4 if P.prm.smoothsat_power == 4 else d ** 2
    return 0.5 * (np.sqrt((s - P.lb) ** 2 + a) - np.sqrt((s - P.ub) ** 2 + a) + P.ub + P.lb)


def activation(c, r):
    """value, Ar, diag(Arr) -- branches on real parts"""
    nr = c["nr"]
    r = _c(r[:nr])
    t = c["activation"]
    if t == 0:
        return 0.5 * (r @ r), r, np.ones(nr)
    if t == 1:
        w = c["act_w"][:nr]
        return 0.5 * ((w * r) @ r), w * r, w
    w = np.ones(nr) if t == 2 else c["act_w"][:nr]
    lo = np.where((r - c["lb"][:nr]).real < 0, r - c["lb"][:nr], 0)
    hi = np.where((r - c["ub"][:nr]).real > 0, r - c["ub"][:nr], 0)
    ind = ((r - c["lb"][:nr]).real <= 0).astype(float) + ((r - c["ub"][:nr]).real >= 0).astype(float)
    return 0.5 * ((w * lo) @ lo) + 0.5 * ((w * hi) @ hi), w * (lo + hi), w * ind


def cone_matrix(c):
    mu = c["ref"][3]
    A = np.array([[1, 0, -mu], [0, 1, -mu], [-1, 0, -mu], [0, -1, -mu], [0, 0, 1.0]])
    n = c["ref"][:3] / np.linalg.norm(c["ref"][:3])
    e3 = np.array([0, 0, 1.0])
    ax = np.cross(e3, n)
    sn, cs = np.linalg.norm(ax), n[2]
    Rn = np.eye(3)
    if sn > 1e-12:
        k = ax / sn
        K = skew(k).real
        ang = np.arctan2(sn, cs)
        Rn = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    elif cs < 0:
        Rn = np.diag([1.0, -1.0, -1.0])
    return A @ Rn.T


def dam(P, cset, x, s, smooth, terminal):
    """differential action model value: acceleration, contact force, list of (cost, residual) pairs, unscaled cost"""
    md = P.md
    x, s = _c(x), _c(s)
    q, v = x[:P.nq], x[P.nq:]
    u = squash(P, s, smooth) if P.d.use_squash else s
    tau = np.concatenate([P.tau_f @ u[:P.nrot], u[P.nrot:]])
    use_contact = bool(P.d.has_contact) and len(cset["contacts"]) > 0
    lam = np.zeros(12, dtype=complex)
    coff = []
    R, p, X, vel, acc0 = kinematics(md, q, v, np.zeros(P.nv), gravity=False)
    if not use_contact:
        a = aba(md, q, v, tau)
    else:
        # ContactModelMultiple (src/stage.cpp:38-48): the rows of every contact of the stage stacked in the stage's order
        Js, a0s = [], []
        unit_vel = []
        for j in range(P.nv):
            e = np.zeros(P.nv)
            e[j] = 1.0
            unit_vel.append(kinematics(md, q, e, np.zeros(P.nv), gravity=False)[3])
        for ct in cset["contacts"]:
            nc = 3 if ct["type"] == 0 else 6
            coff.append(sum(len(x_) for x_ in a0s))
            Rf, pf, vf, af, Xf = frame_kin(md, ct["frame"], R, p, vel, acc0)
            J = np.zeros((nc, P.nv), dtype=complex)
            for j in range(P.nv):
                J[:, j] = (Xf @ unit_vel[j][md.frame_body[ct["frame"]]])[:nc]
            a0 = af[:nc].copy()
            if nc == 3:
                a0 = a0 + np.cross(vf[3:], vf[:3])
            if ct["gains"][0] != 0.0:
                if nc == 3:
                    a0 = a0 + ct["gains"][0] * (pf - ct["ref_p"])
                else:
                    a0 = a0 + ct["gains"][0] * log6(ct["ref_R"].T @ Rf, ct["ref_R"].T @ (pf - ct["ref_p"]))
            if ct["gains"][1] != 0.0:
                a0 = a0 + ct["gains"][1] * vf[:nc]
            Js.append(J)
            a0s.append(a0)
        J = np.vstack(Js)
        a0 = np.concatenate(a0s)
        nc = len(a0)
        M = mass_matrix(md, q)
        h = rnea(md, q, v, np.zeros(P.nv))
        KKT = np.zeros((P.nv + nc, P.nv + nc), dtype=complex)
        KKT[:P.nv, :P.nv] = M
        KKT[:P.nv, P.nv:] = J.T
        KKT[P.nv:, :P.nv] = J
        sol = np.linalg.solve(KKT, np.concatenate([tau - h, -a0]))
        a = sol[:P.nv]
        lam[:nc] = -sol[P.nv:]
    # costs
    ell = 0
    items = []
    for c in cset["costs"]:
        if not c["active"]:
            continue
        t = c["type"]
        if t == 0:
            r = state_diff(P.nq, c["ref"][:P.nx], x)
        elif t == 1:
            r = s - c["ref"][:P.nu]
        elif t in (2, 3, 4, 5):
            Rf, pf, vf, _, _ = frame_kin(md, c["frame"], R, p, vel, None)
            if t == 2:
                Rr = c["ref"][3:12].reshape(3, 3)
                r = log6(Rr.T @ Rf, Rr.T @ (pf - c["ref"][:3]))
            elif t == 3:
                Rr = c["ref"][:9].reshape(3, 3)
                r = quat_log3(R_to_quat(Rr.T @ Rf))
            elif t == 5:
                r = pf - c["ref"][:3]
            else:
                r = vf - c["ref"][:6]
        else:
            fo = 0  # the force of the contact on the cost's frame (one contact: that contact, whatever the frame)
            if use_contact and len(cset["contacts"]) > 1:
                for k, ct in enumerate(cset["contacts"]):
                    if ct["frame"] == c["frame"]:
                        fo = coff[k]
            r = cone_matrix(c) @ lam[fo:fo + 3] if use_contact else np.zeros(5, dtype=complex)
        val, _, _ = activation(c, r)
        ell = ell + c["weight"] * val
        items.append((c, _c(r)))
    return a, lam, ell, items, u


def node_value(P, cset, x, s, smooth, terminal):
    """IAM.calc: xnext, cost (scaled), acceleration, contact force; s = None at the terminal node (u = 0)"""
    s_ = np.zeros(P.nu) if s is None else s
    cscale_dt = 1.0 if (terminal and not P.prm.terminal_dt_scaling) else P.dt
    if P.d.integrator == 0:
        a, lam, ell, items, u = dam(P, cset, x, s_, smooth, terminal)
        v = _c(x)[P.nq:]
        dx = np.concatenate([v * P.dt + a * P.dt ** 2, a * P.dt])
        return state_integrate(P.nq, x, dx), cscale_dt * ell, a, lam, u, [(1.0, items, _c(x))]
    c4, w4 = [0.0, 0.5, 0.5, 1.0], [1.0, 2.0, 2.0, 1.0]
    ks, stages, tot = [], [], 0
    a0 = lam0 = u0 = None
    for i in range(4):
        y = _c(x) if i == 0 else state_integrate(P.nq, x, c4[i] * P.dt * ks[i - 1])
        a, lam, ell, items, u = dam(P, cset, y, s_, smooth, terminal)
        if i == 0:
            a0, lam0, u0 = a, lam, u
        ks.append(np.concatenate([y[P.nq:], a]))
        stages.append((w4[i], items, y))
        tot = tot + w4[i] * ell
    dx = (ks[0] + 2 * ks[1] + 2 * ks[2] + ks[3]) * P.dt / 6.0
    return state_integrate(P.nq, x, dx), tot * cscale_dt / 6.0, a0, lam0, u0, stages


def node(P, cset, x, s, smooth):
    """IAM.calc + calcDiff by complex-step differentiation of node_value.  Returns a dict with the fields of the C++ oracle's
    NodeData.  Gauss-Newton Hessians: L** = sum_c w_c R*^T diag(Arr) R*, R* = complex-step Jacobians of the residuals (for RK4
    nodes per stage, chained with the stage states' Jacobians, second derivatives of the stage states dropped)."""
    terminal = s is None
    s0 = np.zeros(P.nu) if terminal else np.asarray(s, dtype=float)
    x = np.asarray(x, dtype=float)
    n, m = P.ndx, P.nu
    xn0, cost0, a0, lam0, u0, stages0 = node_value(P, cset, x, None if terminal else s0, smooth, terminal)
    xn0r = xn0.real
    scale = (1.0 if (terminal and not P.prm.terminal_dt_scaling) else P.dt) * (1.0 if P.d.integrator == 0 else 1.0 / 6.0)
    Fx, Fu = np.zeros((n, n)), np.zeros((n, m))
    Lx, Lu = np.zeros(n), np.zeros(m)
    nst = len(stages0)
    Rjac = [[(np.zeros((it[0]["nr"], n)), np.zeros((it[0]["nr"], m))) for it in st[1]] for st in stages0]
    for j in range(n + m):
        if j < n:
            e = np.zeros(n, dtype=complex)
            e[j] = 1j * H
            xj, sj = state_integrate(P.nq, x, e), s0.astype(complex)
        else:
            xj = x.astype(complex)
            sj = s0.astype(complex)
            sj[j - n] += 1j * H
        if terminal and j >= n:
            continue
        xn, cost, _, _, _, stages = node_value(P, cset, xj, sj if not terminal else None, smooth, terminal)
        if terminal:
            # terminal call: calc(x) == calc(x, u = 0); only x-derivatives exist
            pass
        d = state_diff(P.nq, xn0r, xn).imag / H
        if j < n:
            Fx[:, j] = d
            Lx[j] = cost.imag / H
        else:
            Fu[:, j - n] = d
            Lu[j - n] = cost.imag / H
        for si in range(nst):
            for ci, (c, r) in enumerate(stages[si][1]):
                col = r.imag[:c["nr"]] / H
                if j < n:
                    Rjac[si][ci][0][:, j] = col
                else:
                    Rjac[si][ci][1][:, j - n] = col
    Lxx, Lxu, Luu = np.zeros((n, n)), np.zeros((n, m)), np.zeros((m, m))
    for si in range(nst):
        wst = stages0[si][0]
        for ci, (c, r) in enumerate(stages0[si][1]):
            _, _, Arr = activation(c, r)
            Arr = np.real(Arr)
            Rx, Ru = Rjac[si][ci]
            w = c["weight"] * wst * scale
            Lxx += w * Rx.T @ (Arr[:, None] * Rx)
            Lxu += w * Rx.T @ (Arr[:, None] * Ru)
            Luu += w * Ru.T @ (Arr[:, None] * Ru)
    return dict(xnext=xn0r, cost=float(cost0.real), acc=a0.real, lam=lam0.real[:6], u_squash=np.real(u0), Fx=Fx, Fu=Fu, Lx=Lx, Lu=Lu,
                Lxx=Lxx, Lxu=Lxu, Luu=Luu)


# ---- SolverDDP::backwardPass + computeGains ---------------------------------------------------------------
def backward_pass(tapes, fs, xreg, feasible):
    """tapes: list of T+1 dicts (Fx, Fu, Lx, Lu, Lxx, Lxu, Luu); fs: (T+1) x ndx gaps.  Returns K, k, Vx, Vxx."""
    T = len(tapes) - 1
    n = tapes[0]["Lx"].shape[0]
    Vxx = [None] * (T + 1)
    Vx = [None] * (T + 1)
    K, k = [None] * T, [None] * T
    Vxx[T] = tapes[T]["Lxx"] + xreg * np.eye(n)
    Vx[T] = tapes[T]["Lx"].copy()
    if not feasible:
        Vx[T] = Vx[T] + Vxx[T] @ fs[T]
    for t in range(T - 1, -1, -1):
        d = tapes[t]
        Fx, Fu = d["Fx"], d["Fu"]
        Qxx = d["Lxx"] + Fx.T @ Vxx[t + 1] @ Fx
        Qxu = d["Lxu"] + Fx.T @ Vxx[t + 1] @ Fu
        Quu = d["Luu"] + Fu.T @ Vxx[t + 1] @ Fu + xreg * np.eye(Fu.shape[1])
        Qx = d["Lx"] + Fx.T @ Vx[t + 1]
        Qu = d["Lu"] + Fu.T @ Vx[t + 1]
        Lc = np.linalg.cholesky(Quu)
        K[t] = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Qxu.T))
        k[t] = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Qu))
        Vx[t] = Qx + K[t].T @ (Quu @ k[t]) - 2 * K[t].T @ Qu
        V = Qxx - Qxu @ K[t]
        Vxx[t] = 0.5 * (V + V.T) + xreg * np.eye(n)
        if not feasible:
            Vx[t] = Vx[t] + Vxx[t] @ fs[t]
    return np.array(K), np.array(k), np.array(Vx), np.array(Vxx)
