"""ctypes binding of the ORACLE (oracle/liboracle.so).  Test infrastructure: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import empc_loader  # noqa: E402

empc = empc_loader.load()
T = empc.T
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build_oracle():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_orc = None
_orc_variants = {}


def orc(variant=None):
    """the oracle library; variant="fma" loads the same sources built with FMA contraction (the rounding-noise yardstick of
    the teacher-forced tests, never the reference value)"""
    global _orc
    if variant == "native":
        # -O3 -march=native build for the TIMED cpu baseline only (oracle/Makefile): compiled on the machine it runs on
        if variant not in _orc_variants:
            path = os.path.join(ORACLE_DIR, "liboracle_native.so")
            try:
                here = [l for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except (OSError, IndexError):
                here = "unknown\n"
            stamp = os.path.join(ORACLE_DIR, "liboracle_native.cpu")
            if not os.path.exists(path) or not os.path.exists(stamp) or open(stamp).read() != here:
                subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B", "native"])
            _orc_variants[variant] = _bind(C.CDLL(path))
        return _orc_variants[variant]
    if variant is not None:
        if variant not in _orc_variants:
            path = os.path.join(ORACLE_DIR, "liboracle_%s.so" % variant)
            if not os.path.exists(path):
                build_oracle()
            _orc_variants[variant] = _bind(C.CDLL(path))
        return _orc_variants[variant]
    if _orc is None:
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        _orc = _bind(C.CDLL(ORACLE_LIB))
    return _orc


def _bind(L):
    L.oracle_solver_create.restype = C.c_void_p
    L.oracle_solver_create.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams)]
    L.oracle_solver_destroy.argtypes = [C.c_void_p]
    L.oracle_solver_set_x0.argtypes = [C.c_void_p, _dp]
    L.oracle_solver_set_convergence_init.argtypes = [C.c_void_p, C.c_double]
    L.oracle_solver_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int]
    L.oracle_solver_get.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
    L.oracle_solver_trace.argtypes = [C.c_void_p, _dp, C.c_int]
    L.oracle_solver_set_smooth.argtypes = [C.c_void_p, C.c_double]
    L.oracle_node_calc.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int] + [_dp] * 12
    L.oracle_phase_calcdiff.restype = C.c_double
    L.oracle_phase_calcdiff.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, _dp, _ip]
    L.oracle_phase_backward.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.oracle_phase_forward.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp, _dp]
    L.oracle_phase_tape.argtypes = [C.c_void_p, C.c_int] + [_dp] * 9
    L.oracle_phase_expected_ddp.argtypes = [C.c_void_p, _dp]
    L.oracle_box_qp.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _dp]
    L.oracle_solver_record_iterates.argtypes = [C.c_void_p, C.c_int]
    L.oracle_solver_n_iterates.argtypes = [C.c_void_p]
    L.oracle_solver_get_iterate.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, _dp, _dp]
    L.oracle_iter_probe.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp]
    L.oracle_iter_step.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _dp]
    L.oracle_get_gains.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.oracle_solve_batch.restype = C.c_double
    L.oracle_solve_batch.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams), C.c_int, _dp, C.c_int, C.c_int,
                                     _dp, _dp, _dp, _dp, _ip, _ip]
    L.oracle_state_integrate.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.oracle_state_diff.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.oracle_rnea.argtypes = [C.POINTER(T.ModelDesc), _dp, _dp, _dp, _dp]
    L.oracle_crba.argtypes = [C.POINTER(T.ModelDesc), _dp, _dp]
    L.oracle_plant_rk4.argtypes = [C.POINTER(T.ProblemDesc), _dp, _dp, C.c_double, _dp]
    L.oracle_energy.restype = C.c_double
    L.oracle_energy.argtypes = [C.POINTER(T.ModelDesc), _dp, _dp]
    for f in ("exp6", "log6"):
        getattr(L, "oracle_" + f).argtypes = [_dp, _dp, _dp]
    for f in ("Jexp6", "Jlog6", "exp3", "log3", "Jexp3", "Jlog3"):
        getattr(L, "oracle_" + f).argtypes = [_dp, _dp]
    return L


def P(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def default_params():
    p = T.SolverParams()
    orc().oracle_solver_params_default(C.byref(p))
    return p


class OracleSolver:
    def __init__(self, desc, params=None, variant=None):
        self.d = desc
        self.prm = params if params is not None else default_params()
        self.L = orc(variant)
        self.h = C.c_void_p(self.L.oracle_solver_create(C.byref(desc), C.byref(self.prm)))
        self.nx, self.ndx, self.nu, self.nv, self.T = desc.nx, desc.ndx, desc.nu, desc.model.nv, desc.T

    def __del__(self):
        try:
            self.L.oracle_solver_destroy(self.h)
        except Exception:
            pass

    def set_x0(self, x0):
        self.L.oracle_solver_set_x0(self.h, P(np.ascontiguousarray(x0, dtype=np.float64)))

    def set_smooth(self, s):
        self.L.oracle_solver_set_smooth(self.h, float(s))

    def solve(self, xs=None, us=None, maxiter=100, is_feasible=False):
        xs = None if xs is None else np.ascontiguousarray(xs, dtype=np.float64)
        us = None if us is None else np.ascontiguousarray(us, dtype=np.float64)
        return self.L.oracle_solver_solve(self.h, P(xs), P(us), int(maxiter), int(is_feasible))

    def result(self):
        xs = np.zeros((self.T + 1, self.nx))
        us = np.zeros((self.T, self.nu))
        usq = np.zeros((self.T, self.nu))
        cost = C.c_double()
        it = C.c_int()
        st = C.c_int()
        stop = C.c_double()
        self.L.oracle_solver_get(self.h, P(xs), P(us), P(usq), C.cast(C.byref(cost), _dp), C.byref(it), C.byref(st),
                                C.cast(C.byref(stop), _dp))
        return dict(xs=xs, us=us, us_squash=usq, cost=cost.value, iter=it.value, status=st.value, stop=stop.value)

    def trace(self):
        n = self.L.oracle_solver_trace(self.h, None, 0)
        tr = np.zeros((n, 12))
        self.L.oracle_solver_trace(self.h, P(tr), n)
        return tr

    def record_iterates(self, on=True):
        self.L.oracle_solver_record_iterates(self.h, int(on))

    def iterates(self):
        """the iterates of the last solve (record_iterates first): list of dicts with the candidate (xs, us, k), the solver
        scalars at the top of the iteration and the iteration's outcome"""
        out = []
        for i in range(self.L.oracle_solver_n_iterates(self.h)):
            xs = np.zeros((self.T + 1, self.nx))
            us = np.zeros((self.T, self.nu))
            k = np.zeros((self.T, self.nu))
            ints = np.zeros(9, dtype=np.int32)
            reals = np.zeros(5)
            self.L.oracle_solver_get_iterate(self.h, i, P(xs), P(us), ints.ctypes.data_as(_ip), P(reals), P(k))
            d = dict(zip(("phase", "iter", "is_feasible", "was_feasible", "recalc", "trace_index", "accepted_alpha", "ended",
                          "returned"), [int(v) for v in ints]))
            d.update(dict(zip(("xreg", "smooth", "th_stop", "cost", "cost_prev"), [float(v) for v in reals])))
            d.update(xs=xs, us=us, k=k)
            out.append(d)
        return out

    def iter_probe(self, xs, us, is_feasible, was_feasible, ddp, xreg, smooth, k=None):
        """one iteration's work from an iterate, every step length rolled out (oracle_iter_probe)"""
        na = self.prm.n_alphas
        scal_in = np.array([float(is_feasible), float(was_feasible), float(ddp), float(xreg), float(smooth)])
        scal_out = np.zeros(7)
        ok = np.zeros(na, dtype=np.int32)
        cost_try, d0, d1 = np.zeros(na), np.zeros(na), np.zeros(na)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        us = np.ascontiguousarray(us, dtype=np.float64)
        k = None if k is None else np.ascontiguousarray(k, dtype=np.float64)
        self.L.oracle_iter_probe(self.h, P(xs), P(us), P(k), P(scal_in), P(scal_out), ok.ctypes.data_as(_ip), P(cost_try), P(d0),
                                P(d1))
        return dict(cost=scal_out[0], is_feasible=bool(scal_out[1]), gapnorm=scal_out[2], xreg=scal_out[3], dg=scal_out[4],
                    dq=scal_out[5], direction_ok=bool(scal_out[6]), ok=ok, cost_try=cost_try, d0=d0, d1=d1)

    def iter_step(self, xs, us, is_feasible, was_feasible, ddp, xreg, smooth, th_stop, cost, cost_prev, it, k=None, upstream=False):
        """exactly one pass through the loop body of solveFDDP / solveDDP from the given iterate (oracle_iter_step)"""
        scal_in = np.array([float(is_feasible), float(was_feasible), float(ddp), float(xreg), float(smooth), float(th_stop),
                            float(cost), float(cost_prev), float(it), float(upstream)])
        ints = np.zeros(5, dtype=np.int32)
        out = np.zeros(10)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        us = np.ascontiguousarray(us, dtype=np.float64)
        k = None if k is None else np.ascontiguousarray(k, dtype=np.float64)
        self.L.oracle_iter_step(self.h, P(xs), P(us), P(k), P(scal_in), ints.ctypes.data_as(_ip), P(out))
        d = dict(zip(("result", "accepted_alpha", "is_feasible", "was_feasible", "recorded"), [int(v) for v in ints]))
        d.update(dict(zip(("steplength", "xreg", "cost", "cost_prev", "stop", "dV", "dVexp", "d0", "d1", "gapnorm"),
                          [float(v) for v in out])))
        return d

    def last_gains(self):
        """K, k, Vx as the last backward pass left them (no recomputation)"""
        K = np.zeros((self.T, self.nu, self.ndx))
        k = np.zeros((self.T, self.nu))
        Vx = np.zeros((self.T + 1, self.ndx))
        self.L.oracle_get_gains(self.h, P(K), P(k), P(Vx))
        return K, k, Vx

    def node_calc(self, t, x, u, diff=True):
        n, m = self.ndx, self.nu
        out = dict(xnext=np.zeros(self.nx), Fx=np.zeros((n, n)), Fu=np.zeros((n, m)), Lx=np.zeros(n), Lu=np.zeros(m),
                   Lxx=np.zeros((n, n)), Lxu=np.zeros((n, m)), Luu=np.zeros((m, m)), acc=np.zeros(self.nv),
                   u_squash=np.zeros(m), lam=np.zeros(6))
        c = C.c_double()
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self.L.oracle_node_calc(self.h, t, P(x), P(u), int(diff), P(out["xnext"]), C.cast(C.byref(c), _dp), P(out["Fx"]),
                               P(out["Fu"]), P(out["Lx"]), P(out["Lu"]), P(out["Lxx"]), P(out["Lxu"]), P(out["Luu"]),
                               P(out["acc"]), P(out["u_squash"]), P(out["lam"]))
        out["cost"] = c.value
        return out

    def integrate(self, x, dx):
        o = np.zeros(self.nx)
        self.L.oracle_state_integrate(self.h, P(np.ascontiguousarray(x)), P(np.ascontiguousarray(dx)), P(o))
        return o

    def diff(self, a, b):
        o = np.zeros(self.ndx)
        self.L.oracle_state_diff(self.h, P(np.ascontiguousarray(a)), P(np.ascontiguousarray(b)), P(o))
        return o

    def phase_calcdiff(self, xs, us, is_feasible=False, was_feasible=False):
        fs = np.zeros((self.T + 1, self.ndx))
        feas = C.c_int()
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        us = np.ascontiguousarray(us, dtype=np.float64)
        cost = self.L.oracle_phase_calcdiff(self.h, P(xs), P(us), int(is_feasible), int(was_feasible), P(fs), C.byref(feas))
        return cost, fs, bool(feas.value)

    def phase_tape(self, t):
        n, m = self.ndx, self.nu
        o = dict(Fx=np.zeros((n, n)), Fu=np.zeros((n, m)), Lx=np.zeros(n), Lu=np.zeros(m), Lxx=np.zeros((n, n)),
                 Lxu=np.zeros((n, m)), Luu=np.zeros((m, m)), xnext=np.zeros(self.nx))
        c = C.c_double()
        self.L.oracle_phase_tape(self.h, t, P(o["Fx"]), P(o["Fu"]), P(o["Lx"]), P(o["Lu"]), P(o["Lxx"]), P(o["Lxu"]), P(o["Luu"]),
                                P(o["xnext"]), C.cast(C.byref(c), _dp))
        o["cost"] = c.value
        return o

    def phase_backward(self, xreg=1e-9):
        n, m, T_ = self.ndx, self.nu, self.T
        K = np.zeros((T_, m, n))
        k = np.zeros((T_, m))
        Vx = np.zeros((T_ + 1, n))
        Vxx = np.zeros((T_ + 1, n, n))
        dgdq = np.zeros(2)
        ok = self.L.oracle_phase_backward(self.h, float(xreg), P(K), P(k), P(Vx), P(Vxx), P(dgdq))
        return bool(ok), K, k, Vx, Vxx, dgdq

    def box_qp(self, H, q, lb, ub, xinit):
        m = len(q)
        H, q, lb, ub, xinit = [np.ascontiguousarray(a, dtype=np.float64) for a in (H, q, lb, ub, xinit)]
        x, Hinv = np.zeros(m), np.zeros((m, m))
        fm = np.zeros(m, dtype=np.int32)
        ok = self.L.oracle_box_qp(self.h, m, P(H), P(q), P(lb), P(ub), P(xinit), P(x), fm.ctypes.data_as(_ip), P(Hinv))
        return bool(ok), x, fm.astype(bool), Hinv

    def phase_expected_ddp(self):
        d01 = np.zeros(2)
        self.L.oracle_phase_expected_ddp(self.h, P(d01))
        return d01

    def phase_forward(self, alpha, ddp=False):
        xs = np.zeros((self.T + 1, self.nx))
        us = np.zeros((self.T, self.nu))
        c = C.c_double()
        d01 = np.zeros(2)
        ok = self.L.oracle_phase_forward(self.h, float(alpha), int(ddp), P(xs), P(us), C.cast(C.byref(c), _dp), P(d01))
        return bool(ok), xs, us, c.value, d01


def solve_batch(desc, x0s, maxiter=100, nthreads=1, params=None, want_traj=True, variant=None):
    prm = params if params is not None else default_params()
    x0s = np.ascontiguousarray(x0s, dtype=np.float64)
    B = x0s.shape[0]
    T_, nx, nu = desc.T, desc.nx, desc.nu
    xs = np.zeros((B, T_ + 1, nx)) if want_traj else None
    us = np.zeros((B, T_, nu)) if want_traj else None
    usq = np.zeros((B, T_, nu)) if want_traj else None
    cost = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    secs = orc(variant).oracle_solve_batch(C.byref(desc), C.byref(prm), B, P(x0s), int(maxiter), int(nthreads), P(xs), P(us), P(usq),
                                    P(cost), iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip))
    return dict(xs=xs, us=us, us_squash=usq, cost=cost, iter=iters, status=status, seconds=secs)


def plant_rk4(desc, x, u, dt_s, substeps=1):
    """AerialSimulator.simulateStep restated on the CPU (oracle_plant_rk4): x, u are single vectors or batches."""
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    u = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(u), (x.shape[0], desc.nu)), dtype=np.float64)
    out = np.zeros_like(x)
    for b in range(x.shape[0]):
        cur = x[b].copy()
        nxt = np.zeros_like(cur)
        for _ in range(substeps):
            orc().oracle_plant_rk4(C.byref(desc), P(cur), P(np.ascontiguousarray(u[b])), float(dt_s), P(nxt))
            cur, nxt = nxt.copy(), cur
        out[b] = cur
    return out
