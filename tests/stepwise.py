"""Teacher-forced parity: the device solver is put at iterates recorded from the ORACLE's own solves and must reproduce
each iteration -- tape, gains, the cost of EVERY step length, the accepted step, the new regularisation, feasibility and the
stop decision.  Test infrastructure (used by tests/ only).

Why.  On the perturbed eagle_catch batch the iteration paths of any two FP64 implementations part ways (the oracle against
its own FMA build does, profiles/r02_oracle_sensitivity.json), so comparing FINAL trajectories says little about whether
every device step is right.  Comparing ONE step from identical inputs does: the differences are rounding (1e-12 relative on
costs), decisions are exact unless an inequality of src/sbfddp.cpp:271-288 / :309 is tied to that precision.

Reference semantics exercised per iterate: SolverDDP::calcDiff, backwardPass, computeGains, SolverFDDP::
updateExpectedImprovement / forwardPass / expectedImprovement and the loop body of SolverSbFDDP::solveFDDP /
solveDDP (src/sbfddp.cpp:241-311, 329-389).

Two back ends with one interface: GpuBackend (the C ABI of libempc.so: empc_solver_set_states / empc_sweep_batch /
empc_select_batch ...) and EmuBackend (tests/csrc/lane_emulator.cpp: the same kernel bodies, lane by lane on the CPU).
"""
import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# EMU_MACROS="EMPC_BWD_SYMTILES=1 ...": the emulator of a build-time VARIANT of the kernel bodies (empc_variants.hpp), in a library
# of its own -- any emulator test or tool runs on a variant this way (tools/variant_verdicts.py)
EMU_MACROS = os.environ.get("EMU_MACROS", "").split()
EMU = os.path.join(ROOT, "tests", "csrc", "liblane_emulator%s.so" % ("".join("_" + m.replace("=", "").replace("EMPC_", "").lower() for m in EMU_MACROS)))
T = ob.T
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# tolerances of the teacher-forced comparison (relative to 1 + |reference| unless stated)
TOL_TAPE = 1e-9
TOL_GAINS = 1e-6
TOL_COST = 1e-9
# quantities that pass through a trial rollout: max(TOL_COST, NOISE_FACTOR x the distance between the oracle and its own
# FMA-contracted build on that very trial)
NOISE_FACTOR = 100.0
CHAOTIC = 1e-4
PREFIX_SPLIT = 1e-6  # an accepted chaotic trial is compared knot by knot up to where the oracle's own variants part by this much
FAR_TRIAL = 100.0   # a rejected trial this many times costlier than the iterate ...
TOL_COST_FAR = 1e-6  # ... is compared at this relative tolerance
BLOWN_UP = 1e3       # positions / joint angles / rates beyond this (3 is normal) ...
COST_EXPLODED = 1e8  # ... or a cost beyond this (2 ... 1e4 is normal): the iterate of a rollout that has exploded
TIE = 1e-8


def load_emulator():
    src = os.path.join(ROOT, "tests", "csrc", "lane_emulator.cpp")
    csrc = os.path.join(ROOT, "eagle-mpc_amd", "csrc")
    hdrs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "empc_types.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(h) > os.path.getmtime(EMU) for h in hdrs + [src]):
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include")] + ["-D" + m for m in EMU_MACROS] + [src,
                               "-o", EMU])
    L = C.CDLL(EMU)
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams), C.c_int]
    L.emu_destroy.argtypes = [C.c_void_p]
    L.emu_rec.argtypes = [C.c_void_p]
    L.emu_set_x0.argtypes = [C.c_void_p, _dp]
    L.emu_set_warmstart.argtypes = [C.c_void_p, _dp, _dp]
    L.emu_solve_c.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.emu_get.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip]
    L.emu_get_states.argtypes = [C.c_void_p, C.POINTER(T.TrajState)]
    L.emu_set_states.argtypes = [C.c_void_p, C.POINTER(T.TrajState)]
    L.emu_sweep.argtypes = [C.c_void_p, C.c_int]
    L.emu_set_trials.argtypes = [C.c_void_p, _ip, _dp, _dp]
    L.emu_get_trials.argtypes = [C.c_void_p, _dp, _dp, _ip]
    L.emu_get_tape.argtypes = [C.c_void_p, _dp]
    L.emu_get_gains.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.emu_set_gains.argtypes = [C.c_void_p, _dp, _dp]
    L.emu_stream_row_doubles.argtypes = [C.c_void_p]
    L.emu_stream_c.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.POINTER(C.c_longlong)]
    return L


class EmuBackend:
    """the kernel bodies on the CPU lane emulator"""

    def __init__(self, emu, desc, prm, batch):
        self.L, self.batch = emu, batch
        self.h = C.c_void_p(emu.emu_create(C.byref(desc), C.byref(prm), batch))
        assert self.h.value
        self.T, self.nx, self.ndx, self.nu, self.na = desc.T, desc.nx, desc.ndx, desc.nu, prm.n_alphas
        self.rec = emu.emu_rec(self.h)

    def __del__(self):
        try:
            self.L.emu_destroy(self.h)
        except Exception:
            pass

    def set_x0s(self, x0s):
        self.L.emu_set_x0(self.h, ob.P(np.ascontiguousarray(x0s, dtype=np.float64).reshape(self.batch, self.nx)))

    def set_candidates(self, xs, us):
        xs = None if xs is None else np.ascontiguousarray(xs, dtype=np.float64).reshape(self.batch, self.T + 1, self.nx)
        us = None if us is None else np.ascontiguousarray(us, dtype=np.float64).reshape(self.batch, self.T, self.nu)
        self.L.emu_set_warmstart(self.h, ob.P(xs), ob.P(us))

    def set_states(self, st):
        self.L.emu_set_states(self.h, st)

    def get_states(self):
        st = (T.TrajState * self.batch)()
        self.L.emu_get_states(self.h, st)
        return st

    def sweep(self, stages=T.STAGE_ALL):
        self.L.emu_sweep(self.h, int(stages))

    def select(self, try_ok=None, try_cost=None, try_dv=None):
        ok = None if try_ok is None else np.ascontiguousarray(try_ok, dtype=np.int32)
        c = None if try_cost is None else np.ascontiguousarray(try_cost, dtype=np.float64)
        dv = None if try_dv is None else np.ascontiguousarray(try_dv, dtype=np.float64)
        self.L.emu_set_trials(self.h, None if ok is None else ok.ctypes.data_as(_ip), ob.P(c), ob.P(dv))
        self.L.emu_sweep(self.h, T.STAGE_SELECT)

    def trials(self):
        c, dv = np.zeros((self.batch, self.na)), np.zeros((self.batch, self.na))
        ok = np.zeros((self.batch, self.na), dtype=np.int32)
        self.L.emu_get_trials(self.h, ob.P(c), ob.P(dv), ok.ctypes.data_as(_ip))
        return c, dv, ok

    def tape(self):
        t = np.zeros((self.batch, self.T + 1, self.rec))
        self.L.emu_get_tape(self.h, ob.P(t))
        return t

    def gains(self):
        K = np.zeros((self.batch, self.T, self.nu, self.ndx))
        k = np.zeros((self.batch, self.T, self.nu))
        Vx = np.zeros((self.batch, self.T + 1, self.ndx))
        self.L.emu_get_gains(self.h, ob.P(K), ob.P(k), ob.P(Vx))
        return K, k, Vx

    def set_gains(self, K=None, k=None):
        K = None if K is None else np.ascontiguousarray(K, dtype=np.float64)
        k = None if k is None else np.ascontiguousarray(k, dtype=np.float64)
        self.L.emu_set_gains(self.h, ob.P(K), ob.P(k))

    def candidates(self):
        xs = np.zeros((self.batch, self.T + 1, self.nx))
        us = np.zeros((self.batch, self.T, self.nu))
        self.L.emu_get(self.h, ob.P(xs), ob.P(us), None, None, None, None)
        return xs, us


class GpuBackend:
    """the C ABI of libempc.so (the product path)"""

    def __init__(self, empc, problem, prm, batch, solver_cls=None):
        cls = solver_cls or empc.SolverSbFDDP
        self.s = cls(problem, batch=batch, params=prm)
        self.batch = batch
        self.T, self.nx, self.ndx, self.nu, self.na, self.rec = self.s.T, self.s.nx, self.s.ndx, self.s.nu, prm.n_alphas, self.s.rec
        self._has_k = cls is not empc.SolverSbFDDP

    def set_x0s(self, x0s):
        self.s.set_x0s(x0s)

    def set_candidates(self, xs, us):
        self.s.set_candidates(xs, us)

    def set_states(self, st):
        self.s.set_states(st)

    def get_states(self):
        return self.s.get_states()

    def sweep(self, stages=T.STAGE_ALL):
        self.s.sweep(stages)

    def select(self, try_ok=None, try_cost=None, try_dv=None):
        self.s.select(try_ok, try_cost, try_dv)

    def trials(self):
        return self.s.trials()

    def tape(self):
        return self.s.tape()

    def gains(self):
        return self.s.gains()

    def set_gains(self, K=None, k=None):
        self.s.set_gains(K, k)

    def candidates(self):
        return self.s.xs_batch, self.s.us_batch


# ---------------------------------------------------------------------------------------------------------------------
def fresh_state(prm, maxiter):
    """TrajState at the start of solve([], [], maxiter, false) on a fresh SolverSbFDDP (src/sbfddp.cpp:198-210)"""
    st = T.TrajState()
    st.maxiter = maxiter
    st.smooth = st.smooth_next = prm.smooth_init
    st.convergence = st.th_stop = prm.convergence_init
    st.xreg = st.ureg = prm.reg_init
    st.steplength = 1.0
    st.need_calc = st.need_lin = 1
    st.accepted_alpha = st.last_alpha = -1
    st.job = -1
    if prm.solver_type != T.SOLVER_SBFDDP:
        st.phase = T.PHASE_DDP if prm.solver_type == T.SOLVER_BOXDDP else 0
        st.th_stop = prm.box_th_stop
    return st


def state_at_iterate(prm, it, maxiter):
    """TrajState that puts a trajectory at the top of oracle iteration `it` (oracle_binding.OracleSolver.iterates())"""
    st = fresh_state(prm, maxiter)
    st.phase = it["phase"]
    st.iter = it["iter"]
    st.is_feasible = it["is_feasible"]
    st.was_feasible = it["was_feasible"]
    st.smooth = st.smooth_next = it["smooth"]
    st.convergence = st.th_stop = it["th_stop"]
    st.xreg = st.ureg = it["xreg"]
    st.cost = it["cost"]
    st.cost_prev = it["cost_prev"]
    if it["phase"] == T.PHASE_DDP and prm.solver_type == T.SOLVER_SBFDDP:
        st.status = T.STATUS_DDP_CLEANUP
    return st


def tape_blocks(rec, n, m):
    """record layout of empc_dev_model.hpp (Dims): A = [Fx Fu], HX = [Lxx Lxu], LUU, LX, LU, GAP, COST"""
    nm = n + m
    A = rec[:n * nm].reshape(n, nm)
    HX = rec[n * nm:2 * n * nm].reshape(n, nm)
    o = 2 * n * nm
    return {"Fx": A[:, :n], "Fu": A[:, n:], "Lxx": HX[:, :n], "Lxu": HX[:, n:], "Luu": rec[o:o + m * m].reshape(m, m),
            "Lx": rec[o + m * m:o + m * m + n], "Lu": rec[o + m * m + n:o + m * m + n + m],
            "gap": rec[o + m * m + n + m:o + m * m + 2 * n + m], "cost": rec[o + m * m + 2 * n + m:o + m * m + 2 * n + m + 1]}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max())) if a.size else 0.0


def oracle_paths(desc, prm, x0s, maxiter=100, warm=None):
    """solve every rollout with the oracle, recording iterates and the iteration trace"""
    out = []
    for b in range(len(x0s)):
        o = ob.OracleSolver(desc, prm)
        o.set_x0(x0s[b])
        o.record_iterates(True)
        if warm is None:
            o.solve(None, None, maxiter)
        else:
            o.solve(warm[0][b], warm[1][b], maxiter, warm[2])
        out.append(dict(iterates=o.iterates(), trace=o.trace(), result=o.result()))
    return out


_THIRD = {}


def _third(desc, prm, smooth):
    """the NumPy restatement's view of a problem at one smoothing value, kept for the last (problem, smoothing) asked for; the cache
    holds the descriptor itself, so that a later descriptor at the same address is never mistaken for it"""
    sys.path.insert(0, os.path.join(ROOT, "oracle")) if os.path.join(ROOT, "oracle") not in sys.path else None
    import numpy_restatement as nr
    if _THIRD.get("desc") is not desc or _THIRD.get("smooth") != smooth or _THIRD.get("prm") is not prm:
        _THIRD.clear()
        _THIRD.update(desc=desc, prm=prm, smooth=smooth, P=nr.Problem(desc, prm), sets=nr.cost_sets_of(desc, prm, smooth))
    return nr, _THIRD["P"], _THIRD["sets"]


def third_algorithm_distance(desc, prm, it, t, key, ref_block):
    """distance of the NumPy second restatement (oracle/numpy_restatement.py) from the oracle on block `key` of knot t of iterate
    `it` -- the arbiter of tape entries on ill-conditioned nodes (see its call site)"""
    smooth = float(it["smooth"])
    nr, P, sets = _third(desc, prm, smooth)
    node = nr.node(P, sets[desc.knot_set[t]], np.asarray(it["xs"][t]), None if t == desc.T else np.asarray(it["us"][t]), smooth)
    return rel(np.ravel(np.asarray(node[key], dtype=float)), np.ravel(ref_block))


def constraint_condition(desc, prm, it, t):
    """condition number of the constraint matrix Jc M^-1 Jc^T of knot t of iterate `it` (1 for a knot without contacts): what the
    Schur-complement form of the contact dynamics -- the reference's, the oracle's and the kernels' -- amplifies rounding by"""
    smooth = float(it["smooth"])
    nr, P, sets = _third(desc, prm, smooth)
    contacts = sets[desc.knot_set[t]]["contacts"]
    if not contacts or not desc.has_contact:
        return 1.0
    md = P.md
    x = nr._c(np.asarray(it["xs"][t]))
    q = x[:P.nq]
    R, p, _, vel, acc0 = nr.kinematics(md, q, x[P.nq:], np.zeros(P.nv), gravity=False)
    unit = []
    for j in range(P.nv):
        e = np.zeros(P.nv)
        e[j] = 1.0
        unit.append(nr.kinematics(md, q, e, np.zeros(P.nv), gravity=False)[3])
    rows = []
    for ct in contacts:
        nc = 3 if ct["type"] == 0 else 6
        Xf = nr.frame_kin(md, ct["frame"], R, p, vel, acc0)[4]
        rows.append(np.array([np.real((Xf @ unit[j][md.frame_body[ct["frame"]]])[:nc]) for j in range(P.nv)]).T)
    J = np.vstack(rows)
    M = np.real(nr.mass_matrix(md, q))
    return float(np.linalg.cond(J @ np.linalg.solve(M, J.T)))


# what the device's arithmetic (its own sin / cos, reciprocal pivots, reciprocal square roots: a few units in the last place per
# operation where the oracle and NumPy have half a unit) leaves in a contact node's derivatives PER UNIT of cond(Jc M^-1 Jc^T):
# measured 3e-13 ... 3e-12 on the lane emulator (one contact: 1.6e-10 in the acceleration at cond ~ 50; two contacts: 1e-8 at 4e4)
COND_UNIT = 1e-12


def third_algorithm_trial_cost(desc, prm, it, x0, K, k, alpha, ddp, feasible):
    """cost of the trial rollout of step length alpha from iterate `it` with the ORACLE's gains, by the NumPy second restatement:
    SolverFDDP::forwardPass / SolverSbFDDP::forwardPassDDP as oracle/solver.hpp forward_pass states them (gap-aware unless the
    iterate is feasible or alpha = 1; the clean-up pass ignores gaps), on dynamics and costs evaluated by a third algorithm --
    the arbiter of trial costs on rollouts through ill-conditioned nodes.  None when the rollout does not survive."""
    smooth = float(it["smooth"])
    nr, P, sets = _third(desc, prm, smooth)
    nq, Tn = P.nq, desc.T
    xs, us = np.asarray(it["xs"]), np.asarray(it["us"])
    box = prm.solver_type != T.SOLVER_SBFDDP
    lb = np.array([desc.u_lb[i] for i in range(desc.nu)])
    ub = np.array([desc.u_ub[i] for i in range(desc.nu)])

    def step(t, x, u):
        r = nr.node_value(P, sets[desc.knot_set[t]], x, u, smooth, t == Tn)
        return np.real(r[0]), float(np.real(r[1]))
    plain = ddp or feasible or alpha == 1.0
    fs = None
    if not plain:  # gaps of the iterate: fs[0] = x0 (-) xs[0], fs[t+1] = f(xs[t], us[t]) (-) xs[t+1]
        fs = [np.real(nr.state_diff(nq, xs[0], x0))]
        for t in range(Tn):
            fs.append(np.real(nr.state_diff(nq, xs[t + 1], step(t, xs[t], us[t])[0])))
    cost, xnext = 0.0, np.asarray(x0, dtype=float)
    for t in range(Tn):
        xt = xnext if plain else np.real(nr.state_integrate(nq, xnext, fs[t] * (alpha - 1.0)))
        dx = np.real(nr.state_diff(nq, xs[t], xt))
        ut = us[t] - alpha * k[t] - K[t] @ dx
        if box:
            ut = np.minimum(np.maximum(ut, lb), ub)
        xnext, c = step(t, xt, ut)
        cost += c
        if not (np.isfinite(cost) and np.isfinite(xnext).all()) or abs(cost) > 1e30 or np.abs(xnext).max() > 1e30:
            return None
    xT = xnext if plain else np.real(nr.state_integrate(nq, xnext, fs[Tn] * (alpha - 1.0)))
    cost += step(Tn, xT, None)[1]
    return cost if np.isfinite(cost) else None


def teacher_forced(backend_factory, desc, prm, x0s, paths, maxiter=100, chunk=1024, tape_every=37, report=None, workers=None,
                   tol_tape=TOL_TAPE):
    """Every (rollout, iterate) pair of `paths` as one trajectory of a device batch: one iteration each, compared with the
    oracle.  backend_factory(batch) -> backend.  Returns the report dict; raises AssertionError on any mismatch."""
    pairs = [(b, i) for b in range(len(paths)) for i in range(len(paths[b]["iterates"]))]
    workers = workers or min(os.cpu_count() or 1, 32)
    box = prm.solver_type != T.SOLVER_SBFDDP
    n, m, na = desc.ndx, desc.nu, prm.n_alphas
    rep = dict(pairs=len(pairs), rollouts=len(paths), max_rel={}, decisions_checked=0, tapes_checked=0, trial_costs_checked=0,
               direction_failures=0, min_margin=None, margins=[])
    mx = rep["max_rel"]

    def upd(key, v):
        mx[key] = max(mx.get(key, 0.0), float(v))

    probe = {}
    for c0 in range(0, len(pairs), chunk):
        sub = pairs[c0:c0 + chunk]
        B = len(sub)
        be = backend_factory(B)
        its = [paths[b]["iterates"][i] for b, i in sub]
        be.set_x0s(np.array([x0s[b] for b, _ in sub]))
        be.set_candidates(np.array([it["xs"] for it in its]), np.array([it["us"] for it in its]))
        if box:
            be.set_gains(None, np.array([it["k"] for it in its]))
        st = (T.TrajState * B)()
        for j, it in enumerate(its):
            st[j] = state_at_iterate(prm, it, maxiter)
        be.set_states(st)
        be.sweep(T.STAGE_LINEARIZE | T.STAGE_BACKWARD | T.STAGE_ROLLOUT)
        mid = be.get_states()
        mid = [T.TrajState.from_buffer_copy(mid[j]) for j in range(B)]
        cost_try, dv, ok = be.trials()
        want_tape = [j for j in range(B) if (c0 + j) % tape_every == 0]
        tape = be.tape() if want_tape else None
        K, k, Vx = be.gains() if want_tape else (None, None, None)
        be.sweep(T.STAGE_SELECT)
        fin = be.get_states()
        xs_new, us_new = be.candidates()
        dev_gains = [None]  # (K, k, Vx) of the whole chunk, fetched only if the trial-cost arbiter needs them

        def oracle_side(j):
            # the oracle's view of pair j: the iteration itself (all step lengths), the same by its FMA-contracted build and
            # from the iterate moved by one unit in the last place (two draws) -- how far correct FP64 evaluations of this
            # very iteration lie apart, the yardstick for quantities a near-unstable trial rollout amplifies
            (b, i), it = sub[j], its[j]
            ddp = it["phase"] == T.PHASE_DDP
            kk = it["k"] if box else None
            o = ob.OracleSolver(desc, prm)
            o.set_x0(x0s[b])
            p = o.iter_probe(it["xs"], it["us"], it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"], k=kk)
            of = ob.OracleSolver(desc, prm, variant="fma")
            of.set_x0(x0s[b])
            pv = [of.iter_probe(it["xs"], it["us"], it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"], k=kk)]
            moved = []
            for draw in range(2):
                rng = np.random.default_rng(1000003 * b + 101 * i + draw)
                xs_u = np.nextafter(it["xs"], np.where(rng.random(it["xs"].shape) < 0.5, -np.inf, np.inf))
                us_u = np.nextafter(it["us"], np.where(rng.random(it["us"].shape) < 0.5, -np.inf, np.inf))
                xs_u[:, 3:7] /= np.linalg.norm(xs_u[:, 3:7], axis=1, keepdims=True)
                moved.append((xs_u, us_u))
                ou = ob.OracleSolver(desc, prm)
                ou.set_x0(x0s[b])
                pv.append(ou.iter_probe(xs_u, us_u, it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"], k=kk))
            # the whole iteration (decision included) by both builds: yardstick for the scalars select leaves and for the
            # accepted candidate
            step_noise = None
            if p["direction_ok"]:
                res = []
                for variant in (None, "fma", "ulp0", "ulp1"):
                    os_ = ob.OracleSolver(desc, prm, variant="fma" if variant == "fma" else None)
                    os_.set_x0(x0s[b])
                    xs_i, us_i = it["xs"], it["us"]
                    if variant in ("ulp0", "ulp1"):
                        xs_i, us_i = moved[int(variant[3])]
                    r_ = os_.iter_step(xs_i, us_i, it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"],
                                       it["th_stop"], it["cost"], it["cost_prev"], it["iter"], k=kk,
                                       upstream=(prm.solver_type == T.SOLVER_BOXFDDP))
                    res.append((r_, os_.result()))
                ra, xa = res[0]
                if all(rb["accepted_alpha"] == ra["accepted_alpha"] for rb, _ in res[1:]):
                    step_noise = {key: max(abs(ra[key] - rb[key]) for rb, _ in res[1:]) for key in ("stop", "dV", "dVexp", "cost")}
                    step_noise["xs"] = max(float(np.abs(xa["xs"] - xb["xs"]).max() / (1.0 + np.abs(xa["xs"]).max())) for _, xb in res[1:])
                    step_noise["us"] = max(float(np.abs(xa["us"] - xb["us"]).max() / (1.0 + np.abs(xa["us"]).max())) for _, xb in res[1:])
                    # knot by knot: how far the oracle's own variants lie apart on the accepted candidate (a rollout that is
                    # about to blow up agrees to rounding over its first knots and parts ways exponentially after)
                    step_noise["xs_knot"] = np.max([np.abs(xa["xs"] - xb["xs"]).max(axis=1) for _, xb in res[1:]], axis=0)
                    step_noise["us_knot"] = np.max([np.abs(xa["us"] - xb["us"]).max(axis=1) for _, xb in res[1:]], axis=0)
            ref_tape = ref_gains = None
            if j in want_tape and p["direction_ok"]:
                ref_tape = [o.phase_tape(t) for t in range(desc.T + 1)]
                noise_tape = [of.phase_tape(t) for t in range(desc.T + 1)]
                for t in range(desc.T + 1):
                    ref_tape[t]["noise"] = {key: rel(np.ravel(noise_tape[t][key]), np.ravel(ref_tape[t][key])) for key in
                                            ("Fx", "Fu", "Lxx", "Lxu", "Luu", "Lx", "Lu", "cost")}
                ref_gains = o.last_gains() + (of.last_gains(),)
            p["gains"] = o.last_gains()[:2] if p["direction_ok"] else None  # (K, k) of this iterate, for the trial-cost arbiter
            return p, pv, ref_tape, ref_gains, step_noise

        with ThreadPoolExecutor(max_workers=workers) as pool:
            oracle = list(pool.map(oracle_side, range(B)))
        for j, ((b, i), it) in enumerate(zip(sub, its)):
            ddp = it["phase"] == T.PHASE_DDP
            p, pv, ref_tape, ref_gains, sn = oracle[j]
            sn = sn or dict(stop=np.inf, dV=np.inf, dVexp=np.inf, cost=np.inf, xs=np.inf, us=np.inf, xs_knot=None, us_knot=None)  # the builds disagree on the step: no value bound
            pf = pv[0]
            pu = pv[1:]
            g = mid[j]
            amp = 1.0
            where = "rollout %d iterate %d (pass %d iter %d)" % (b, i, it["phase"], it["iter"])
            # ---- after calcDiff + computeDirection -------------------------------------------------------------------
            # (an iterate that has exploded is set aside BEFORE the outcome of computeDirection is compared: the tapes of the two
            #  sides still agree to 1e-14, but the Riccati recursion inverts Quu of 1e17-sized terms at a condition number that
            #  turns the last bit into the leading digit, and so the number of regularisation retries may differ -- found by the
            #  round-5 emulator soak, seed 53: eagle_catch iterate at cost 4e17 on which the oracle's LLT fails at every
            #  regularisation (pivot -1.7e8 among entries of 2e11) and the device's passes at 1e3)
            if (float(np.abs(it["xs"][:, 7:]).max()) > BLOWN_UP or abs(p["cost"]) > COST_EXPLODED or float(np.abs(it["xs"][:, :3]).max()) > BLOWN_UP):
                # joint angles / rates beyond 1e3 (3 is normal): a rollout that has already exploded (cost 1e13) and iterates on at
                # that level until the iteration limit.  Every cost term cancels at 1e13: nothing is comparable at rounding level
                # (the scalar functions themselves stay accurate).  Counted (callers cap the count), finite outputs required; the
                # outcome of computeDirection is still recorded, and a disagreement of the two sides is counted, not asserted
                rep["blown_up_iterates"] = rep.get("blown_up_iterates", 0) + 1
                rep["iterates_skipped_exploded"] = rep.get("iterates_skipped_exploded", 0) + 1
                assert np.isfinite(g.cost) or not np.isfinite(p["cost"]), where
                if not p["direction_ok"]:
                    rep["direction_failures"] += 1
                if bool(g.bwd_failed) != (not p["direction_ok"]):
                    rep["exploded_direction_disagreements"] = rep.get("exploded_direction_disagreements", 0) + 1
                rep["decisions_checked"] += 1
                continue
            if bool(g.bwd_failed) != (not p["direction_ok"]):
                # computeDirection gives up (regularisation at its maximum) on one side only: legitimate when the oracle's own
                # builds disagree on the retries for this iterate (a Quu pivot tied to rounding precision)
                assert any(q["direction_ok"] != p["direction_ok"] or q["xreg"] != p["xreg"] for q in pv), \
                    (where, g.bwd_failed, g.xreg, p["direction_ok"], p["xreg"], [(q["direction_ok"], q["xreg"]) for q in pv])
                rep["direction_ties_excused"] = rep.get("direction_ties_excused", 0) + 1
                rep["decisions_checked"] += 1
                continue
            if not p["direction_ok"]:
                rep["direction_failures"] += 1
            scale = 1.0 + abs(p["cost"])
            # an iterate that has blown up (joint rates of 1e3 ... 1e6 where 3 is normal, costs of 1e13): the rounding error of every
            # term grows with the magnitude of the state; the base tolerances scale with it (1 for any sane iterate)
            xmax = float(np.abs(it["xs"][:, 7:]).max())
            blow = max(1.0, xmax / 10.0)
            if blow > 1.0:
                rep["blown_up_iterates"] = rep.get("blown_up_iterates", 0) + 1
            upd("cost", abs(g.cost - p["cost"]) / scale / blow)
            # (the yardstick of every other quantity, for the iterate's own cost too: what the oracle's FMA build and its evaluations
            #  of the iterate moved by one unit in the last place differ by -- on a node next to a rank-deficient pair of constraints,
            #  cond 2e6, that is 1e-9 of the cost: round-6 soak of the two-contact class, seed 93)
            ncost = max(abs(q["cost"] - p["cost"]) for q in pv)
            if not abs(g.cost - p["cost"]) <= max(TOL_COST * blow * scale, NOISE_FACTOR * ncost) and desc.has_contact:
                # (a contact node next to a rank-deficient pair of constraints in this iterate: the contact forces, and with them
                #  the friction-cone cost, carry cond(Jc M^-1 Jc^T) x the arithmetic's error floor -- see COND_UNIT)
                cmax = max(constraint_condition(desc, prm, it, t) for t in range(desc.T + 1))
                rep.setdefault("costs_arbitrated", []).append((b, i, float(abs(g.cost - p["cost"]) / scale), float(cmax)))
                assert abs(g.cost - p["cost"]) <= COND_UNIT * cmax * scale, (where, g.cost, p["cost"], ncost, "max cond(Jc M^-1 Jc^T)", cmax)
                rep["costs_excused_ill_conditioned"] = rep.get("costs_excused_ill_conditioned", 0) + 1
            else:
                assert abs(g.cost - p["cost"]) <= max(TOL_COST * blow * scale, NOISE_FACTOR * ncost), (where, g.cost, p["cost"], ncost)
            assert bool(g.is_feasible) == p["is_feasible"], where
            upd("gapnorm", abs(g.gapnorm - p["gapnorm"]) / (1.0 + abs(p["gapnorm"])))
            ngap = max(abs(q["gapnorm"] - p["gapnorm"]) for q in pv)
            # (gaps are differences of states: on an iterate that has blown up -- joint rates of 4e3 rad/s, controls of 4e7 --
            #  their absolute error scales with the magnitude of the states)
            xmax = float(np.abs(it["xs"]).max())
            if not abs(g.gapnorm - p["gapnorm"]) <= max(1e-9 * (1.0 + abs(p["gapnorm"]) + xmax), NOISE_FACTOR * ngap) and desc.has_contact:
                # (gaps are next states minus states: the next state of a contact node next to a rank-deficient pair of constraints
                #  carries cond(Jc M^-1 Jc^T) x the arithmetic's error floor -- see COND_UNIT; two-contact soak, seed 97: 2e-8 at 3e8)
                cmax = max(constraint_condition(desc, prm, it, t) for t in range(desc.T + 1))
                rep.setdefault("gapnorms_arbitrated", []).append((b, i, float(abs(g.gapnorm - p["gapnorm"])), float(cmax)))
                assert abs(g.gapnorm - p["gapnorm"]) <= COND_UNIT * cmax * (1.0 + abs(p["gapnorm"]) + xmax), \
                    (where, g.gapnorm, p["gapnorm"], ngap, "max cond(Jc M^-1 Jc^T)", cmax)
                rep["gapnorms_excused_ill_conditioned"] = rep.get("gapnorms_excused_ill_conditioned", 0) + 1
            else:
                assert abs(g.gapnorm - p["gapnorm"]) <= max(1e-9 * (1.0 + abs(p["gapnorm"]) + xmax), NOISE_FACTOR * ngap), (where, g.gapnorm, p["gapnorm"], ngap)
            if p["direction_ok"] and g.xreg != p["xreg"] and any(q["xreg"] != p["xreg"] or not q["direction_ok"] for q in pv):
                # the number of regularisation retries (LLT of Quu failing or not) differs between the oracle's OWN builds
                # on this iterate: a pivot tied to rounding precision; everything downstream depends on it
                rep["direction_ties_excused"] = rep.get("direction_ties_excused", 0) + 1
                rep["decisions_checked"] += 1
                continue
            if p["direction_ok"]:
                assert g.xreg == p["xreg"], (where, g.xreg, p["xreg"])
                feas = bool(g.is_feasible)
                dg = g.dg_u + (0.0 if (feas or ddp) else g.dg_f)
                dq = g.dq_u + (0.0 if (feas or ddp) else g.dq_f)
                sc = 1.0 + abs(p["dg"]) + abs(p["dq"])
                ndg = max([max(abs(q["dg"] - p["dg"]), abs(q["dq"] - p["dq"])) for q in pv if q["direction_ok"]] or [0.0]) / sc
                edg = max(abs(dg - p["dg"]), abs(dq - p["dq"])) / sc
                upd("dg_dq", edg)
                upd("dg_dq_over_tol", edg / max(1e-7, NOISE_FACTOR * ndg))
                assert edg <= max(1e-7, NOISE_FACTOR * ndg), (where, dg, p["dg"], dq, p["dq"], ndg)
                # ---- every step length --------------------------------------------------------------------------------
                mism = (ok[j] != 0) != (p["ok"] != 0)
                if mism.any():
                    # a trial that one side reports as failed ("forward_error": NaN or a number beyond 1e30) and the other
                    # does not: legitimate only for a rollout in the middle of overflowing -- the oracle's own builds disagree on
                    # its flag, or its cost is already beyond 1e15
                    var_dis = np.zeros(na, dtype=bool)
                    for q in pv:
                        var_dis |= (q["ok"] != 0) != (p["ok"] != 0)
                    assert (var_dis | ~(np.abs(p["cost_try"]) < 1e15))[mism].all(), (where, ok[j], p["ok"], p["cost_try"])
                    rep["trial_flags_excused_overflowing"] = rep.get("trial_flags_excused_overflowing", 0) + int(mism.sum())
                good = (p["ok"] != 0) & (ok[j] != 0)
                if good.any():
                    # The trial the line search accepts is held to TOL_COST.  The others only enter an accept / reject
                    # inequality (compared exactly through the accepted step below); their values are compared at
                    # TOL_COST_REJECTED: a rejected step is often a rollout on the verge of blowing up (cost 1e3 ... 1e43
                    # next to an accepted cost of 30), which amplifies the last-bit differences of the two libm's by 1e3 ... 1e6.
                    e = np.abs(cost_try[j] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"]))
                    noise = np.abs(pf["cost_try"] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"]))
                    for q in pu:
                        noise = np.maximum(noise, np.abs(q["cost_try"] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"])))
                    noise[~((pf["ok"] != 0) & good)] = np.inf  # a trial only one build survives: no bound from this side
                    # a trial on which the oracle's two builds differ by more than CHAOTIC is a rollout that blows up (costs of
                    # 1e7 ... 1e43): no value of it is comparable, only the fact that it is rejected
                    chaotic = noise > CHAOTIC
                    rep["chaotic_trials"] = rep.get("chaotic_trials", 0) + int((chaotic & good).sum())
                    noise[chaotic] = np.inf
                    tol = np.maximum(TOL_COST * blow, NOISE_FACTOR * noise)
                    acc = it["accepted_alpha"]
                    # A trial that costs FAR_TRIAL x the iterate (2e10, 3e6 ... next to 1e3) is a rollout that leaves the region
                    # of the iterate in its last knots: two implementations that agree to 3e-13 until then (RNEA + CRBA here,
                    # ABA there: the oracle's own rounding variants sit at 1e-16 and understate this) end 1e-8 apart.  Such a
                    # trial is rejected on both sides whatever its last digits -- the decision is asserted exactly below --
                    # so its value is held to TOL_COST_FAR only.
                    far = np.abs(p["cost_try"]) > FAR_TRIAL * (1.0 + abs(p["cost"]))
                    if acc >= 0:
                        far[acc] = False
                    rep["far_trials"] = rep.get("far_trials", 0) + int((far & good).sum())
                    tol = np.where(far, np.maximum(tol, TOL_COST_FAR), tol)
                    # the trial whose numbers the scalars keep: the accepted one, or the last one tried when none is accepted
                    kept = acc if acc >= 0 else na - 1
                    amp = max(1.0, tol[kept] / TOL_COST) if good[kept] else np.inf  # amplification of rounding by that rollout
                    if acc >= 0:
                        if not np.isfinite(amp):
                            rep["accepted_trials_chaotic"] = rep.get("accepted_trials_chaotic", 0) + 1
                        upd("cost_try_accepted", e[acc])
                        upd("cost_try_accepted_over_tol", e[acc] / tol[acc])
                        assert e[acc] <= tol[acc], (where, acc, e, noise, cost_try[j], p["cost_try"])
                    upd("cost_try_any", e[good].max())
                    upd("cost_try_any_over_tol", (e[good] / tol[good]).max())
                    beyond = good & (e > TOL_COST * blow)
                    rep["trial_costs_beyond_1e-9"] = rep.get("trial_costs_beyond_1e-9", 0) + int(beyond.sum())
                    for a_ in np.nonzero(beyond)[0]:
                        rep.setdefault("beyond", []).append((b, i, int(a_), float(e[a_]), float(noise[a_])))
                    over = good & ~(e <= tol)
                    if acc >= 0:
                        over[acc] = False  # (the accepted trial was asserted above, on its own terms)
                    if over.any() and prm.solver_type == T.SOLVER_SBFDDP and p.get("gains") is not None:
                        # Rejected trials beyond the tolerance AND beyond what the oracle's own builds differ by: arbitration by the
                        # third algorithm, as for the tape -- a rollout through nodes next to a rank-deficient pair of constraints
                        # (cond 1e6 ... 1e10) carries the difference of two correct algorithms, which the same algorithm in another
                        # rounding understates by orders of magnitude.  Excused only if the NumPy restatement's rollout (the oracle's
                        # gains, its own dynamics) sits as far from the oracle's cost as the device's does (within 10 x).
                        for a_ in np.nonzero(over)[0]:
                            try:
                                c3 = third_algorithm_trial_cost(desc, prm, it, x0s[b], p["gains"][0], p["gains"][1], float(2.0 ** -int(a_)),
                                                                ddp, bool(p["is_feasible"]))
                            except Exception as ex:
                                assert False, (where, int(a_), e, noise, "third algorithm unavailable: %s" % str(ex)[:200])
                            e3 = np.inf if c3 is None else abs(c3 - p["cost_try"][a_]) / (1.0 + abs(p["cost_try"][a_]))
                            # ... or the difference comes from the GAINS: the device rolls out with the gains of ITS backward pass over
                            # ITS tape, and next to such nodes those differ from the oracle's in the sixth digit (controls of the grasp
                            # knots 6e-6 apart with states equal to 3e-10: seed 97 of the two-contact soak).  Then the question for the
                            # rollout kernel is whether it is right GIVEN its gains: the third algorithm rolled out with the device's
                            # gains must land on the device's cost as closely as it lands on the oracle's with the oracle's gains.
                            e_dev = np.inf
                            if not e[a_] <= 10.0 * e3:
                                if dev_gains[0] is None:
                                    dev_gains[0] = be.gains()
                                try:
                                    c3d = third_algorithm_trial_cost(desc, prm, it, x0s[b], dev_gains[0][0][j], dev_gains[0][1][j],
                                                                     float(2.0 ** -int(a_)), ddp, bool(p["is_feasible"]))
                                except Exception as ex:
                                    assert False, (where, int(a_), e, noise, "third algorithm unavailable: %s" % str(ex)[:200])
                                e_dev = np.inf if c3d is None else abs(c3d - cost_try[j][a_]) / (1.0 + abs(cost_try[j][a_]))
                            rep.setdefault("trial_costs_arbitrated", []).append((b, i, int(a_), float(e[a_]), float(e3), float(e_dev)))
                            assert e[a_] <= 10.0 * e3 or e_dev <= 10.0 * max(e3, TOL_COST * blow), \
                                (where, int(a_), e, noise, cost_try[j], p["cost_try"], "NumPy restatement: oracle's gains", c3, "device's gains", e_dev)
                            rep["trial_costs_excused_ill_conditioned"] = rep.get("trial_costs_excused_ill_conditioned", 0) + 1
                            tol[a_] = np.inf
                    assert (e[good] <= tol[good]).all(), (where, e, noise, cost_try[j], p["cost_try"])
                    rep["trial_costs_checked"] += int(good.sum())
                    if not ddp:
                        d0 = dg + (0.0 if feas else 1.0) * dv[j]
                        d1 = dq - 2.0 * (0.0 if feas else 1.0) * dv[j]
                        sc2 = 1.0 + np.abs(p["d0"][good]).max() + np.abs(p["d1"][good]).max()
                        n2 = np.maximum(np.abs(pf["d0"] - p["d0"]), np.abs(pf["d1"] - p["d1"])) / sc2
                        for q in pu:
                            n2 = np.maximum(n2, np.maximum(np.abs(q["d0"] - p["d0"]), np.abs(q["d1"] - p["d1"])) / sc2)
                        n2[~((pf["ok"] != 0) & good) | chaotic] = np.inf
                        e2 = np.maximum(np.abs(d0 - p["d0"]), np.abs(d1 - p["d1"])) / sc2
                        upd("d0_d1", e2[good].max())
                        assert (e2[good] <= np.maximum(1e-7, NOISE_FACTOR * n2[good])).all(), (where, d0, p["d0"], d1, p["d1"])
                # ---- tape and gains of sampled iterates -----------------------------------------------------------------
                if j in want_tape:
                    excused_here = 0
                    for t in range(desc.T + 1):
                        ref = ref_tape[t]
                        got = tape_blocks(tape[j, t], n, m)
                        for key in ("Fx", "Fu", "Lxx", "Lxu", "Luu", "Lx", "Lu", "cost"):
                            if t == desc.T and key in ("Fx", "Fu", "Lxu", "Luu", "Lu"):
                                continue
                            r_ = rel(np.ravel(got[key]), np.ravel(ref[key]))
                            upd("tape_" + key, r_)
                            if r_ > max(tol_tape, NOISE_FACTOR * ref["noise"][key]) and key != "cost":
                                # Beyond the tolerance AND beyond what the oracle's own builds differ by.  Those builds run the SAME
                                # algorithm; on an ill-conditioned node (two point contacts near a stretched arm: cond(Jc M^-1 Jc^T) of
                                # 1e8 ... 1e10) they understate what two correct but DIFFERENT algorithms differ by.  Arbitration by a
                                # third algorithm: the NumPy restatement (articulated-body algorithm, dense KKT solve, complex-step
                                # derivatives) on this node.  If it sits as far from the oracle as the device does (within 10 x), the
                                # node is ill conditioned and the entry is counted, not asserted; otherwise the device is wrong.
                                try:
                                    r3 = third_algorithm_distance(desc, prm, it, t, key, ref[key])
                                except Exception as ex:  # (e.g. a singular dense KKT system): no arbiter, the plain assertion decides
                                    assert False, (where, t, key, r_, ref["noise"][key], "third algorithm unavailable: %s" % str(ex)[:200])
                                # (... or within what the conditioning of this node's constraint matrix explains on the device's
                                #  arithmetic: two algorithms that both round to half a unit can agree far better than either does with
                                #  a third that does not -- seed 97 of the two-contact soak, cond 3e8: device 2.8e-8, NumPy 1.3e-11)
                                cond_t = constraint_condition(desc, prm, it, t) if not r_ <= 10.0 * r3 else 0.0
                                rep.setdefault("tape_entries_arbitrated", []).append((b, i, t, key, float(r_), float(r3), float(cond_t)))
                                assert r_ <= max(10.0 * r3, COND_UNIT * cond_t), \
                                    (where, t, key, r_, ref["noise"][key], "NumPy restatement vs oracle", r3, "cond(Jc M^-1 Jc^T)", cond_t)
                                rep["tape_entries_excused_ill_conditioned"] = rep.get("tape_entries_excused_ill_conditioned", 0) + 1
                                excused_here += 1
                                continue
                            assert r_ <= max(tol_tape, NOISE_FACTOR * ref["noise"][key]), (where, t, key, r_, ref["noise"][key])
                    Ko, ko, Vxo, (Kf, kf_, Vxf) = ref_gains
                    ng = max(rel(Kf, Ko), rel(kf_, ko), rel(Vxf, Vxo))  # the oracle's own builds on these gains
                    upd("K", rel(K[j], Ko))
                    upd("k", rel(k[j], ko))
                    upd("Vx", rel(Vx[j], Vxo))
                    upd("gains_over_tol", max(rel(K[j], Ko), rel(k[j], ko), rel(Vx[j], Vxo)) / max(TOL_GAINS, NOISE_FACTOR * ng))
                    if excused_here:
                        # gains computed from a tape with an ill-conditioned node inherit its disagreement (the Riccati recursion
                        # carries it to every earlier knot): counted, not asserted -- the decisions of this iterate still are, below
                        rep["gains_excused_ill_conditioned_tape"] = rep.get("gains_excused_ill_conditioned_tape", 0) + 1
                    else:
                        assert max(rel(K[j], Ko), rel(k[j], ko), rel(Vx[j], Vxo)) <= max(TOL_GAINS, NOISE_FACTOR * ng), (where, ng)
                    rep["tapes_checked"] += 1
            # ---- the decision ---------------------------------------------------------------------------------------
            f = fin[j]
            ended = f.phase != it["phase"]
            if f.accepted_alpha != it["accepted_alpha"] and p["direction_ok"]:
                # legitimate only where a trial that takes part in the decision is one the oracle's own builds disagree on
                # (a rollout that blows up: costs of 1e11 ... 1e54): nothing about it is comparable
                hi = max(f.accepted_alpha if f.accepted_alpha >= 0 else na - 1, it["accepted_alpha"] if it["accepted_alpha"] >= 0 else na - 1)
                nz = np.abs(pf["cost_try"] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"]))
                for q in pu:
                    nz = np.maximum(nz, np.abs(q["cost_try"] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"])))
                if bool((~np.isfinite(nz[:hi + 1]) | (nz[:hi + 1] > CHAOTIC) | (pf["ok"][:hi + 1] != p["ok"][:hi + 1])).any()):
                    rep["decisions_excused_chaotic"] = rep.get("decisions_excused_chaotic", 0) + 1
                    rep["decisions_checked"] += 1
                    continue
                # ... or where the deciding inequality is tied to within the rounding noise of the trial costs that enter it
                # (an iterate that has blown up: every trial costs 8.1176e13, the acceptance test compares differences of 1e8)
                mg = decision_margin(prm, dict(it, accepted_alpha=hi), p, None)
                if mg is not None and mg <= NOISE_FACTOR * max(float(nz[:hi + 1].max()), 1e-13):
                    rep["decisions_excused_tied"] = rep.get("decisions_excused_tied", 0) + 1
                    rep["decisions_checked"] += 1
                    continue
            assert f.accepted_alpha == it["accepted_alpha"], (where, f.accepted_alpha, it["accepted_alpha"], cost_try[j], p["cost_try"])
            assert bool(ended) == bool(it["ended"]), (where, f.phase, it["phase"], it["ended"])
            if it["ended"]:
                assert bool(f.last_ok) == bool(it["returned"]), where
            if it["trace_index"] >= 0:
                r = paths[b]["trace"][it["trace_index"]]
                # record layout: phase iter cost stop xreg steplength feasible dV dVexp gapnorm d0 d1
                assert f.steplength == r[5], (where, f.steplength, r[5])
                if not ended:  # (the next pass starts from reg_init and an infeasible flag: src/sbfddp.cpp:210,232-235)
                    assert f.xreg == r[4], (where, f.xreg, r[4])
                    assert float(f.is_feasible) == r[6], (where, f.is_feasible, r[6])
                sc = 1.0 + abs(r[2])
                upd("cost_after", abs(f.cost - r[2]) / sc)
                assert abs(f.cost - r[2]) <= max(TOL_COST * amp * sc, NOISE_FACTOR * sn["cost"]), (where, f.cost, r[2], amp)
                upd("stop", abs(f.stop - r[3]) / (sc + abs(r[3])))
                assert abs(f.stop - r[3]) <= max(1e-8 * amp * (sc + abs(r[3])), NOISE_FACTOR * sn["stop"]), (where, f.stop, r[3], sn)
                if (p["ok"] != 0).any():  # (with every trial failing both sides keep the values of an earlier iteration)
                    upd("dV", abs(f.dV - r[7]) / (sc + abs(r[7])))
                    assert abs(f.dV - r[7]) <= max(1e-8 * amp * (sc + abs(r[7])), NOISE_FACTOR * sn["dV"]), (where, f.dV, r[7], sn)
                    assert abs(f.dVexp - r[8]) <= max(1e-7 * amp * (sc + abs(r[8])), NOISE_FACTOR * sn["dVexp"]), (where, f.dVexp, r[8], sn)
            # (an iteration without a record ended its pass at reg_max: `ended` and `returned` above are its whole outcome -- the
            #  status bits are cleared when the next pass starts)
            # the accepted candidate is the oracle's next iterate
            if it["accepted_alpha"] >= 0 and i + 1 < len(paths[b]["iterates"]):
                nxt = paths[b]["iterates"][i + 1]
                ex_ = np.abs(xs_new[j] - nxt["xs"]).max() / (1.0 + np.abs(nxt["xs"]).max())
                eu_ = np.abs(us_new[j] - nxt["us"]).max() / (1.0 + np.abs(nxt["us"]).max())
                upd("xs_next", ex_)
                upd("us_next", eu_)
                assert ex_ <= max(1e-7 * amp, NOISE_FACTOR * sn["xs"]) and eu_ <= max(1e-7 * amp, NOISE_FACTOR * sn["us"]), (where, amp, ex_, eu_, sn)
                if not np.isfinite(amp) and sn.get("xs_knot") is not None:
                    # the accepted trial is one the oracle's builds differ on by more than CHAOTIC: its cost carries no bound,
                    # but the candidate itself is comparable knot by knot up to the knot where the oracle's own variants part by
                    # more than PREFIX_SPLIT -- that prefix is held to max(1e-9, NOISE_FACTOR x the variants' distance there)
                    kx, ku = sn["xs_knot"], sn["us_knot"]
                    split = np.nonzero(kx > PREFIX_SPLIT)[0]
                    upto = int(split[0]) if len(split) else len(kx)
                    if upto > 0:
                        tx = np.maximum(1e-9 * (1.0 + np.abs(nxt["xs"][:upto]).max(axis=1)), NOISE_FACTOR * kx[:upto])
                        exk = np.abs(xs_new[j][:upto] - nxt["xs"][:upto]).max(axis=1)
                        assert (exk <= tx).all(), (where, "prefix xs", upto, exk.max(), tx.min())
                        nu_ = min(upto, len(ku))
                        if nu_ > 0:
                            tu = np.maximum(1e-9 * (1.0 + np.abs(nxt["us"][:nu_]).max(axis=1)), NOISE_FACTOR * ku[:nu_])
                            euk = np.abs(us_new[j][:nu_] - nxt["us"][:nu_]).max(axis=1)
                            assert (euk <= tu).all(), (where, "prefix us", nu_, euk.max(), tu.min())
                        rep["accepted_chaotic_prefix_checked"] = rep.get("accepted_chaotic_prefix_checked", 0) + 1
                        rep["accepted_chaotic_prefix_knots"] = rep.get("accepted_chaotic_prefix_knots", 0) + upto
                    else:
                        rep["accepted_chaotic_no_prefix"] = rep.get("accepted_chaotic_no_prefix", 0) + 1
                elif not np.isfinite(amp):
                    rep["accepted_chaotic_variants_disagree_on_step"] = rep.get("accepted_chaotic_variants_disagree_on_step", 0) + 1
                upd("amplification", amp)
            rep["decisions_checked"] += 1
            # margins of the inequalities that decided this iteration (for the near-tie study)
            probe[(b, i)] = decision_margin(prm, it, p, paths[b]["trace"][it["trace_index"]] if it["trace_index"] >= 0 else None)
        del be
    ms = [v for v in probe.values() if v is not None]
    rep["min_margin"] = float(min(ms)) if ms else None
    rep["margins"] = probe
    if report is not None:
        report.update(rep)
    return rep


def decision_margin(prm, it, p, rec):
    """Smallest relative distance to a tie among the inequalities that decided this oracle iteration
    (src/sbfddp.cpp:269-288 acceptance at every step length tried, :309 / :387 stopping test).  scale = 1 + |cost|."""
    if not p["direction_ok"]:
        return None
    ddp = it["phase"] == T.PHASE_DDP
    scale = 1.0 + abs(p["cost"])
    best = np.inf
    last = it["accepted_alpha"] if it["accepted_alpha"] >= 0 else prm.n_alphas - 1
    for a in range(last + 1):
        if not p["ok"][a]:
            continue
        alpha = 2.0 ** -a
        dV = p["cost"] - p["cost_try"][a]
        dVexp = alpha * (p["d0"][a] + 0.5 * alpha * p["d1"][a])
        cands = [abs(dVexp)]  # sign of dVexp picks the branch
        if dVexp >= 0:
            cands.append(abs(dV - prm.th_acceptstep * dVexp))
        elif not ddp:
            cands.append(abs(dV - prm.th_acceptnegstep * dVexp))
        best = min(best, min(cands) / scale)
    if rec is not None:
        th = it["th_stop"]
        best = min(best, abs(rec[3] - th) / scale)
        if not ddp and prm.solver_type == T.SOLVER_SBFDDP:
            best = min(best, abs(rec[9] - prm.th_stop_gaps) / (1.0 + prm.th_stop_gaps))
    return float(best)


def free_run(be, prm, x0s, maxiter=100, max_sweeps=2000, warm=None):
    """The device solver stepped sweep by sweep from solve([], [], maxiter)'s initial state, recording every iterate and
    decision of every rollout: the same kernels and state machine as empc_solver_solve, observable per iteration."""
    B = len(x0s)
    box = prm.solver_type != T.SOLVER_SBFDDP
    be.set_x0s(x0s)
    if box:
        be.set_gains(None, np.zeros((B, be.T, be.nu)))
    zero_xs = np.zeros((B, be.T + 1, be.nx))
    zero_xs[:, :, 6] = 1.0
    if warm is None:
        be.set_candidates(zero_xs, np.zeros((B, be.T, be.nu)))
    else:  # solve(init_xs, init_us, maxiter): the caller's initial guess instead of the zero state on every knot
        be.set_candidates(np.ascontiguousarray(warm[0]), np.ascontiguousarray(warm[1]))
    st = (T.TrajState * B)()
    for b in range(B):
        st[b] = fresh_state(prm, maxiter)
    be.set_states(st)
    hist = [[] for _ in range(B)]
    for _ in range(max_sweeps):
        cur = be.get_states()
        cur = [T.TrajState.from_buffer_copy(cur[b]) for b in range(B)]
        if all(c.phase == T.PHASE_DONE for c in cur):
            break
        xs, us = be.candidates()
        kk = be.gains()[1] if box else None
        be.sweep(T.STAGE_ALL)
        new = be.get_states()
        for b in range(B):
            if cur[b].phase == T.PHASE_DONE:
                continue
            n_ = T.TrajState.from_buffer_copy(new[b])
            e = dict(xs=xs[b].copy(), us=us[b].copy(), before=cur[b], after=n_)
            if box:
                e["k"] = kk[b].copy()
            hist[b].append(e)
    xs, us = be.candidates()
    fin = be.get_states()
    return hist, xs, us, [T.TrajState.from_buffer_copy(fin[b]) for b in range(B)]


def decisions_of_history(h):
    """discrete decision sequence of a stepped device run: (pass, iter, accepted alpha, xreg after, feasible after, pass ended,
    stop test passed); regularisation and feasibility are compared on iterations that do not end their pass (the next pass
    restarts both: src/sbfddp.cpp:210, 232-235)"""
    out = []
    for e in h:
        b_, a_ = e["before"], e["after"]
        ended = int(a_.phase != b_.phase)
        out.append((b_.phase, b_.iter, a_.accepted_alpha, None if ended else a_.xreg, None if ended else a_.is_feasible, ended,
                    int(a_.last_ok) if ended else 0))
    return out


def decisions_of_path(path):
    out = []
    for it in path["iterates"]:
        ended = it["ended"]
        if it["trace_index"] >= 0 and not ended:
            r = path["trace"][it["trace_index"]]
            out.append((it["phase"], it["iter"], it["accepted_alpha"], float(r[4]), int(r[6]), 0, 0))
        else:
            out.append((it["phase"], it["iter"], it["accepted_alpha"], None, None, ended, it["returned"] if ended else 0))
    return out


def first_divergence(hist_b, path):
    dg, do = decisions_of_history(hist_b), decisions_of_path(path)
    n = min(len(dg), len(do))
    for i in range(n):
        if dg[i] != do[i]:
            return i, dg[i], do[i]
    return (None, None, None) if len(dg) == len(do) else (n, None, None)


def reverse_teacher_forced(desc, prm, x0, hist_b, upto=None):
    """The other direction: the ORACLE is put at every iterate of the device's own free run and must reproduce the device's
    decision (one pass through the loop body of solveFDDP / solveDDP: oracle_iter_step).  Returns the list of iterations on
    which it does not, each with the margin of the closest inequality (a tie to rounding precision is the only legitimate
    reason)."""
    bad = []
    box = prm.solver_type != T.SOLVER_SBFDDP
    for i, e in enumerate(hist_b if upto is None else hist_b[:upto + 1]):
        b_, a_ = e["before"], e["after"]
        o = ob.OracleSolver(desc, prm)
        o.set_x0(x0)
        ddp = b_.phase == T.PHASE_DDP
        r = o.iter_step(e["xs"], e["us"], b_.is_feasible, b_.was_feasible, ddp, b_.xreg, b_.smooth, b_.th_stop, b_.cost,
                        b_.cost_prev, b_.iter, k=e.get("k") if box else None, upstream=(prm.solver_type == T.SOLVER_BOXFDDP))
        ended_g = int(a_.phase != b_.phase)
        ended_o = int(r["result"] != 0 or (r["result"] == 0 and b_.iter + 1 >= b_.maxiter))
        same = (r["accepted_alpha"] == a_.accepted_alpha) and (ended_g == ended_o)
        if same and ended_g:
            same = (int(a_.last_ok) == int(r["result"] > 0))
        if same and not ended_g:
            same = (r["xreg"] == a_.xreg) and (r["is_feasible"] == a_.is_feasible)
        if not same:
            # a legitimate reason exists only when the deciding trial is one on which the oracle's own builds disagree
            # (a rollout that blows up) or an inequality is tied to rounding precision
            kk = e.get("k") if box else None
            o2 = ob.OracleSolver(desc, prm)
            o2.set_x0(x0)
            p = o2.iter_probe(e["xs"], e["us"], b_.is_feasible, b_.was_feasible, ddp, b_.xreg, b_.smooth, k=kk)
            of = ob.OracleSolver(desc, prm, variant="fma")
            of.set_x0(x0)
            pf = of.iter_probe(e["xs"], e["us"], b_.is_feasible, b_.was_feasible, ddp, b_.xreg, b_.smooth, k=kk)
            hi = max(a_.accepted_alpha if a_.accepted_alpha >= 0 else prm.n_alphas - 1,
                     r["accepted_alpha"] if r["accepted_alpha"] >= 0 else prm.n_alphas - 1)
            noise = np.abs(pf["cost_try"] - p["cost_try"]) / (1.0 + np.abs(p["cost_try"]))
            chaotic = bool((~np.isfinite(noise[:hi + 1]) | (noise[:hi + 1] > CHAOTIC) | (pf["ok"][:hi + 1] != p["ok"][:hi + 1])).any()) \
                or not (p["direction_ok"] and pf["direction_ok"])
            it = dict(phase=b_.phase, accepted_alpha=r["accepted_alpha"], th_stop=b_.th_stop)
            margin = decision_margin(prm, it, p, None) if p["direction_ok"] else None
            bad.append(dict(iteration=i, device=(a_.accepted_alpha, a_.xreg, a_.is_feasible, ended_g, int(a_.last_ok)),
                            oracle=(r["accepted_alpha"], r["xreg"], r["is_feasible"], ended_o, int(r["result"] > 0)),
                            chaotic_trial=chaotic, margin=margin, noise=float(np.nanmax(noise[:hi + 1])),
                            xmax=float(max(np.abs(e["xs"][:, 7:]).max(), np.abs(e["xs"][:, :3]).max())), cost=float(b_.cost)))
    return bad


def same_minimum(be_factory, desc, prm, x0s, xs0, us0, tight=1e-9, maxiter=300):
    """Both solvers restarted from the SAME point (the device's final trajectories) with the convergence threshold
    tightened to `tight` (one pass at the final smoothing): do they reach the same minimiser?  Returns per-rollout dicts."""
    prm2 = T.SolverParams.from_buffer_copy(prm)
    if prm.solver_type == T.SOLVER_SBFDDP:
        prm2.smooth_init = prm.smooth_init * prm.smooth_mult
        prm2.convergence_init = prm2.convergence_stop = tight
    else:
        prm2.box_th_stop = tight
    B = len(x0s)
    be = be_factory(B, prm2)
    be.set_x0s(x0s)
    if prm.solver_type != T.SOLVER_SBFDDP:
        be.set_gains(None, np.zeros((B, be.T, be.nu)))
    be.set_candidates(xs0, us0)
    st = (T.TrajState * B)()
    for b in range(B):
        st[b] = fresh_state(prm2, maxiter)
    be.set_states(st)
    for _ in range(4 * maxiter + 16):
        be.sweep(T.STAGE_ALL)
        cur = be.get_states()
        if all(cur[b].phase == T.PHASE_DONE for b in range(B)):
            break
    xs, us = be.candidates()
    fin = be.get_states()

    def orc(b):
        o = ob.OracleSolver(desc, prm2)
        o.set_x0(x0s[b])
        o.solve(xs0[b], us0[b], maxiter)
        return o.result()
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 32)) as pool:
        refs = list(pool.map(orc, range(B)))
    # the controls the platform receives: sigma(us) with the final smoothing (squashed problems; the box solvers' us are the
    # controls themselves).  In a saturated direction sigma'(s) ~ 0 and the unsquashed s is a flat direction of the problem.
    lb = np.array([desc.u_lb[i] for i in range(desc.nu)])
    ub = np.array([desc.u_ub[i] for i in range(desc.nu)])

    def sigma(s_):
        if not desc.use_squash:
            return s_
        a_ = (prm2.smooth_init * (ub - lb)) ** (4 if prm2.smoothsat_power == 4 else 2)
        return 0.5 * (np.sqrt((s_ - lb) ** 2 + a_) - np.sqrt((s_ - ub) ** 2 + a_) + ub + lb)
    out = []
    for b in range(B):
        r = refs[b]
        out.append(dict(iters_device=int(fin[b].iter), iters_oracle=int(r["iter"]), status_device=int(fin[b].status),
                        status_oracle=int(r["status"]), cost_device=float(fin[b].cost), cost_oracle=float(r["cost"]),
                        xs_err=float(np.abs(xs[b] - r["xs"]).max()), us_err=float(np.abs(us[b] - r["us"]).max()),
                        usq_err=float(np.abs(sigma(us[b]) - sigma(r["us"])).max()),
                        moved=float(np.abs(xs[b] - xs0[b]).max())))
    return out


def stepwise_parity(backend_factory, desc, prm, x0s, maxiter=100, chunk=1024, tape_every=37, do_same_minimum=True,
                    tight=1e-9, tight_maxiter=300, tol_tape=TOL_TAPE, warm=None):
    """The whole argument for one problem and one batch of initial states:
      1. the device reproduces EVERY iteration of the oracle's own paths (teacher_forced),
      2. the oracle reproduces every iteration of the device's own free-running paths (reverse_teacher_forced), except where
         the deciding trial is one the oracle's own builds disagree on (a rollout that blows up) or an inequality is tied
         to rounding precision,
      3. restarted from the device's final points with a tight threshold both reach the same minimiser (same_minimum).
    1 + 2 prove that wherever two free-running paths part ways, each side's decision is the other's decision on the same
    inputs: the paths differ through accumulated rounding (drift), not through a different rule.  Returns a report;
    raises AssertionError when a claim fails.  backend_factory(batch, params=None) -> backend."""
    B = len(x0s)
    # warm = (xs [B, T+1, nx], us [B, T, nu]): both sides start from this initial guess (SolverSbFDDP::solve(init_xs, init_us, ...),
    # is_feasible = false) instead of the default one -- for problems whose default guess is a singular configuration
    paths = oracle_paths(desc, prm, x0s, maxiter, None if warm is None else (warm[0], warm[1], False))
    rep = teacher_forced(lambda n: backend_factory(n, None), desc, prm, x0s, paths, maxiter=maxiter, chunk=chunk,
                         tape_every=tape_every, tol_tape=tol_tape)
    margins = rep.pop("margins")
    rep.pop("beyond", None)
    be = backend_factory(B, None)
    hist, xs, us, fin = free_run(be, prm, x0s, maxiter, warm=warm)
    del be
    div, unexplained, excused, exploded = [], [], 0, 0

    def rev(b):
        return reverse_teacher_forced(desc, prm, x0s[b], hist[b])
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 32)) as pool:
        bads = list(pool.map(rev, range(B)))
    same_path = 0
    for b in range(B):
        fd, g, o = first_divergence(hist[b], paths[b])
        if fd is None:
            same_path += 1
        elif fd < min(len(hist[b]), len(paths[b]["iterates"])):
            drift = float(np.abs(hist[b][fd]["xs"] - paths[b]["iterates"][fd]["xs"]).max())
            div.append(dict(rollout=b, iteration=fd, device=g, oracle=o, drift_xs=drift, margin_oracle=margins.get((b, fd))))
        for m_ in bads[b]:
            # (a tie: the two sides of the deciding inequality differ by less than TIE = 10 x TOL_COST relative -- the trial
            #  costs that enter it are themselves only compared at TOL_COST)
            if m_["xmax"] > BLOWN_UP or abs(m_["cost"]) > COST_EXPLODED:
                exploded += 1  # an iterate of a rollout that has exploded (rates of 1e6, costs of 1e15): as in teacher_forced
            elif m_["chaotic_trial"] or (m_["margin"] is not None and m_["margin"] <= TIE):
                excused += 1
            else:
                unexplained.append(dict(rollout=b, **m_))
    rep["free_run"] = dict(rollouts=B, same_path_as_oracle=same_path, diverging=len(div),
                           device_iterations=int(sum(len(h) for h in hist)),
                           oracle_reproduces_device_decision=int(sum(len(h) for h in hist)) - excused - exploded - len(unexplained),
                           excused_chaotic_or_tied=excused, skipped_exploded_iterates=exploded, unexplained=len(unexplained), first_divergences=div)
    assert not unexplained, unexplained[:5]
    final = np.array([paths[b]["result"]["xs"] for b in range(B)])
    rep["free_run"]["final_xs_err_same_path_max"] = float(max(
        [np.abs(xs[b] - final[b]).max() for b in range(B) if first_divergence(hist[b], paths[b])[0] is None] or [0.0]))
    if do_same_minimum:
        good = [b for b in range(B) if fin[b].phase == T.PHASE_DONE and np.isfinite(xs[b]).all() and np.abs(xs[b]).max() < 1e3
                and (fin[b].status & T.STATUS_CONVERGED)]
        if good:
            sm = same_minimum(lambda n, p2: backend_factory(n, p2), desc, prm, x0s[good], xs[good], us[good], tight=tight,
                              maxiter=tight_maxiter)
            both = [r for r in sm if (r["status_oracle"] & 1) and not (r["status_oracle"] & 6)]
            rep["same_minimum"] = dict(rollouts=len(good), converged_on_oracle=len(both), tight=tight,
                                       xs_err_max=max([r["xs_err"] for r in both] or [0.0]),
                                       us_err_max=max([r["us_err"] for r in both] or [0.0]),
                                       us_squash_err_max=max([r["usq_err"] for r in both] or [0.0]),
                                       iterations_equal=int(sum(r["iters_device"] == r["iters_oracle"] for r in both)),
                                       moved_from_plain_solution_max=max([r["moved"] for r in both] or [0.0]))
            for r in both:
                # the north-star bound at the common minimiser: states and the controls the platform receives; the unsquashed
                # controls within 1e-3 (flat where the squashing saturates: 1.3e-4 seen on one displacement rollout of a soak run)
                assert r["xs_err"] <= 1e-4 and r["usq_err"] <= 1e-4 and r["us_err"] <= 1e-3, r
    return rep


def select_in_isolation(backend_factory, desc, prm, x0s, paths, pairs, maxiter=100):
    """select alone: the device's decision stage is fed the ORACLE's numbers (cost, gap norm, expected-improvement sums,
    trial costs and gap terms of every step length, from oracle_iter_probe) and must return the oracle's decision and
    scalars bit for bit -- the state machine itself, with no device arithmetic in front of it."""
    B = len(pairs)
    be = backend_factory(B)
    st = (T.TrajState * B)()
    na = prm.n_alphas
    ok = np.zeros((B, na), dtype=np.int32)
    cost_try = np.zeros((B, na))
    dv = np.zeros((B, na))
    keep = []
    for j, (b, i) in enumerate(pairs):
        it = paths[b]["iterates"][i]
        ddp = it["phase"] == T.PHASE_DDP
        o = ob.OracleSolver(desc, prm)
        o.set_x0(x0s[b])
        p = o.iter_probe(it["xs"], it["us"], it["is_feasible"], it["was_feasible"], ddp, it["xreg"], it["smooth"])
        s_ = state_at_iterate(prm, it, maxiter)
        s_.need_lin = s_.need_calc = 0
        s_.cost = p["cost"]
        s_.is_feasible = int(p["is_feasible"])
        s_.gapnorm = p["gapnorm"]
        s_.xreg = s_.ureg = p["xreg"]
        s_.bwd_failed = 0 if p["direction_ok"] else 1
        # the sums as the backward pass leaves them: control part / gap part (the oracle reports their total)
        s_.dg_u, s_.dq_u, s_.dg_f, s_.dq_f = p["dg"], p["dq"], 0.0, 0.0
        feas = p["is_feasible"]
        ok[j] = p["ok"]
        cost_try[j] = p["cost_try"]
        # d0 = dg + dv, d1 = dq - 2 dv  (SolverFDDP::expectedImprovement): recover dv of every trial
        dv[j] = 0.0 if (feas or ddp) else (p["d0"] - p["dg"])
        st[j] = s_
        keep.append((it, p))
    be.set_states(st)
    be.select(ok, cost_try, dv)
    fin = be.get_states()
    n_ok = 0
    for j, (b, i) in enumerate(pairs):
        it, p = keep[j]
        f = fin[j]
        where = "rollout %d iterate %d" % (b, i)
        assert f.accepted_alpha == it["accepted_alpha"], (where, f.accepted_alpha, it["accepted_alpha"])
        ended = f.phase != it["phase"]
        assert bool(ended) == bool(it["ended"]), where
        if it["ended"]:
            assert bool(f.last_ok) == bool(it["returned"]), where
        if it["trace_index"] >= 0:
            r = paths[b]["trace"][it["trace_index"]]
            assert f.steplength == r[5] and f.cost == r[2], (where, f.cost, r[2])
            assert f.stop == r[3] and f.dV == r[7], (where, f.stop, r[3], f.dV, r[7])
            if not ended:
                assert f.xreg == r[4] and float(f.is_feasible) == r[6], where
        n_ok += 1
    return n_ok
