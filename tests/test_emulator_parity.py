"""The HIP kernel bodies (eagle-mpc_amd/csrc/empc_*.hpp), executed lane by lane by tests/csrc/lane_emulator.cpp, against
the oracle.  This is the strongest check available without a GPU: identical templates, identical index arithmetic and
LDS staging; only the wave execution is emulated.  The emulator is test infrastructure (never loaded by the package)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# EMU_MACROS="EMPC_BWD_SYMTILES=1 ...": the emulator of a build-time VARIANT of the kernel bodies (empc_variants.hpp), in a library
# of its own -- any emulator test or tool runs on a variant this way (tools/variant_verdicts.py)
EMU_MACROS = os.environ.get("EMU_MACROS", "").split()
EMU = os.path.join(ROOT, "tests", "csrc", "liblane_emulator%s.so" % ("".join("_" + m.replace("=", "").replace("EMPC_", "").lower() for m in EMU_MACROS)))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def emu(empc):
    src = os.path.join(ROOT, "tests", "csrc", "lane_emulator.cpp")
    hdrs = [os.path.join(ROOT, "eagle-mpc_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "eagle-mpc_amd", "csrc"))
            if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "empc_types.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(h) > os.path.getmtime(EMU) for h in hdrs + [src]):
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include")] + ["-D" + m for m in EMU_MACROS] + [src,
                               "-o", EMU])
    L = C.CDLL(EMU)
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.POINTER(empc.T.ProblemDesc), C.POINTER(empc.T.SolverParams), C.c_int]
    L.emu_destroy.argtypes = [C.c_void_p]
    L.emu_rec.argtypes = [C.c_void_p]
    L.emu_set_x0.argtypes = [C.c_void_p, _dp]
    L.emu_set_warmstart.argtypes = [C.c_void_p, _dp, _dp]
    L.emu_solve_c.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.emu_get.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip]
    L.emu_phase_setup.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int]
    L.emu_phase_linearize.argtypes = [C.c_void_p, _dp, _dp]
    L.emu_phase_backward.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
    L.emu_phase_rollout.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip]
    L.emu_node_nominal.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
    return L


def rel(a, b):
    return np.abs(a - b).max() / (1.0 + np.abs(b).max())


def candidate(d, seed, scale=1.0):
    """random candidate trajectory; scale < 1 keeps it near hover (an RK4 step is unstable on the stiff arm dynamics of the
    full-size perturbation: |da/dx| dt >> 1 there, Fx entries of 1e5)"""
    rng = np.random.default_rng(seed)
    T, nx, nu = d.T, d.nx, d.nu
    xs = np.zeros((T + 1, nx))
    xs[:, :3] = rng.normal(size=(T + 1, 3)) * 0.3
    q = np.array([0, 0, 0, 1.0]) + rng.normal(size=(T + 1, 4)) * 0.2 * scale
    xs[:, 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    xs[:, 7:] = rng.normal(size=(T + 1, nx - 7)) * 0.3 * scale
    us = rng.uniform(2, 6, size=(T, nu)) if scale == 1.0 else 4.0 + rng.uniform(-1, 1, size=(T, nu)) * scale
    us[:, d.n_rotors:] = rng.normal(size=(T, nu - d.n_rotors)) * 0.2 * scale
    return xs, us


@pytest.mark.parametrize("lin,bwd,roll", [(2, 4, 6), (2, 3, 5), (2, 2, 5), (2, 1, 1)])
@pytest.mark.parametrize("name", ["hover", "displacement", "push_slide", "eagle_catch"])
def test_kernel_bodies_vs_oracle(empc, problems, emu, name, lin, bwd, roll):
    _, problem = problems[name]
    kernel_bodies(emu, problem, name, lin, bwd, roll)


@pytest.mark.parametrize("contact,gains", [("ContactModel3D", (9.0, 4.0)), ("ContactModel6D", (0.0, 0.0)), ("ContactModel6D", (11.0, 5.0))])
def test_contact_options_kernel_bodies_vs_oracle(empc, emu, tmp_path, contact, gains):
    """The contact factory's other options (src/factory/contacts.cpp:26-79) through the device code: ContactModel6D (six
    constraint rows: its own kernel instantiation) and Baumgarte gains, tape / gains / rollouts against the oracle."""
    from conftest import contact_variant
    _, problem = contact_variant(empc, tmp_path, contact, gains)
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6)


@pytest.mark.parametrize("contact,gains", [("ContactModel3D", (7.0, 3.0)), ("ContactModel6D", (0.0, 0.0))])
def test_arm5_contact_kernel_bodies_vs_oracle(empc, emu, tmp_path, contact, gains):
    """Contact dynamics on the 11-dof arm class (64 lanes per linearize unit; empc_inst_6_6_contact*.hip) through the kernel
    bodies: tape / gains / rollouts against the oracle.  The GPU-only fault of round 4 lived in code only this class runs."""
    from conftest import arm5_contact_variant
    _, problem = arm5_contact_variant(empc, tmp_path, contact, gains)
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6)  # ("eagle_catch": phases only, no emulated solve)


@pytest.mark.parametrize("contact,gains", [("ContactModel3D", (0.0, 0.0)), ("ContactModel6D", (5.0, 2.0))])
@pytest.mark.parametrize("robot", ["hexacopter370", "iris", "hexacopter680_flying_arm_2"])
def test_small_class_contact_kernel_bodies_vs_oracle(empc, emu, tmp_path, monkeypatch, robot, contact, gains):
    """Contact dynamics on the robot classes (1, 6), (1, 4), (3, 6) -- empc_inst_{1_6,1_4,3_6}_contact.hip, both contact types
    behind the branch of the mixed instantiation: tape / gains / rollouts of the kernel bodies against the oracle, and the
    factory's answer: refused with the reason until the opt-in is set (the kernels have not run on hardware yet), kernels
    found with it."""
    from conftest import small_class_contact_variant
    _, problem = small_class_contact_variant(empc, tmp_path, robot, contact, gains)
    monkeypatch.delenv("EMPC_EXPERIMENTAL_CONTACT", raising=False)
    assert not empc.solver_supported(problem) and "EMPC_EXPERIMENTAL_CONTACT" in empc.last_error()
    monkeypatch.setenv("EMPC_EXPERIMENTAL_CONTACT", "1")
    assert empc.solver_supported(problem), empc.last_error()
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6)  # ("eagle_catch": phases only, no emulated solve)


def test_arm5_mixed_contact_kernel_bodies_vs_oracle(empc, emu, tmp_path, monkeypatch):
    """Stages of BOTH contact types on the (6, 6) robot class (empc_inst_6_6_contact_mixed.hip): refused with the reason until the
    opt-in is set (never run on hardware), accepted with it; tape / gains / rollouts of the kernel bodies against the oracle."""
    from conftest import arm5_mixed_contact_variant
    _, problem = arm5_mixed_contact_variant(empc, tmp_path)
    d = problem.desc
    assert {d.sets[d.knot_set[t]].contacts[0].type for t in range(d.T + 1) if d.sets[d.knot_set[t]].ncontacts > 0} == {empc.T.CONTACT_3D, empc.T.CONTACT_6D}
    monkeypatch.delenv("EMPC_EXPERIMENTAL_CONTACT", raising=False)
    assert not empc.solver_supported(problem) and "EMPC_EXPERIMENTAL_CONTACT" in empc.last_error()
    monkeypatch.setenv("EMPC_EXPERIMENTAL_CONTACT", "1")
    assert empc.solver_supported(problem), empc.last_error()
    kernel_bodies(emu, problem, "eagle_catch", 2, 4, 6)  # ("eagle_catch": phases only, no emulated solve)


@pytest.fixture(scope="module")
def emu_baked(empc):
    """the lane emulator built over the BAKED robot tables (-DEMU_BAKED, tests/csrc/lane_emulator.cpp): the kernel family the
    shipped library runs for every shipped robot"""
    global EMU
    src = os.path.join(ROOT, "tests", "csrc", "lane_emulator.cpp")
    lib = EMU.replace(".so", "_baked.so")
    hdrs = [os.path.join(ROOT, "eagle-mpc_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "eagle-mpc_amd", "csrc")) if f.endswith(".hpp")]
    hdrs += [os.path.join(ROOT, "eagle-mpc_amd", "csrc", "baked", "empc_baked_models.hpp"), os.path.join(ROOT, "include", "empc_types.h"), src]
    if not os.path.exists(lib) or any(os.path.getmtime(h) > os.path.getmtime(lib) for h in hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-DEMU_BAKED", "-I" + os.path.join(ROOT, "include")] + ["-D" + m for m in EMU_MACROS] + [src, "-o", lib])
    keep, EMU = EMU, lib
    try:
        return emu.__wrapped__(empc)
    finally:
        EMU = keep


@pytest.mark.parametrize("name", ["displacement", "eagle_catch", "push_slide", "hover"])
def test_baked_family_equals_runtime_family(empc, problems, emu, emu_baked, name):
    """The kernel bodies instantiated over the baked constants of a shipped robot (what libempc.so runs by default) against the
    runtime-model instantiation of the same source, on the CPU: tape, gains, Vx, expected-improvement sums and six step lengths
    of the rollout from a random candidate with open gaps -- bit for bit (on the host a product with a structural zero is
    still computed, so the two families perform the same operations; on the device the baked one folds them away and the
    families agree to rounding, tests/test_gpu_baked.py).  Covers the generated tables and the BakedView / BakedPlatform code
    paths without a GPU."""
    _, problem = problems[name]
    families_equal(emu, emu_baked, problem)


def families_equal(emu, emu_baked, problem):
    """tape, gains, Vx, expected-improvement sums and six step lengths of the rollout, bit for bit between the runtime-model and the
    baked instantiation of the kernel bodies (both on the CPU lane emulator)"""
    d = problem.desc
    prm = ob.default_params()
    T, nx, nu, nv = d.T, d.nx, d.nu, d.model.nv
    out = []
    for L in (emu, emu_baked):
        L.emu_set_linearize_version(2)
        L.emu_set_backward_version(4)
        L.emu_set_rollout_version(6)
        e = C.c_void_p(L.emu_create(C.byref(d), C.byref(prm), 1))
        assert e.value, "the baked build refused a shipped robot"
        xs, us = candidate(d, 5, scale=0.3)
        L.emu_set_warmstart(e, ob.P(xs), ob.P(us))
        L.emu_phase_setup(e, 0.1, 0, 1e-9, 0)
        tape, acc = np.zeros((T + 1, L.emu_rec(e))), np.zeros((T + 1, nv))
        L.emu_phase_linearize(e, ob.P(tape), ob.P(acc))
        K, k, Vx, dg = np.zeros((T, nu, d.ndx)), np.zeros((T, nu)), np.zeros((T + 1, d.ndx)), np.zeros(2)
        ok, fe, ce = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(1)
        L.emu_phase_backward(e, ob.P(K), ob.P(k), ob.P(Vx), ob.P(dg), ok.ctypes.data_as(_ip), fe.ctypes.data_as(_ip), ob.P(ce))
        res = [tape, acc, K, k, Vx, dg, ok, ce]
        finite = 0
        for ai in (1, 3, 5, 7, 8, 9):
            xt, ut, ct, dv, okr = np.zeros((T + 1, nx)), np.zeros((T, nu)), np.zeros(1), np.zeros(1), np.zeros(1, dtype=np.int32)
            L.emu_phase_rollout(e, ai, ob.P(xt), ob.P(ut), ob.P(ct), ob.P(dv), okr.ctypes.data_as(_ip))
            res += [xt, ut, ct, dv, okr]
            finite += int(np.isfinite(xt).all(axis=1).sum())
        L.emu_destroy(e)
        out.append((res, finite))
    (a, fa), (b, fb) = out
    assert np.isfinite(a[0]).all() and a[6][0] == 1 and fa == fb and fa > 2 * (T + 1)  # (at least two trials finite to the end)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def test_unweighted_quadratic_barrier_kernel_bodies(empc, emu, tmp_path):
    """ActivationModelQuadraticBarrier (bounds, no weights; src/factory/activation.cpp:53-68) through the kernel bodies"""
    from conftest import unweighted_barrier_variant
    _, problem = unweighted_barrier_variant(empc, tmp_path)
    kernel_bodies(emu, problem, "displacement", 2, 4, 6)


def kernel_bodies(emu, problem, name, lin, bwd, roll, tape_tol=1e-11, gain_tol=1e-6, vx_tol=1e-8):
    d = problem.desc
    prm = ob.default_params()
    emu.emu_set_linearize_version(lin)
    emu.emu_set_backward_version(bwd)
    emu.emu_set_rollout_version(roll)
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    assert e.value
    o = ob.OracleSolver(d)
    T, nx, ndx, nu, nv = d.T, d.nx, d.ndx, d.nu, d.model.nv
    rec = emu.emu_rec(e)
    xs, us = candidate(d, 3)
    o.set_smooth(0.1)
    cost_o, fs, feas = o.phase_calcdiff(xs, us)
    emu.emu_set_warmstart(e, ob.P(xs), ob.P(us))
    emu.emu_phase_setup(e, 0.1, 0, 1e-9, 0)
    tape = np.zeros((T + 1, rec))
    acc = np.zeros((T + 1, nv))
    emu.emu_phase_linearize(e, ob.P(tape), ob.P(acc))
    n, m = ndx, nu
    nm = n + m

    def blocks(r):  # record layout of empc_dev_model.hpp (Dims): A = [Fx Fu], HX = [Lxx Lxu], LUU, LX, LU, GAP, COST
        A = r[:n * nm].reshape(n, nm)
        HX = r[n * nm:2 * n * nm].reshape(n, nm)
        o2 = 2 * n * nm
        return {"Fx": A[:, :n], "Fu": A[:, n:], "Lxx": HX[:, :n], "Lxu": HX[:, n:], "Luu": r[o2:o2 + m * m].reshape(m, m),
                "Lx": r[o2 + m * m:o2 + m * m + n], "Lu": r[o2 + m * m + n:o2 + m * m + n + m],
                "gap": r[o2 + m * m + n + m:o2 + m * m + 2 * n + m], "cost": r[o2 + m * m + 2 * n + m:o2 + m * m + 2 * n + m + 1]}
    for t in range(T + 1):
        ref = o.phase_tape(t)
        ref["gap"] = fs[t]
        ref["cost"] = np.array([ref["cost"]])
        got = blocks(tape[t])
        for key in got:
            if t == T and key in ("Fx", "Fu", "Lxu", "Luu", "Lu"):
                continue
            assert rel(np.asarray(got[key]).ravel(), np.asarray(ref[key]).ravel()) < tape_tol, (t, key)
    ok, Ko, ko, Vxo, _, dgo = o.phase_backward(1e-9)
    K = np.zeros((T, m, n))
    k = np.zeros((T, m))
    Vx = np.zeros((T + 1, n))
    dg = np.zeros(2)
    oke = np.zeros(1, dtype=np.int32)
    fe = np.zeros(1, dtype=np.int32)
    ce = np.zeros(1)
    emu.emu_phase_backward(e, ob.P(K), ob.P(k), ob.P(Vx), ob.P(dg), oke.ctypes.data_as(_ip), fe.ctypes.data_as(_ip), ob.P(ce))
    assert ok and oke[0] == 1 and fe[0] == int(feas)
    assert rel(K, Ko) < gain_tol and rel(k, ko) < gain_tol and rel(Vx, Vxo) < vx_tol and np.allclose(dg, dgo, rtol=1e-7 * max(1.0, gain_tol / 1e-6))
    assert abs(ce[0] - cost_o) < 1e-10 * (1 + abs(cost_o))
    for ai in (2, 4):
        oko, xo, uo, co, d01 = o.phase_forward(2.0 ** -ai)
        xt = np.zeros((T + 1, nx))
        ut = np.zeros((T, nu))
        ct = np.zeros(1)
        dv = np.zeros(1)
        okr = np.zeros(1, dtype=np.int32)
        emu.emu_phase_rollout(e, ai, ob.P(xt), ob.P(ut), ob.P(ct), ob.P(dv), okr.ctypes.data_as(_ip))
        assert bool(okr[0]) == oko
        if oko and abs(co) < 1e12:
            assert rel(xt, xo) < 1e-7 and rel(ut, uo) < 1e-7 and abs(ct[0] - co) < 1e-7 * (1 + abs(co))
            assert np.allclose([dg[0] + dv[0], dg[1] - 2 * dv[0]], d01, rtol=1e-6, atol=1e-6 * abs(d01).max())
    if name == "eagle_catch":  # contact problem: 64 iterations, too slow for the lane-by-lane emulator; phases only
        emu.emu_destroy(e)
        return
    # full solve through the emulated kernels (state machine `select_decide` included)
    emu.emu_set_warmstart(e, None, None)
    emu.emu_solve_c(e, 100, 0)
    xs_e = np.zeros((T + 1, nx))
    us_e = np.zeros((T, nu))
    ul = np.zeros((T, nu))
    it = np.zeros(1, dtype=np.int32)
    st = np.zeros(1, dtype=np.int32)
    emu.emu_get(e, ob.P(xs_e), ob.P(us_e), ob.P(ul), ob.P(ce), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    o.solve(None, None, 100)
    r = o.result()
    assert it[0] == r["iter"] and st[0] == r["status"]
    assert np.abs(xs_e - r["xs"]).max() < 1e-6 and np.abs(us_e - r["us"]).max() < 1e-6
    assert abs(ce[0] - r["cost"]) < 1e-9 * (1 + abs(r["cost"]))
    emu.emu_destroy(e)


def test_emulated_batch_with_perturbed_states(empc, problems, emu):
    _, problem = problems["displacement"]
    d = problem.desc
    B = 3
    prm = ob.default_params()
    emu.emu_set_linearize_version(2)
    emu.emu_set_backward_version(4)
    emu.emu_set_rollout_version(6)  # three trajectories packed into one wavefront
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), B))
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq)
    emu.emu_set_x0(e, ob.P(x0s))
    emu.emu_set_warmstart(e, None, None)
    emu.emu_solve_c(e, 100, 0)
    xs = np.zeros((B, d.T + 1, d.nx))
    cost = np.zeros(B)
    it = np.zeros(B, dtype=np.int32)
    st = np.zeros(B, dtype=np.int32)
    emu.emu_get(e, ob.P(xs), None, None, ob.P(cost), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    ref = ob.solve_batch(d, x0s, 100, nthreads=2)
    assert (it == ref["iter"]).all() and (st == ref["status"]).all()
    assert np.abs(xs - ref["xs"]).max() < 1e-5
    emu.emu_destroy(e)


@pytest.mark.parametrize("integrator", ["IntegratedActionModelEuler", "IntegratedActionModelRK4"])
@pytest.mark.parametrize("name,dt", [("hover", 40), ("displacement", 80), ("eagle_catch", 32)])
def test_node_nominal_vs_oracle(empc, emu, name, dt, integrator):
    """IAM.calc of single nodes through the device code (node_nominal: dam_nominal + Euler step, or the four RK4 stages)
    against the oracle's node_calc: next state, stage-0 acceleration / contact force / squashed control, cost."""
    from conftest import CONFIGS
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = t.createProblem(dt, True, integrator)
    d = problem.desc
    prm = ob.default_params()
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    assert e.value
    o = ob.OracleSolver(d)
    o.set_smooth(0.07)
    rng = np.random.default_rng(5)
    knots = {"hover": [0, 49, 50], "displacement": [2, 25, 100], "eagle_catch": [3, 43, 46, 99]}[name]
    for tk in knots:
        x = np.zeros(d.nx)
        x[:3] = rng.normal(size=3) * 0.5
        q = np.array([0, 0, 0, 1.0]) + rng.normal(size=4) * 0.3
        x[3:7] = q / np.linalg.norm(q)
        x[7:] = rng.normal(size=d.nx - 7) * 0.2
        u = rng.uniform(1, 8, size=d.nu)
        u[d.n_rotors:] = rng.normal(size=d.nu - d.n_rotors) * 0.3
        uu = None if tk == d.T else u
        r = o.node_calc(tk, x, uu, False)
        xn, acc, cost, usq, lam = np.zeros(d.nx), np.zeros(d.model.nv), np.zeros(1), np.zeros(d.nu), np.zeros(6)
        emu.emu_node_nominal(e, tk, ob.P(x), None if uu is None else ob.P(u), 0.07, ob.P(xn), ob.P(acc), ob.P(cost), ob.P(usq), ob.P(lam))
        if uu is not None:
            assert np.abs(xn - r["xnext"]).max() < 1e-12 * (1 + np.abs(r["xnext"]).max()), (name, tk)
        assert np.abs(acc - r["acc"]).max() < 1e-10 * (1 + np.abs(r["acc"]).max())
        assert abs(cost[0] - r["cost"]) < 1e-12 * (1 + abs(r["cost"]))
        assert np.abs(usq - r["u_squash"]).max() < 1e-13 and np.abs(lam - r["lam"]).max() < 1e-9 * (1 + np.abs(r["lam"]).max())
    emu.emu_destroy(e)


@pytest.mark.parametrize("name,dt", [("hover", 40), ("displacement", 80), ("eagle_catch", 32)])
def test_rk4_kernel_bodies_vs_oracle(empc, emu, name, dt):
    """IntegratedActionModelRK4 through the device code on the CPU lane emulator: stage kernel + linearize in RAW mode +
    assembly kernel against the oracle's RK4 calcDiff (every block of every node), backward + rollout on that tape, and a
    whole solve (iteration count, status, trajectory)."""
    from conftest import CONFIGS
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = tr.createProblem(dt, True, "IntegratedActionModelRK4")
    kernel_bodies_rk4(emu, problem, name)


def kernel_bodies_rk4(emu, problem, name="eagle_catch", tape_tol=1e-10, seed=3):
    """phases of an RK4 problem against the oracle; a whole solve as well unless `name` is eagle_catch (contact problems: too slow
    lane by lane)"""
    d = problem.desc
    prm = ob.default_params()
    emu.emu_set_linearize_version(2)
    emu.emu_set_backward_version(4)
    emu.emu_set_rollout_version(6)  # the role-split form, four stages per knot
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    assert e.value
    o = ob.OracleSolver(d)
    T, nx, ndx, nu, nv = d.T, d.nx, d.ndx, d.nu, d.model.nv
    rec = emu.emu_rec(e)
    xs, us = candidate(d, seed, scale=0.1)
    o.set_smooth(0.1)
    cost_o, fs, feas = o.phase_calcdiff(xs, us)
    emu.emu_set_warmstart(e, ob.P(xs), ob.P(us))
    emu.emu_phase_setup(e, 0.1, 0, 1e-9, 0)
    tape = np.zeros((T + 1, rec))
    acc = np.zeros((T + 1, nv))
    emu.emu_phase_linearize(e, ob.P(tape), ob.P(acc))
    n, m = ndx, nu
    nm = n + m

    def blocks(r):
        A = r[:n * nm].reshape(n, nm)
        HX = r[n * nm:2 * n * nm].reshape(n, nm)
        o2 = 2 * n * nm
        return {"Fx": A[:, :n], "Fu": A[:, n:], "Lxx": HX[:, :n], "Lxu": HX[:, n:], "Luu": r[o2:o2 + m * m].reshape(m, m),
                "Lx": r[o2 + m * m:o2 + m * m + n], "Lu": r[o2 + m * m + n:o2 + m * m + n + m],
                "gap": r[o2 + m * m + n + m:o2 + m * m + 2 * n + m], "cost": r[o2 + m * m + 2 * n + m:o2 + m * m + 2 * n + m + 1]}
    for t in range(T + 1):
        ref = o.phase_tape(t)
        ref["gap"] = fs[t]
        ref["cost"] = np.array([ref["cost"]])
        got = blocks(tape[t])
        for key in got:
            if t == T and key in ("Fx", "Fu", "Lxu", "Luu", "Lu"):
                continue
            assert rel(np.asarray(got[key]).ravel(), np.asarray(ref[key]).ravel()) < tape_tol, (t, key)
    ok, Ko, ko, Vxo, _, dgo = o.phase_backward(1e-9)
    K = np.zeros((T, m, n))
    k = np.zeros((T, m))
    Vx = np.zeros((T + 1, n))
    dg = np.zeros(2)
    oke = np.zeros(1, dtype=np.int32)
    fe = np.zeros(1, dtype=np.int32)
    ce = np.zeros(1)
    emu.emu_phase_backward(e, ob.P(K), ob.P(k), ob.P(Vx), ob.P(dg), oke.ctypes.data_as(_ip), fe.ctypes.data_as(_ip), ob.P(ce))
    assert ok and oke[0] == 1
    # the RK4 cost Hessians of this candidate reach 1e9: the LLT of Quu at xreg = 1e-9 amplifies the 1e-10 tape differences
    assert rel(K, Ko) < 1e-3 and rel(k, ko) < 1e-3 and rel(Vx, Vxo) < 1e-5
    for ai in (2, 4):
        oko, xo, uo, co, d01 = o.phase_forward(2.0 ** -ai)
        xt = np.zeros((T + 1, nx))
        ut = np.zeros((T, nu))
        ct = np.zeros(1)
        dv = np.zeros(1)
        okr = np.zeros(1, dtype=np.int32)
        emu.emu_phase_rollout(e, ai, ob.P(xt), ob.P(ut), ob.P(ct), ob.P(dv), okr.ctypes.data_as(_ip))
        assert bool(okr[0]) == oko
        if oko and abs(co) < 1e12:
            assert rel(xt, xo) < 1e-5 and rel(ut, uo) < 1e-4 and abs(ct[0] - co) < 1e-5 * (1 + abs(co))
    if name == "eagle_catch":
        emu.emu_destroy(e)
        return
    emu.emu_set_warmstart(e, None, None)
    emu.emu_solve_c(e, 100, 0)
    xs_e = np.zeros((T + 1, nx))
    us_e = np.zeros((T, nu))
    ul = np.zeros((T, nu))
    it = np.zeros(1, dtype=np.int32)
    st = np.zeros(1, dtype=np.int32)
    emu.emu_get(e, ob.P(xs_e), ob.P(us_e), ob.P(ul), ob.P(ce), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    o.solve(None, None, 100)
    r = o.result()
    assert it[0] == r["iter"] and st[0] == r["status"]
    # north-star bound on the states.  One arm-joint torque near the end of the horizon is a flat direction of this problem
    # (states equal to 2e-9 while that control differs by 1.4e-4 after 20 iterations; the matrix-core sums of the RK4 assembly
    # are ordered differently from the oracle's): the decisive statement for RK4 nodes is the step-wise one, every iteration
    # reproduced from the other side's iterate and the same minimiser to 5e-8 on us (tests/test_gpu_teacher_forced.py::test_rk4_nodes)
    assert np.abs(xs_e - r["xs"]).max() < 1e-6
    # controls: the north-star 1e-4 everywhere except on that flat direction -- the controls of at most ONE knot may exceed it
    # (measured: knot 75, four rotor commands, 1.5e-4), and stay within 3e-4; everywhere else 1e-4
    du = np.abs(us_e - r["us"])
    knots = np.unique(np.argwhere(du >= 1e-4)[:, 0])
    print("RK4 solve: max |us - oracle|", du.max(), "entries beyond 1e-4:", int((du >= 1e-4).sum()), "at knots", knots)
    assert len(knots) <= 1 and du.max() < 3e-4, (knots, du.max())
    emu.emu_destroy(e)


@pytest.mark.parametrize("solver_type", [1, 2])
@pytest.mark.parametrize("name,dt,warm", [("hover", 40, False), ("displacement", 80, False), ("displacement", 80, True)])
def test_box_solvers_vs_oracle(empc, emu, name, dt, warm, solver_type):
    """crocoddyl SolverBoxFDDP (1) / SolverBoxDDP (2) through the device code on the CPU lane emulator -- BoxQP gains with
    clamped Qu in the backward pass, clamped trial controls in the rollout, the upstream solve loop in select -- against
    the oracle: iteration count, status, controls inside their limits, same trajectory.  `warm`: from the SbFDDP solution
    (feasible start: the BoxQP path from the first iteration, also for BoxFDDP)."""
    from conftest import CONFIGS
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = tr.createProblem(dt, False, "IntegratedActionModelEuler")  # useSquash = False (examples/python/trajectory.py:11,20-23)
    d = problem.desc
    prm = ob.default_params()
    prm.solver_type = solver_type
    maxiter = 30
    xs0 = us0 = None
    if warm:
        sq = tr.createProblem(dt, True, "IntegratedActionModelEuler")
        o0 = ob.OracleSolver(sq.desc)
        o0.solve(None, None, 100)
        r0 = o0.result()
        xs0, us0 = r0["xs"], np.ascontiguousarray(r0["us_squash"])  # squashed controls: inside the limits
    emu.emu_set_linearize_version(2)
    emu.emu_set_backward_version(4)
    emu.emu_set_rollout_version(6)
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    assert e.value
    emu.emu_set_warmstart(e, None if xs0 is None else ob.P(xs0), None if us0 is None else ob.P(us0))
    emu.emu_solve_c(e, maxiter, 1 if warm else 0)
    T, nx, nu = d.T, d.nx, d.nu
    xs_e, us_e, ul, ce = np.zeros((T + 1, nx)), np.zeros((T, nu)), np.zeros((T, nu)), np.zeros(1)
    it, st = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    emu.emu_get(e, ob.P(xs_e), ob.P(us_e), ob.P(ul), ob.P(ce), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    o = ob.OracleSolver(d, prm)
    o.solve(xs0, us0, maxiter, is_feasible=warm)
    r = o.result()
    assert it[0] == r["iter"] and st[0] == r["status"], (it, r["iter"], st, r["status"])
    lb = np.array([d.u_lb[i] for i in range(nu)])
    ub = np.array([d.u_ub[i] for i in range(nu)])
    assert (us_e >= lb - 1e-12).all() and (us_e <= ub + 1e-12).all()
    if abs(r["cost"]) < 1e6:
        assert np.abs(xs_e - r["xs"]).max() < 1e-5 and np.abs(us_e - r["us"]).max() < 1e-4
        assert abs(ce[0] - r["cost"]) < 1e-7 * (1 + abs(r["cost"]))
    emu.emu_destroy(e)
