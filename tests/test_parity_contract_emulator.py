"""The parity contract of the north star (tests/parity_criteria.py north_star_contract) exercised on the CPU:
the kernel bodies on the lane emulator play the device.  Same three assertions as `__graft_entry__.smoke()` and bench.py's
`parity.contract` make on the GPU: same minimiser from a common restart <= 1e-4, plain-solve cost within 1e-5 relative, identical
iteration count; the plain-solve distance in xs / us is reported against a 2e-4 tripwire.
Reference: SolverSbFDDP::solve, src/sbfddp.cpp:192-226 (the plain solve), stop rule src/sbfddp.cpp:271-276."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity_criteria as pc
import stepwise as sw

_ip = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def emu():
    return sw.load_emulator()


def emulated_plain_solve(emu, d, prm, x0, maxiter=100):
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    emu.emu_set_x0(e, ob.P(np.ascontiguousarray(x0)))
    emu.emu_set_warmstart(e, None, None)
    emu.emu_solve_c(e, maxiter, 0)
    xs, us, ul, ce = np.zeros((d.T + 1, d.nx)), np.zeros((d.T, d.nu)), np.zeros((d.T, d.nu)), np.zeros(1)
    it, st = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    emu.emu_get(e, ob.P(xs), ob.P(us), ob.P(ul), ob.P(ce), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    emu.emu_destroy(e)
    return xs, us, float(ce[0]), int(it[0])


@pytest.mark.parametrize("name", ["displacement", "eagle_catch"])
def test_contract_holds_for_the_kernel_bodies_on_the_emulator(empc, problems, emu, name):
    _, problem = problems[name]
    d = problem.desc
    prm = ob.default_params()
    xs, us, cost, it = emulated_plain_solve(emu, d, prm, problem.x0)
    c = pc.north_star_contract(empc, ob, sw, problem, xs, us, cost, it, backend=lambda n, p2: sw.EmuBackend(emu, d, p2, n))
    print(name, c)
    assert c["passed"], c["failures"]
    assert c["restart_xs_err"] < 1e-8  # (the restart is well conditioned: measured 5e-13)


def test_contract_reports_a_wrong_solution(empc, problems, emu):
    """the checker itself: a trajectory that is NOT the solve's result (the oracle's, shifted) fails all of the plain-solve parts"""
    _, problem = problems["displacement"]
    d = problem.desc
    ref = ob.solve_batch(d, np.array([problem.x0]), 100, nthreads=1)
    xs = ref["xs"][0].copy()
    xs[5:, 0] += 5e-4
    c = pc.north_star_contract(empc, ob, sw, problem, xs, ref["us"][0], float(ref["cost"][0]) * (1 + 1e-4), int(ref["iter"][0]) + 1,
                               backend=lambda n, p2: sw.EmuBackend(emu, d, p2, n))
    assert not c["passed"] and len(c["failures"]) >= 3, c
