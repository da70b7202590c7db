"""Streamed solves ("continuous batching", empc_solver_stream_*) on the CPU lane emulator: a queue of initial states pushed
through fewer slots than jobs gives, row by row, bitwise the result of plain solves of the same initial states -- the
hand-over inside select (result row out, next job in, fresh solver scalars) changes nothing a trajectory can see.
Reference: one SolverSbFDDP::solve([], [], maxiter) per initial state (src/sbfddp.cpp:192-226, benchmark/utils/utils.hpp:15-27)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import stepwise as sw


@pytest.mark.parametrize("name,slots,jobs", [("eagle_catch", 3, 7), ("displacement", 2, 3), ("hover", 4, 2)])
def test_stream_rows_equal_plain_solves(empc, problems, name, slots, jobs):
    emu = sw.load_emulator()
    _, problem = problems[name]
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, jobs, nq=d.model.nq, amplitude=0.05 if name != "hover" else 0.01)
    # plain: every job in its own slot
    h = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), jobs))
    emu.emu_set_x0(h, ob.P(x0s))
    emu.emu_set_warmstart(h, None, None)
    emu.emu_solve_c(h, 100, 0)
    xs = np.zeros((jobs, d.T + 1, d.nx))
    us = np.zeros((jobs, d.T, d.nu))
    ul = np.zeros((jobs, d.T, d.nu))
    cost = np.zeros(jobs)
    it = np.zeros(jobs, dtype=np.int32)
    st = np.zeros(jobs, dtype=np.int32)
    emu.emu_get(h, ob.P(xs), ob.P(us), ob.P(ul), ob.P(cost), it.ctypes.data_as(sw._ip), st.ctypes.data_as(sw._ip))
    emu.emu_destroy(h)
    # streamed through `slots` slots
    h2 = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), slots))
    row = emu.emu_stream_row_doubles(h2)
    rows = np.zeros((jobs, row))
    total = C.c_longlong()
    sweeps = emu.emu_stream_c(h2, jobs, ob.P(x0s), 100, ob.P(rows), C.byref(total))
    emu.emu_destroy(h2)
    nxs, nus = (d.T + 1) * d.nx, d.T * d.nu
    assert row == nxs + 2 * nus + 3
    assert np.array_equal(rows[:, :nxs].reshape(xs.shape), xs)
    assert np.array_equal(rows[:, nxs:nxs + nus].reshape(us.shape), us)
    assert np.array_equal(rows[:, nxs + 2 * nus], cost)
    assert np.array_equal(rows[:, nxs + 2 * nus + 1].astype(np.int32), it)
    assert np.array_equal(rows[:, nxs + 2 * nus + 2].astype(np.int32), st)
    assert total.value == int((it + 1).sum())
    # fewer sweeps than the jobs solved one batch after the other would need, never fewer than the longest job
    assert sweeps >= int(it.max()) + 1
    # us_squash of the row: sigma of the last control evaluated at every node, with the final smoothing
    o = ob.OracleSolver(d, prm)
    o.set_x0(x0s[0])
    o.solve(None, None, 100)
    r = o.result()
    if r["iter"] == it[0]:
        assert np.abs(rows[0, nxs + nus:nxs + 2 * nus].reshape(d.T, d.nu) - r["us_squash"]).max() < 1e-4
