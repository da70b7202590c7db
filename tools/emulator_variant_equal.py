#!/usr/bin/env python3
"""CPU only: is a build-time variant of the kernel bodies (a -D macro) bit-identical to the default on the lane emulator?
Builds tests/csrc/lane_emulator.cpp twice (with and without the macro), runs linearize -> backward -> four step lengths of the
rollout from a seeded random candidate with open gaps on the four BASELINE workloads (Euler nodes; RK4 nodes on the arm-3 files)
and compares every trial state, control, cost and expected-improvement term bit for bit.

    python3 tools/emulator_variant_equal.py EMPC_ROLL_GAP_EARLY [MORE_MACROS ...]      (a macro may carry a value: EMPC_BWD_R4B=1)
The tape, the gains, Vx, the expected-improvement sums and the status of the backward pass are compared as well.
Then whole emulated solves with the box solvers (SolverBoxFDDP / SolverBoxDDP: the box-QP gains of the backward pass) on hover and
displacement, cold and warm: iterations, status, states, controls, cost bit for bit."""
import os
import subprocess
import sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import empc_loader; empc = empc_loader.load()
import oracle_binding as ob
import test_emulator_parity as tep
from conftest import CONFIGS
_ip = C.POINTER(C.c_int)
def load(path):
    tep.EMU = path
    return tep.emu.__wrapped__(empc)
SCALE = 1.0  # size of the random candidate (tests/test_emulator_parity.py candidate): 1.0 = far from hover, 0.15 = near hover; both are run
def run(emu, problem, name):
    d = problem.desc; prm = ob.default_params()
    emu.emu_set_linearize_version(2); emu.emu_set_backward_version(4); emu.emu_set_rollout_version(6)
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    T, nx, nu, nv = d.T, d.nx, d.nu, d.model.nv
    xs, us = tep.candidate(d, 3, scale=SCALE)
    emu.emu_set_warmstart(e, ob.P(xs), ob.P(us)); emu.emu_phase_setup(e, 0.1, 0, 1e-9, 0)
    tape = np.zeros((T+1, emu.emu_rec(e))); acc = np.zeros((T+1, nv))
    emu.emu_phase_linearize(e, ob.P(tape), ob.P(acc))
    K=np.zeros((T,nu,d.ndx)); k=np.zeros((T,nu)); Vx=np.zeros((T+1,d.ndx)); dg=np.zeros(2); ok=np.zeros(1,dtype=np.int32); fe=np.zeros(1,dtype=np.int32); ce=np.zeros(1)
    emu.emu_phase_backward(e, ob.P(K), ob.P(k), ob.P(Vx), ob.P(dg), ok.ctypes.data_as(_ip), fe.ctypes.data_as(_ip), ob.P(ce))
    outs=[(tape.copy(), acc.copy(), K.copy(), k.copy(), Vx.copy(), dg.copy(), ok.copy(), fe.copy(), ce.copy())]
    for ai in (1,2,4,6,8,9):
        xt=np.zeros((T+1,nx)); ut=np.zeros((T,nu)); ct=np.zeros(1); dv=np.zeros(1); okr=np.zeros(1,dtype=np.int32)
        emu.emu_phase_rollout(e, ai, ob.P(xt), ob.P(ut), ob.P(ct), ob.P(dv), okr.ctypes.data_as(_ip))
        outs.append((xt.copy(), ut.copy(), ct.copy(), dv.copy(), okr.copy()))
    emu.emu_destroy(e)
    return outs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
macros = sys.argv[1:]
macro = "_".join(m.replace("=", "") for m in macros)
for out, flags in (("/tmp/emu_variant_base.so", []), ("/tmp/emu_variant_%s.so" % macro, ["-D" + m for m in macros])):
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include")] + flags +
                          [os.path.join(ROOT, "tests", "csrc", "lane_emulator.cpp"), "-o", out])
a = load("/tmp/emu_variant_base.so"); b = load("/tmp/emu_variant_%s.so" % macro)
all_same = True
for name in ("displacement","eagle_catch","push_slide","hover"):
    tr = empc.Trajectory(); tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    for integ in ("IntegratedActionModelEuler","IntegratedActionModelRK4"):
        if integ.endswith("RK4") and name in ("push_slide","hover"): continue
        problem = tr.createProblem(CONFIGS[name][1], True, integ)
        for SCALE in (1.0, 0.15):
            ra, rb = run(a, problem, name), run(b, problem, name)
            same = all(all(np.array_equal(x, y, equal_nan=True) for x, y in zip(p, q)) for p, q in zip(ra, rb))
            all_same = all_same and same
            # (a trial with a long step from a random candidate overflows part of the way: its states up to there are compared as
            #  numbers, the rest as NaN == NaN)
            print(name, integ, "candidate scale", SCALE, "bitwise equal:", same, "| trial rollouts finite to the end: %d of %d; finite trial "
                  "states compared: %d of %d" % (sum(int(np.isfinite(p[2]).all() and np.isfinite(p[0]).all()) for p in ra[1:]), len(ra) - 1,
                                                sum(int(np.isfinite(p[0]).all(axis=1).sum()) for p in ra[1:]), sum(len(p[0]) for p in ra[1:])), flush=True)

def box_solve(emu, tr, dt, warm, solver_type, maxiter=30):
    problem = tr.createProblem(dt, False, "IntegratedActionModelEuler")
    d = problem.desc; prm = ob.default_params(); prm.solver_type = solver_type
    xs0 = us0 = None
    if warm:
        sq = tr.createProblem(dt, True, "IntegratedActionModelEuler")
        o0 = ob.OracleSolver(sq.desc); o0.solve(None, None, 100); r0 = o0.result()
        xs0, us0 = r0["xs"], np.ascontiguousarray(r0["us_squash"])
    emu.emu_set_linearize_version(2); emu.emu_set_backward_version(4); emu.emu_set_rollout_version(6)
    e = C.c_void_p(emu.emu_create(C.byref(d), C.byref(prm), 1))
    emu.emu_set_warmstart(e, None if xs0 is None else ob.P(xs0), None if us0 is None else ob.P(us0))
    emu.emu_solve_c(e, maxiter, 1 if warm else 0)
    T, nx, nu = d.T, d.nx, d.nu
    xs_e, us_e, ul, ce = np.zeros((T + 1, nx)), np.zeros((T, nu)), np.zeros((T, nu)), np.zeros(1)
    it, st = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    emu.emu_get(e, ob.P(xs_e), ob.P(us_e), ob.P(ul), ob.P(ce), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip))
    emu.emu_destroy(e)
    return xs_e, us_e, ce, it, st
for name, dt, warm in (("hover", 40, False), ("displacement", 80, False), ("displacement", 80, True)):
    tr = empc.Trajectory(); tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    for solver_type in (1, 2):
        ra, rb = box_solve(a, tr, dt, warm, solver_type), box_solve(b, tr, dt, warm, solver_type)
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ra, rb))
        all_same = all_same and same
        print("box solve", name, "warm" if warm else "cold", "solver_type", solver_type, "bitwise equal:", same, "| iterations", int(ra[3][0]), "status", int(ra[4][0]))

sys.exit(0 if all_same else 1)
