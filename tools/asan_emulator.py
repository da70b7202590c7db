#!/usr/bin/env python3
"""Address/UB sanitizer pass over the HIP kernel bodies, executed by the CPU lane emulator (GPU ASan is not available on
the pool).  Builds tests/csrc/lane_emulator.cpp with -fsanitize=address,undefined and runs short solves of all four model
shapes through both forms of the backward and rollout bodies and the lean / full linearize bodies.

    LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
        python tools/asan_emulator.py [MACRO[=value] ...]
Macros select a build-time variant of the kernel bodies (empc_variants.hpp), e.g. EMPC_BWD_R4B=1 EMPC_BOXQP_ONE_EXIT=1: the
16-byte record moves of that variant read rows that end inside the next record (tape slack), which is what ASan checks here.
"""
import subprocess
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
MACROS = [a for a in sys.argv[1:] if a.startswith("EMPC_")]
ONLY_PAIR = "pair" in sys.argv[1:]  # only the two-contact (CT_PAIR3) section at the end
LIB = '/tmp/liblane_emulator_asan%s.so' % "".join("_" + m.replace("=", "") for m in MACROS)
subprocess.check_call(['g++', '-O1', '-g', '-std=c++20', '-pthread', '-fPIC', '-shared', '-fsanitize=address,undefined', '-fno-omit-frame-pointer'] + ["-D" + m for m in MACROS] + [
                       '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'csrc', 'lane_emulator.cpp'), '-o', LIB])
print("sanitizer build with", MACROS or "the default switches", flush=True)
import empc_loader, oracle_binding as ob
empc = empc_loader.load()
L = C.CDLL(LIB)
L.emu_create.restype = C.c_void_p
L.emu_create.argtypes = [C.POINTER(empc.T.ProblemDesc), C.POINTER(empc.T.SolverParams), C.c_int]
L.emu_destroy.argtypes=[C.c_void_p]; L.emu_set_x0.argtypes=[C.c_void_p, C.POINTER(C.c_double)]
L.emu_set_warmstart.argtypes=[C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.emu_solve_c.argtypes=[C.c_void_p, C.c_int, C.c_int]
for rel, dt, it in () if ONLY_PAIR else (("hexacopter370/trajectories/hover.yaml",40,100),("hexacopter370_flying_arm_3/trajectories/displacement.yaml",80,3),("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml",32,3),("hextilt_flying_arm_5/trajectories/push_slide.yaml",13,2)):
    t = empc.Trajectory(); t.autoSetup(empc.yaml_path(rel)); p = t.createProblem(dt, True, "IntegratedActionModelEuler")
    prm = ob.default_params()
    for bwd, roll in ((4, 6), (3, 5), (2, 1)):  # the shipped forms first, then the cross-check forms
        L.emu_set_backward_version(bwd); L.emu_set_rollout_version(roll); L.emu_set_linearize_version(2)
        e = C.c_void_p(L.emu_create(C.byref(p.desc), C.byref(prm), 2))
        L.emu_set_warmstart(e, None, None)
        L.emu_solve_c(e, it, 0)
        L.emu_destroy(e)
    print("ok", rel, flush=True)
# option branches of the shipped forms: box solvers (box-QP gains, clamped rollout), RK4 nodes, the 6D contact with gains
import pathlib, tempfile
from conftest import contact_variant
L.emu_set_backward_version(4); L.emu_set_rollout_version(6); L.emu_set_linearize_version(2)
t = empc.Trajectory(); t.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
for st in () if ONLY_PAIR else (1, 2):
    p = t.createProblem(80, False, "IntegratedActionModelEuler")
    prm = ob.default_params(); prm.solver_type = st
    e = C.c_void_p(L.emu_create(C.byref(p.desc), C.byref(prm), 2)); L.emu_set_warmstart(e, None, None); L.emu_solve_c(e, 4, 0); L.emu_destroy(e)
    print("ok box solver", st, flush=True)
prm = ob.default_params()
if not ONLY_PAIR:
    p = t.createProblem(80, True, "IntegratedActionModelRK4")
    e = C.c_void_p(L.emu_create(C.byref(p.desc), C.byref(prm), 1)); L.emu_set_warmstart(e, None, None); L.emu_solve_c(e, 2, 0); L.emu_destroy(e)
    print("ok rk4", flush=True)
    with tempfile.TemporaryDirectory() as td:
        _, p = contact_variant(empc, pathlib.Path(td), "ContactModel6D", (11.0, 5.0))
        e = C.c_void_p(L.emu_create(C.byref(p.desc), C.byref(prm), 1)); L.emu_set_warmstart(e, None, None); L.emu_solve_c(e, 2, 0); L.emu_destroy(e)
        print("ok 6D contact", flush=True)
# two ContactModel3D per stage (CT_PAIR3): the six-row linearize body, the pair paths of the role-split and per-lane rollouts, RK4
# stages, both arm classes (the second capture slot behind the rollout's LDS block and the second force behind the linearize unit are
# the new memory here)
from conftest import arm5_two_contact_variant, two_contact_variant
with tempfile.TemporaryDirectory() as td:
    for name, make in (("9-dof Euler", lambda: two_contact_variant(empc, pathlib.Path(td), "ContactModel3D", (3.0, 1.5), (2.0, 0.7), cone_on_second=True)),
                       ("9-dof RK4", lambda: two_contact_variant(empc, pathlib.Path(td), "ContactModel3D", integrator="IntegratedActionModelRK4")),
                       ("11-dof Euler", lambda: arm5_two_contact_variant(empc, pathlib.Path(td), (2.0, 1.0), (0.0, 3.0)))):
        _, p = make()
        for roll in (6, 1):
            L.emu_set_rollout_version(roll)
            e = C.c_void_p(L.emu_create(C.byref(p.desc), C.byref(prm), 2)); L.emu_set_warmstart(e, None, None); L.emu_solve_c(e, 2, 0); L.emu_destroy(e)
        print("ok two contacts,", name, flush=True)
