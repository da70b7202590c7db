#!/usr/bin/env python3
"""Time the three hot kernels in isolation (phase-level C ABI) on a full batch: ms per launch from HIP events."""
import argparse, json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import empc_loader
empc = empc_loader.load()
from bench import CONFIGS, algorithmic_words

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="displacement")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
rel, dt = CONFIGS[a.config]
t = empc.Trajectory(); t.autoSetup(empc.yaml_path(rel)); p = t.createProblem(dt, True, "IntegratedActionModelEuler")
d = p.desc
B = a.batch
s = empc.SolverSbFDDP(p, batch=B)
x0s = empc.perturbed_x0s(p.x0, B, nq=d.model.nq)
# a realistic candidate: two iterations of the solver from the empty guess
s.solve([], [], 2, x0s=x0s)
xs, us = s.xs_batch, s.us_batch
res = {}
for name in ("linearize", "backward", "rollout"):
    ts = []
    for r in range(a.reps):
        if name == "linearize":
            s.linearize(xs, us, smooth=0.1, is_feasible=False, x0s=x0s, fetch=False)
        elif name == "backward":
            s.backward(xreg=1e-9, is_feasible=False)
        else:
            s.rollout(1.0, ddp=False, is_feasible=False)
        ts.append(s.stats()["ms_" + name])
    res[name] = float(np.median(ts))
w = algorithmic_words(d.nx, d.ndx, d.nu)
units = {"linearize": B * (d.T + 1), "backward": B * d.T, "rollout": B * 10 * (d.T + 1)}
out = {k: {"ms": v, "GBs": units[k] * w[k] * 8 / (v * 1e-3) / 1e9} for k, v in res.items()}
print(json.dumps(out))
import ctypes as C
cnt = (C.c_ulonglong * 128)()
empc.lib().empc_solver_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
empc.lib().empc_solver_debug_counters(s._h, cnt, 128)
print('stage cycles (EMPC_STAMPS builds; rollout v1, trajectory 0, alpha 1/2): feedback|prep|rnea|crba|chol|kkt|euler|costs|-|tail', list(cnt)[:10])


print('linearize stage cycles (unit b=0,t=10): S0 load|S1 squash/trig|S2 nominal+Euler|S3 tangent|S4 chol|S5 solves+Fx,Fu|S6 state costs (column sums)|ctrl costs|frame costs|S7 store|S6 staging|S6 nominal parts|S2 nominal chain only|S6 activations', list(cnt)[32:46])
print('backward stage cycles (trajectory 0, EMPC_STAMPS builds): load|W|Q|gains tail (sums)|Vxx tail (sync)|sym|gap tail (sums)|looptop', list(cnt)[16:24])
print('backward sub-stages: chol+solves|sync|Quu k+sync|Vx partial|Vxx MFMA + W write|gap dot products|any', list(cnt)[24:31])
print('rollout6 role cycles per rollout (EMPC_STAMPS builds; workgroup 0): role A|B|C|D x {phase I work, wait 1, phase II work, wait 2}',
      [list(cnt)[48 + 4 * r:52 + 4 * r] for r in range(4)])
print('rollout6 role B sub-stages (cycles per rollout): fetch issue|quat,trig,scan|rnea|H,CAP writes|frame costs|-|outside', list(cnt)[0:7])
print('rollout6 role A sub-stages: operand loads|state diff|dv|K dx|squash,tau,stores|-|outside', list(cnt)[64:71])
print('rollout6 role C sub-stages: II operands|II solve|II contact|II ACC,Euler,check|II gap,XT|I crba+chol|outside', list(cnt)[80:87])
print('rollout6 role D sub-stages: I finish_cost|I x,xs store|I state costs|II control costs,us store|-|-|outside', list(cnt)[88:95])
