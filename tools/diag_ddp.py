#!/usr/bin/env python3
"""GPU box diagnostic: short-maxiter solves that go through the DDP clean-up, GPU vs oracle, with iteration traces."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
empc = ob.empc
np.set_printoptions(linewidth=200, precision=6)
CFG = {"displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
       "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
       "hover": ("hexacopter370/trajectories/hover.yaml", 40)}
for name, maxiter in [("displacement", 1), ("displacement", 2), ("eagle_catch", 2), ("hover", 3)]:
    rel, dt = CFG[name]
    t = empc.Trajectory(); t.autoSetup(empc.yaml_path(rel)); p = t.createProblem(dt, True, "IntegratedActionModelEuler"); d = p.desc
    B = 4
    x0s = empc.perturbed_x0s(p.x0, B, nq=d.model.nq)
    s = empc.SolverSbFDDP(p, batch=B); s.enable_trace(64)
    s.solve([], [], maxiter, x0s=x0s)
    r = ob.solve_batch(d, x0s, maxiter, nthreads=4)
    print("==", name, maxiter, "gpu iter", s.iter_batch, "status", s.status_batch, "| oracle iter", r["iter"], "status", r["status"])
    print("   cost gpu", s.cost_batch, "oracle", r["cost"])
    print("   max|dxs|", np.abs(s.xs_batch - r["xs"]).reshape(B, -1).max(axis=1), "max|dus|", np.abs(s.us_batch - r["us"]).reshape(B, -1).max(axis=1),
          "max|dusq|", np.abs(s.us_squash_batch - r["us_squash"]).reshape(B, -1).max(axis=1))
    for b in (0, 1):
        o = ob.OracleSolver(d); o.set_x0(x0s[b]); o.solve(None, None, maxiter)
        print("   trace gpu b=%d (phase iter cost stop xreg alpha feas dV dVexp gap d0 d1)" % b); print(s.trace(b))
        print("   trace oracle"); print(o.trace())
