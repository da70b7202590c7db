#!/usr/bin/env python3
"""GPU box: run-to-run and lane-to-lane determinism of the solve (identical inputs must give bitwise identical outputs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import empc_loader
empc = empc_loader.load()
for rel, dt in (("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32), ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
                ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13)):
    t = empc.Trajectory(); t.autoSetup(empc.yaml_path(rel)); p = t.createProblem(dt, True, "IntegratedActionModelEuler")
    B = 8
    x0s = np.tile(p.x0, (B, 1))
    outs = []
    for rep in range(2):
        s = empc.SolverSbFDDP(p, batch=B)
        s.solve([], [], 100, x0s=x0s)
        outs.append((s.xs_batch.copy(), s.us_batch.copy(), s.iter_batch.copy(), s.cost_batch.copy()))
    xs, us, it, c = outs[0]
    same_lane = all(np.array_equal(xs[0], xs[b]) and np.array_equal(us[0], us[b]) for b in range(B))
    same_run = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    print(rel.split('/')[-1], "iters", it, "identical across the batch:", same_lane, "| identical across runs:", same_run)
