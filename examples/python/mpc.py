#!/usr/bin/env python3
"""Counterpart of the reference's examples/python/mpc.py:30-62: Carrot MPC in closed loop with RK4 plants.

One controller, `batch` plants: the plants, the initial-state hand-over and the warm starts stay on the device.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import empc_loader  # noqa: E402

empc = empc_loader.load()

dt = 80
trajectory = empc.Trajectory()
trajectory.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
problem = trajectory.createProblem(dt, True, "IntegratedActionModelEuler")
solver = empc.SolverSbFDDP(problem)
solver.solve([], [], 100)
xs, us = np.array(solver.xs), np.array(solver.us)

batch = 64
mpc = empc.CarrotMpc(trajectory, xs, dt, empc.yaml_path("hexacopter370_flying_arm_3/mpc/mpc.yaml"), batch=batch)
mpc.updateProblem(0)
mpc.solver.plant_states = empc.perturbed_x0s(xs[0], batch, nq=problem.desc.model.nq, amplitude=0.02)
mpc.solver.solve(xs[:mpc.problem.T + 1], us[:mpc.problem.T], 100, x0s="plant")
mpc.solver.convergence_init = 1e-3

dt_simulator, t = 2, 0
for i in range(200):
    mpc.updateProblem(t)
    mpc.solver.solve("previous", "previous", mpc.iters, x0s="plant")
    mpc.solver.plant_step(dt_simulator)        # control = each plant's us_squash[0]
    t += dt_simulator
states = mpc.solver.plant_states
ref = mpc.computeStateReference(t)
print("t = %d ms: plant positions, mean %s, reference %s" % (t, np.round(states[:, :3].mean(axis=0), 4), np.round(ref[:3], 4)))
print("tracking error (position, max over plants): %.4f m" % np.abs(states[:, :3] - ref[:3]).max())
