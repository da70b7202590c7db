#!/usr/bin/env python3
"""Counterpart of the reference's examples/python/trajectory.py:11-26 on the MI355X solver.

    trajectory.autoSetup(yaml) -> createProblem(dt, squash, integrator) -> SolverSbFDDP(problem).solve([], [], maxiter)

plus the batched extension: the same problem from many perturbed initial states in one call.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import empc_loader  # noqa: E402

empc = empc_loader.load()

dt = 80  # ms
trajectory = empc.Trajectory()
trajectory.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
problem = trajectory.createProblem(dt, True, "IntegratedActionModelEuler")

solver = empc.SolverSbFDDP(problem, trajectory.squash)   # the reference call (examples/python/trajectory.py:21); batch = 1
solver.setCallbacks([empc.CallbackVerbose()])
solver.solve([], [], maxiter=100)
print("iterations", solver.iter, "cost %.6f" % solver.cost, "final position", np.round(solver.xs[-1][:3], 4))
print("first squashed control", np.round(solver.us_squash[0], 3))

batch = 256                                    # the data-parallel extension
x0s = empc.perturbed_x0s(problem.x0, batch, nq=problem.desc.model.nq)
bsolver = empc.SolverSbFDDP(problem, batch=batch)
bsolver.solve([], [], maxiter=100, x0s=x0s)
print("batch of", batch, ": iterations", np.bincount(bsolver.iter_batch)[-1:], "cost range %.4f .. %.4f" %
      (bsolver.cost_batch.min(), bsolver.cost_batch.max()), "| device ms %.1f" % bsolver.stats()["ms_total"])
