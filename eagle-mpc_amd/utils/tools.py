"""Logging helpers with the reference's names and on-disk layout (bindings/python/eagle_mpc/utils/tools.py:68-87).

``CallbackLogger`` carries what crocoddyl's CallbackLogger collects through the solver's callback hook -- here it is filled
from the device iteration trace (``SolverSbFDDP.enable_trace`` / ``trace``) after a solve instead of being called once per
iteration.  ``saveLogfile`` writes the same pickle dictionary, key for key, as the reference's ``saveLogfile``.
"""
import pickle

import numpy as np


class CallbackLogger:
    """Fields of crocoddyl.CallbackLogger.  ``fs`` holds the gap NORM of every iteration (the device trace records the norm
    the stopping test uses, src/sbfddp.cpp:309, not the T + 1 gap vectors)."""

    def __init__(self):
        self.xs, self.us, self.fs = [], [], []
        self.steps, self.iters, self.costs = [], [], []
        self.u_regs, self.x_regs, self.stops, self.grads = [], [], [], []
        self.phases = []  # extension: 0, 1, .. = FDDP pass of the continuation, 100 = DDP clean-up

    @classmethod
    def from_solver(cls, solver, b=0):
        """Log of trajectory ``b`` of the last solve (the solver must have been traced: ``solver.enable_trace(n)``)."""
        log = cls()
        tr = solver.trace(b)
        log.xs = list(solver.xs_batch[b])
        log.us = list(solver.us_batch[b])
        log.phases = [int(v) for v in tr[:, 0]]
        log.iters = [int(v) for v in tr[:, 1]]
        log.costs = [float(v) for v in tr[:, 2]]
        log.stops = [float(v) for v in tr[:, 3]]
        log.x_regs = [float(v) for v in tr[:, 4]]
        log.u_regs = [float(v) for v in tr[:, 4]]  # ureg_ follows xreg_ (increase/decreaseRegularization)
        log.steps = [float(v) for v in tr[:, 5]]
        log.fs = [float(v) for v in tr[:, 9]]
        log.grads = [float(-v) for v in tr[:, 11]]  # CallbackLogger: -expectedImprovement()[1]
        return log


def saveLogfile(filename, log, dt, us_squash=[], forces=[], frame_poses=[], cogs=[]):
    """Same dictionary as the reference's saveLogfile (tools.py:68-87)."""
    data = {
        "xs": log.xs,
        "us": log.us,
        "us_squash": us_squash,
        "fs": log.fs,
        "steps": log.steps,
        "iters": log.iters,
        "costs": log.costs,
        "muLM": log.u_regs,
        "muV": log.x_regs,
        "stops": log.stops,
        "grads": log.grads,
        "dt": dt,
        "forces": forces,
        "frame_poses": frame_poses,
        "cogs": cogs,
    }
    with open(filename, "wb") as f:
        pickle.dump(data, f)


def loadLogfile(filename):
    with open(filename, "rb") as f:
        return pickle.load(f)
