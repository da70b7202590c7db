"""saveLogfile / CallbackLogger: the on-disk layout of the reference's logging helper
(bindings/python/eagle_mpc/utils/tools.py:68-87), filled from the device iteration trace."""
import pickle

import numpy as np
import pytest

REFERENCE_KEYS = ["xs", "us", "us_squash", "fs", "steps", "iters", "costs", "muLM", "muV", "stops", "grads", "dt", "forces",
                  "frame_poses", "cogs"]


def test_save_logfile_layout(empc, tmp_path):
    log = empc.utils.CallbackLogger()
    log.xs, log.us = [np.zeros(3)], [np.ones(2)]
    log.steps, log.iters, log.costs = [1.0, 0.5], [0, 1], [3.0, 2.0]
    log.u_regs, log.x_regs, log.stops, log.grads, log.fs = [1e-9, 1e-9], [1e-9, 1e-9], [1.0, 0.1], [2.0, 1.0], [0.3, 0.0]
    f = tmp_path / "log.pkl"
    empc.utils.saveLogfile(str(f), log, 20, us_squash=[np.ones(2)])
    data = pickle.load(open(f, "rb"))
    assert list(data.keys()) == REFERENCE_KEYS  # same keys, same order as the reference's dictionary
    assert data["dt"] == 20 and data["muLM"] == log.u_regs and data["muV"] == log.x_regs and data["steps"] == [1.0, 0.5]
    assert empc.utils.loadLogfile(str(f))["costs"] == [3.0, 2.0]


@pytest.mark.gpu
def test_logger_from_device_trace(empc, problems, tmp_path):
    import oracle_binding as ob
    _, problem = problems["displacement"]
    s = empc.SolverSbFDDP(problem, batch=2)
    s.enable_trace(128)
    s.solve([], [], 100)
    log = empc.utils.CallbackLogger.from_solver(s, 1)
    o = ob.OracleSolver(problem.desc)
    o.solve(None, None, 100)
    tr = o.trace()
    assert log.iters == [int(v) for v in tr[:, 1]] and log.steps == [float(v) for v in tr[:, 5]]
    assert np.allclose(log.costs, tr[:, 2], rtol=1e-6) and np.allclose(log.grads, -tr[:, 11], rtol=1e-5, atol=1e-9)
    f = tmp_path / "gpu.pkl"
    empc.utils.saveLogfile(str(f), log, 80, us_squash=list(s.us_squash_batch[1]))
    data = empc.utils.loadLogfile(str(f))
    assert len(data["xs"]) == problem.T + 1 and len(data["us"]) == problem.T and len(data["us_squash"]) == problem.T
