"""Streamed solves on the GPU (empc_solver_stream_begin / _run / _results): a queue of initial states pushed through fewer
slots than jobs gives, row by row, bitwise the result of plain batched solves of the same initial states.  Reference: one
SolverSbFDDP::solve([], [], maxiter) per initial state (src/sbfddp.cpp:192-226)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,slots,jobs", [("eagle_catch", 64, 200), ("displacement", 32, 96), ("hover", 16, 40),
                                             ("eagle_catch", 1024, 2048)])  # the last: the bench line's slot count
def test_stream_rows_equal_plain_solves(empc, problems, name, slots, jobs):
    _, problem = problems[name]
    d = problem.desc
    x0s = empc.perturbed_x0s(problem.x0, jobs, nq=d.model.nq, amplitude=0.05 if name != "hover" else 0.01)
    plain = empc.SolverSbFDDP(problem, batch=jobs)
    plain.solve([], [], 100, x0s=x0s)
    s = empc.SolverSbFDDP(problem, batch=slots)
    r = s.solve_stream(x0s, 100)
    # (bitwise, NaNs of rollouts that blow up included: equal_nan)
    assert np.array_equal(r["iter"], plain.iter_batch) and np.array_equal(r["status"], plain.status_batch)
    assert np.array_equal(r["xs"], plain.xs_batch, equal_nan=True) and np.array_equal(r["us"], plain.us_batch, equal_nan=True)
    assert np.array_equal(r["us_squash"], plain.us_squash_batch, equal_nan=True)
    assert np.array_equal(r["cost"], plain.cost_batch, equal_nan=True)
    st = s.stats()
    assert st["total_iters"] == int((plain.iter_batch + 1).sum())
    # the queue keeps the slots busy: fewer sweeps than the jobs solved batch after batch of `slots`
    batches = [plain.iter_batch[i:i + slots] for i in range(0, jobs, slots)]
    assert st["sweeps"] <= sum(int(b.max()) + 1 for b in batches) + len(batches)
    # a second stream on the same solver (other queue length) works and a plain solve afterwards is unaffected
    r2 = s.solve_stream(x0s[:slots // 2], 100)
    assert np.array_equal(r2["xs"], plain.xs_batch[:slots // 2], equal_nan=True)
    s.solve([], [], 100, x0s=x0s[:slots])
    assert np.array_equal(s.xs_batch, plain.xs_batch[:slots], equal_nan=True)


def test_stream_argument_errors(empc, problems):
    _, problem = problems["hover"]
    s = empc.SolverSbFDDP(problem, batch=2)
    with pytest.raises(empc.EmpcError, match="stream_begin"):
        s.stream_run(10)
    # an enabled iteration trace (what setCallbacks switches on) is no obstacle: it records plain solves and is off for the
    # duration of a stream, back on afterwards
    s.enable_trace(8)
    r = s.solve_stream(np.tile(problem.x0, (3, 1)), 10)
    plain = empc.SolverSbFDDP(problem, batch=1)
    plain.solve([], [], 10)
    assert np.array_equal(r["xs"][2], plain.xs_batch[0]) and r["iter"][0] == plain.iter_batch[0]
    s.solve([], [], 10)
    assert len(s.trace(0)) == min(8, s.iter_batch[0] + 1)


@pytest.mark.parametrize("variant", ["SolverBoxFDDP", "SolverBoxDDP", "RK4"])
def test_stream_of_option_variants(empc, variant):
    """The queue form with the box solvers (the QP warm start k_ of a refilled slot is a fresh solver's: zero) and with RK4
    nodes: rows bitwise those of plain batched solves."""
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
    if variant == "RK4":
        problem, cls, maxiter = tr.createProblem(80, True, "IntegratedActionModelRK4"), empc.SolverSbFDDP, 100
    else:
        problem, cls, maxiter = tr.createProblem(80, False, "IntegratedActionModelEuler"), getattr(empc, variant), 30
    d = problem.desc
    jobs, slots = 24, 8
    x0s = empc.perturbed_x0s(problem.x0, jobs, nq=d.model.nq)
    plain = cls(problem, batch=jobs)
    plain.solve([], [], maxiter, x0s=x0s)
    s = cls(problem, batch=slots)
    r = s.solve_stream(x0s, maxiter)
    assert np.array_equal(r["iter"], plain.iter_batch) and np.array_equal(r["status"], plain.status_batch)
    assert np.array_equal(r["xs"], plain.xs_batch, equal_nan=True) and np.array_equal(r["us"], plain.us_batch, equal_nan=True)
    assert np.array_equal(r["us_squash"], plain.us_squash_batch, equal_nan=True)
