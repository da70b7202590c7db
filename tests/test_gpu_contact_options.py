"""The contact factory's other options on the GPU (src/factory/contacts.cpp:26-79):
ContactModel6D -- six constraint rows, its own kernel instantiation -- and Baumgarte gains on either contact type.  No
shipped YAML uses them, so the problems are eagle_catch with its grasp-stage contact swapped (conftest.contact_variant)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity_criteria as pc
from conftest import contact_variant
from test_gpu_parity import phase_parity

pytestmark = pytest.mark.gpu

VARIANTS = [("ContactModel3D", (9.0, 4.0)), ("ContactModel6D", (0.0, 0.0)), ("ContactModel6D", (11.0, 5.0))]


@pytest.mark.parametrize("contact,gains", VARIANTS)
def test_contact_options_phase_parity(empc, tmp_path, contact, gains):
    """linearize / backward / rollout kernels against the oracle's calcDiff / backwardPass / forwardPass."""
    _, problem = contact_variant(empc, tmp_path, contact, gains)
    assert empc.solver_supported(problem), empc.last_error()
    phase_parity(empc, problem, "eagle_catch/" + contact)


@pytest.mark.parametrize("contact,gains", VARIANTS)
def test_contact_options_solve(empc, tmp_path, contact, gains):
    """Full free-running solves: the unperturbed problem and three perturbed initial states.  A rollout either matches the
    oracle completely (iterations, status, 1e-4 on xs / us) or -- the contact problem's iteration path is rounding-sensitive,
    most of all with six constraint rows on this 9-dof arm -- it is covered by the step-wise parity of
    tests/test_gpu_teacher_forced.py::test_contact_options (every iteration from the other side's iterate, same minimiser).
    Asserted here on every rollout the GPU reports as solved: it is a solution of the same problem (the oracle's cost and
    gaps at the returned trajectory)."""
    _, problem = contact_variant(empc, tmp_path, contact, gains)
    d = problem.desc
    B = 4
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    s = empc.SolverSbFDDP(problem, batch=B)
    s.solve([], [], 100, x0s=x0s)
    prm = empc.default_params()
    full = 0  # rollouts that agree completely (informational: printed with -s)
    for b in range(B):
        o = ob.OracleSolver(d)
        o.set_x0(x0s[b])
        o.solve(None, None, 100)
        r = o.result()
        same = (s.iter_batch[b] == r["iter"] and s.status_batch[b] == r["status"]
                and np.abs(s.xs_batch[b] - r["xs"]).max() < 1e-4 and np.abs(s.us_batch[b] - r["us"]).max() < 1e-4)
        full += int(same)
        if pc.solved(s.status_batch[b:b + 1], s.cost_batch[b:b + 1])[0]:
            o2 = ob.OracleSolver(d)
            o2.set_x0(x0s[b])
            o2.set_smooth(prm.smooth_init * prm.smooth_mult)
            c, fs, _ = o2.phase_calcdiff(s.xs_batch[b], s.us_batch[b])
            assert abs(c - s.cost_batch[b]) < 1e-8 * (1 + abs(c)) and np.abs(fs).max() < 1e-7
    print("complete agreement on %d of %d rollouts" % (full, B))
