"""Contact dynamics on the second arm class -- hextilt_flying_arm_5: 6 bodies, 6 tilted rotors, 11 velocity dimensions, the
robot of BASELINE configs[3] -- with ContactModel3D and ContactModel6D (src/factory/contacts.cpp:26-79 builds a contact for any
robot).  No shipped YAML has a contact on this robot: the problem is push_slide with a contact
stage appended (conftest.arm5_contact_variant).  Kernel instantiations: empc_inst_6_6_contact.hip / _contact6.hip."""
import numpy as np
import pytest

import oracle_binding as ob
import parity_criteria as pc
import stepwise as sw
from conftest import arm5_contact_variant
from test_gpu_parity import phase_parity
from test_gpu_teacher_forced import check, factory, save

pytestmark = pytest.mark.gpu

VARIANTS = [("ContactModel3D", (0.0, 0.0)), ("ContactModel3D", (7.0, 3.0)), ("ContactModel6D", (0.0, 0.0)), ("ContactModel6D", (6.0, 2.0))]


@pytest.mark.parametrize("contact,gains", VARIANTS)
def test_arm5_contact_phase_parity(empc, tmp_path, contact, gains):
    """linearize / backward / rollout kernels against the oracle's calcDiff / backwardPass / forwardPass."""
    _, problem = arm5_contact_variant(empc, tmp_path, contact, gains)
    assert empc.solver_supported(problem), empc.last_error()
    assert problem.desc.has_contact and problem.desc.model.nbodies == 6
    phase_parity(empc, problem, "arm5/" + contact)


@pytest.mark.parametrize("contact,gains", [VARIANTS[0], VARIANTS[3]])
def test_arm5_contact_stepwise(empc, tmp_path, contact, gains):
    """Every iteration of the oracle's paths reproduced by the device, every iteration of the device's free-running paths by
    the oracle, same minimiser (tests/stepwise.py)."""
    _, problem = arm5_contact_variant(empc, tmp_path, contact, gains)
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 4, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, maxiter=60, tape_every=23)
    save("arm5_contact_%s_%g" % (contact, gains[0]), rep)
    check(rep, max_exploded=5)  # (measured r04 on hardware: 0 exploded)


@pytest.mark.parametrize("contact,gains", [VARIANTS[0], VARIANTS[2]])
def test_arm5_contact_solve(empc, tmp_path, contact, gains):
    """Free-running solves from the file's state and perturbed ones: a rollout the oracle solves in a few dozen iterations must
    match it (iterations, 1e-4 on xs / us); every rollout the GPU reports as solved is a solution of the same problem."""
    _, problem = arm5_contact_variant(empc, tmp_path, contact, gains)
    d = problem.desc
    B = 4
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    s = empc.SolverSbFDDP(problem, batch=B)
    s.solve([], [], 100, x0s=x0s)
    assert np.isfinite(s.xs_batch).all()
    prm = empc.default_params()
    r = ob.solve_batch(d, x0s, 100, nthreads=4)
    full = 0
    for b in range(B):
        same = (s.iter_batch[b] == r["iter"][b] and s.status_batch[b] == r["status"][b]
                and np.abs(s.xs_batch[b] - r["xs"][b]).max() < 1e-4 and np.abs(s.us_batch[b] - r["us"][b]).max() < 1e-4)
        full += int(same)
        if pc.solved(s.status_batch[b:b + 1], s.cost_batch[b:b + 1])[0]:
            o2 = ob.OracleSolver(d)
            o2.set_x0(x0s[b])
            o2.set_smooth(prm.smooth_init * prm.smooth_mult)
            c, fs, _ = o2.phase_calcdiff(s.xs_batch[b], s.us_batch[b])
            assert abs(c - s.cost_batch[b]) < 1e-8 * (1 + abs(c)) and np.abs(fs).max() < 1e-7
    print("complete agreement on %d of %d rollouts" % (full, B), s.iter_batch, r["iter"])
    assert full >= 1
