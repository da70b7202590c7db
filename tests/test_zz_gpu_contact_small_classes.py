"""Contact dynamics on the robot classes the shipped contact files do not use -- (bodies, rotors) = (1, 6) hexacopter370,
(1, 4) iris, (3, 6) hexacopter680_flying_arm_2 -- with ContactModel3D and ContactModel6D (src/factory/contacts.cpp:26-79 builds
a contact for any robot).  The problems are hover files with a contact stage at the base link
appended (conftest.small_class_contact_variant).  Kernel instantiations: empc_inst_{1_6,1_4,3_6}_contact.hip."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import stepwise as sw
from conftest import SMALL_CLASSES, small_class_contact_variant
from test_gpu_parity import phase_parity
from test_gpu_teacher_forced import check, factory

# These instantiations have never run on hardware, and round 4's only GPU memory fault came from a contact instantiation the lane
# emulator had no complaint about: a fault here would take the whole `pytest -m gpu` process down with it.  So the file is skipped
# unless asked for (tools/gpu_r5.sh runs it in a pytest process of its own); after its first green run on an MI355X the opt-in of
# the kernels (EMPC_EXPERIMENTAL_CONTACT) and this guard go away together.
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("EMPC_RUN_EXPERIMENTAL_GPU_TESTS", "") in ("", "0"),
                                 reason="kernels never run on hardware: set EMPC_RUN_EXPERIMENTAL_GPU_TESTS=1 (tools/gpu_r5.sh experimental)")]


@pytest.fixture(autouse=True)
def _opt_in(monkeypatch):
    """these instantiations are opt-in until this file has passed on hardware (empc_solver.hip find_table)"""
    monkeypatch.setenv("EMPC_EXPERIMENTAL_CONTACT", "1")

VARIANTS = [("ContactModel3D", (0.0, 0.0)), ("ContactModel6D", (5.0, 2.0))]


@pytest.mark.parametrize("contact,gains", VARIANTS)
@pytest.mark.parametrize("robot", sorted(SMALL_CLASSES))
def test_small_class_contact_phase_parity(empc, tmp_path, robot, contact, gains):
    """linearize / backward / rollout kernels against the oracle's calcDiff / backwardPass / forwardPass"""
    _, problem = small_class_contact_variant(empc, tmp_path, robot, contact, gains)
    assert empc.solver_supported(problem), empc.last_error()
    assert problem.desc.has_contact
    phase_parity(empc, problem, robot + "/" + contact)


@pytest.mark.parametrize("amplitude", [0.02, 0.002])
@pytest.mark.parametrize("robot,contact,gains", [("hexacopter370", "ContactModel3D", (0.0, 0.0)), ("hexacopter680_flying_arm_2", "ContactModel6D", (5.0, 2.0))])
def test_small_class_contact_stepwise(empc, tmp_path, robot, contact, gains, amplitude):
    """Every iteration of the oracle's paths reproduced by the device and the other way round (tests/stepwise.py).  Bounds on
    the waived share: no hardware measurement exists yet, so they come from the same driver on the CPU lane emulator of these
    kernel bodies (round 5: 0.257 on the hexacopter370 variant at amplitude 0.02 -- 27 of 113 iterates explode on both sides, the
    cold-start behaviour of the hover files, LABNOTES.md divergence study -- and 0.000 on the three other cases) + 0.10; the
    gentle perturbation leaves nothing to waive.  To be replaced by measured + 0.05 after the first hardware run."""
    _, problem = small_class_contact_variant(empc, tmp_path, robot, contact, gains)
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 3, nq=d.model.nq, amplitude=amplitude)
    x0s[0] = problem.x0
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, maxiter=40, tape_every=13, do_same_minimum=False)
    check(rep, max_waived=0.36 if (robot == "hexacopter370" and amplitude == 0.02) else 0.10, min_asserted=30, max_exploded=40)


def test_small_class_contact_solves_are_finite_and_batch_independent(empc, tmp_path):
    """a batch of four equals four batches of one, bit for bit (the kernels of this class follow the same rules)"""
    _, problem = small_class_contact_variant(empc, tmp_path, "iris", "ContactModel3D", (0.0, 0.0))
    d = problem.desc
    x0s = empc.perturbed_x0s(problem.x0, 4, nq=d.model.nq, amplitude=0.02)
    s = empc.SolverSbFDDP(problem, batch=4)
    s.solve([], [], 30, x0s=x0s)
    assert np.isfinite(s.xs_batch).all() and s.kernel_family == "runtime model"
    one = empc.SolverSbFDDP(problem, batch=1)
    for b in range(4):
        one.solve([], [], 30, x0s=x0s[b:b + 1])
        assert np.array_equal(one.xs_batch[0], s.xs_batch[b]) and one.iter_batch[0] == s.iter_batch[b]


def test_arm5_mixed_contact(empc, tmp_path):
    """stages of both contact types on the (6, 6) robot class (empc_inst_6_6_contact_mixed.hip, opt-in): phase parity and
    step-wise parity on the push_slide robot with a ContactModel3D and a ContactModel6D stage appended"""
    from conftest import arm5_mixed_contact_variant
    _, problem = arm5_mixed_contact_variant(empc, tmp_path)
    assert empc.solver_supported(problem), empc.last_error()
    phase_parity(empc, problem, "arm5/mixed")
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 3, nq=d.model.nq, amplitude=0.002)
    x0s[0] = problem.x0
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, maxiter=40, tape_every=13, do_same_minimum=False)
    check(rep, max_waived=0.10, min_asserted=20, max_exploded=10)
