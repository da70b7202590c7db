"""A problem whose stages use ContactModel3D AND ContactModel6D (src/factory/contacts.cpp:26-79 builds either per stage;
SURVEY.md section 8 row a17): the mixed kernel instantiation (empc_inst_4_6_contact_mixed.hip, CT_MIXED) runs the bodies of
the two single-type instantiations behind a branch on the node's contact type.  No shipped YAML mixes them: the problem is
eagle_catch with a 6D "hold" stage after its 3D "grasp" stage (conftest.mixed_contact_variant)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity_criteria as pc
import stepwise as sw
from conftest import contact_variant, mixed_contact_variant
from test_gpu_parity import phase_parity
from test_gpu_teacher_forced import check, factory, save

pytestmark = pytest.mark.gpu


def contact_types(d):
    return {d.sets[d.knot_set[t]].contacts[0].type for t in range(d.T + 1) if d.sets[d.knot_set[t]].ncontacts > 0}


@pytest.mark.parametrize("gains6", [(0.0, 0.0), (8.0, 3.0)])
def test_mixed_contact_phase_parity(empc, tmp_path, gains6):
    _, problem = mixed_contact_variant(empc, tmp_path, gains6)
    assert contact_types(problem.desc) == {empc.T.CONTACT_3D, empc.T.CONTACT_6D}
    assert empc.solver_supported(problem), empc.last_error()
    phase_parity(empc, problem, "eagle_catch/mixed")


def test_mixed_instantiation_equals_single_type_ones(empc, tmp_path):
    """On a problem with one contact type the mixed instantiation must reproduce that type's own instantiation (same source
    behind a branch; the compiler may contract multiply-adds differently in the two, so: rounding level, not bitwise) --
    forced through EMPC_FORCE_MIXED_CONTACT, a diagnostic switch read at solver creation."""
    import os
    for contact in ("ContactModel3D", "ContactModel6D"):
        _, problem = contact_variant(empc, tmp_path, contact, (5.0, 2.0))
        d = problem.desc
        x0s = empc.perturbed_x0s(problem.x0, 3, nq=d.model.nq, amplitude=0.02)
        # (the single-type runtime-model instantiation: the baked-robot one of ContactModel3D is another compilation of the
        #  arithmetic and differs at rounding level x the amplification of this cold start, tests/test_gpu_baked.py)
        os.environ["EMPC_BAKED"] = "0"
        try:
            a = empc.SolverSbFDDP(problem, batch=3)
        finally:
            del os.environ["EMPC_BAKED"]
        assert a.kernel_family == "runtime model"
        a.solve([], [], 2, x0s=x0s)
        os.environ["EMPC_FORCE_MIXED_CONTACT"] = "1"
        try:
            b = empc.SolverSbFDDP(problem, batch=3)
        finally:
            del os.environ["EMPC_FORCE_MIXED_CONTACT"]
        b.solve([], [], 2, x0s=x0s)
        assert np.array_equal(a.iter_batch, b.iter_batch), contact
        # (two iterations of a cold start with a 1e-9 regularisation amplify the last bit to ~1e-6: a smoke bound; the claim
        #  proper is the phase-level agreement from identical inputs below)
        assert np.abs(a.xs_batch - b.xs_batch).max() < 1e-5 and np.abs(a.us_batch - b.us_batch).max() < 1e-4, contact
        assert np.allclose(a.cost_batch, b.cost_batch, rtol=1e-6)
        from test_gpu_baked import assert_phase_agreement
        assert_phase_agreement(empc, a, b, problem, 3)


def test_mixed_contact_stepwise(empc, tmp_path):
    _, problem = mixed_contact_variant(empc, tmp_path, (8.0, 3.0))
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, 4, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, tape_every=31)
    save("contact_mixed", rep)
    check(rep, max_exploded=5)  # (measured r04 on hardware: 0 exploded)


def test_mixed_contact_solve(empc, tmp_path):
    _, problem = mixed_contact_variant(empc, tmp_path)
    d = problem.desc
    B = 4
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq, amplitude=0.02)
    x0s[0] = problem.x0
    s = empc.SolverSbFDDP(problem, batch=B)
    s.solve([], [], 100, x0s=x0s)
    prm = empc.default_params()
    for b in range(B):
        if pc.solved(s.status_batch[b:b + 1], s.cost_batch[b:b + 1])[0]:
            o2 = ob.OracleSolver(d)
            o2.set_x0(x0s[b])
            o2.set_smooth(prm.smooth_init * prm.smooth_mult)
            c, fs, _ = o2.phase_calcdiff(s.xs_batch[b], s.us_batch[b])
            assert abs(c - s.cost_batch[b]) < 1e-8 * (1 + abs(c)) and np.abs(fs).max() < 1e-7
    assert pc.solved(s.status_batch, s.cost_batch).any()
