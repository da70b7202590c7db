"""The HIP linearize kernel against the NumPy second restatement's fixtures (tests/golden/second_restatement/*.npz: ABA, dense
KKT, complex-step derivatives -- no formula shared with the kernel's hand-derived tangent recursions): every record block of
one node per distinct cost set of the BASELINE problems, the contact-option variants and the RK4 integrator, at 1e-9.
No oracle involved: the fixtures are data.  Reference call sites: src/factory/int-action.cpp:26-31, diff-action.cpp:31,34."""
import ast
import os

import numpy as np
import pytest

from conftest import contact_variant

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "second_restatement")
NAMES = sorted(f[:-4] for f in os.listdir(FIX) if f.endswith(".npz"))


def rel(a, b):
    a, b = np.ravel(np.asarray(a, dtype=float)), np.ravel(np.asarray(b, dtype=float))
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


@pytest.mark.parametrize("name", NAMES)
def test_linearize_kernel_matches_second_restatement(empc, tmp_path, name):
    g = np.load(os.path.join(FIX, name + ".npz"))
    meta = ast.literal_eval(str(g["meta"]))
    integrator = str(g["integrator"])
    if "contact" in meta:
        _, problem = contact_variant(empc, tmp_path, meta["contact"], tuple(meta["gains"]), integrator=integrator)
    else:
        t = empc.Trajectory()
        t.autoSetup(empc.yaml_path(meta["yaml"]))
        problem = t.createProblem(meta["dt_ms"], True, integrator)
    d = problem.desc
    s = empc.SolverSbFDDP(problem, batch=1)
    xs = np.tile(np.array(problem.x0), (d.T + 1, 1))
    us = np.full((d.T, d.nu), 4.0)
    us[:, d.n_rotors:] = 0.0
    for i, t in enumerate(g["knots"]):
        xs[int(t)] = g["xs"][i]
        if int(t) < d.T:
            us[int(t)] = g["us"][i]
    tape = s.linearize(xs[None], us[None], smooth=float(g["smooth"]), is_feasible=False, x0s=np.array(problem.x0)[None])
    tol = 1e-9 if "rk4" not in name else 1e-8  # (RK4 nodes: four stage terms on Hessians of 1e9, see test_gpu_teacher_forced)
    for i, t in enumerate(g["knots"]):
        b = s.tape_blocks(tape[0, int(t)])
        for key in ("Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu"):
            if int(t) == d.T and key in ("Fu", "Lu", "Lxu", "Luu"):
                continue
            assert rel(b[key], g[key][i]) < tol, (name, int(t), key, rel(b[key], g[key][i]))
        assert rel(b["cost"], g["cost"][i]) < tol
