#!/usr/bin/env python3
"""Generates tests/golden/second_restatement/*.npz: node inputs (knot, x, u, smoothing) and the outputs of the NumPy second
restatement (oracle/numpy_restatement.py: ABA, dense KKT, complex-step derivatives) for one node per distinct cost set of the
BASELINE problems, the contact-option variants and the RK4 integrator.  The fixtures pin liboracle.so (tests/
test_second_restatement.py, CPU) and the HIP linearize kernel (tests/test_gpu_second_restatement.py) to a statement of the
arithmetic that shares no derivative formula with either.  Run from the repository root: python tests/golden/make_second_restatement.py
"""
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import empc_loader  # noqa: E402

empc = empc_loader.load()
import numpy_restatement as nr  # noqa: E402
import oracle_binding as ob  # noqa: E402
from conftest import CONFIGS, arm5_two_contact_variant, contact_variant, two_contact_variant  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "second_restatement")
KEYS = ("xnext", "cost", "acc", "lam", "u_squash", "Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu")

CASES = {  # name: (config or contact variant, integrator, smoothing)
    "hover": ("hover", "IntegratedActionModelEuler", 0.1),
    "displacement": ("displacement", "IntegratedActionModelEuler", 0.1),
    "eagle_catch": ("eagle_catch", "IntegratedActionModelEuler", 0.05),
    "push_slide": ("push_slide", "IntegratedActionModelEuler", 0.1),
    "eagle_catch_contact6d_gains": (("ContactModel6D", (11.0, 5.0)), "IntegratedActionModelEuler", 0.1),
    "eagle_catch_contact3d_gains": (("ContactModel3D", (9.0, 4.0)), "IntegratedActionModelEuler", 0.1),
    "eagle_catch_rk4": ("eagle_catch", "IntegratedActionModelRK4", 0.1),
    "displacement_rk4": ("displacement", "IntegratedActionModelRK4", 0.1),
}


# Two ContactModel3D per stage (round 6; the CT_PAIR3 kernels are opt-in): fixtures of their own, in a directory the default GPU test
# does not read -- tests/test_two_contacts_emulator.py (oracle + kernel bodies on the lane emulator) and tests/test_zz_gpu_two_contacts.py
# use them.  `python tests/golden/make_second_restatement.py two_contacts` writes them.
OUT2 = os.path.join(ROOT, "tests", "golden", "second_restatement_two_contacts")
CASES2 = {
    "eagle_catch_two_contacts": (dict(robot="arm3", gains=(3.0, 1.5), gains2=(2.0, 0.7), cone_on_second=True), "IntegratedActionModelEuler", 0.1),
    "eagle_catch_two_contacts_rk4": (dict(robot="arm3", gains=(0.0, 0.0), gains2=(0.0, 0.0), cone_on_second=False), "IntegratedActionModelRK4", 0.1),
    "arm5_two_contacts": (dict(robot="arm5", gains=(2.0, 1.0), gains2=(0.0, 3.0)), "IntegratedActionModelEuler", 0.07),
}


def build_two_contact_problem(meta, integrator, tmp=None):
    tmp = tmp or pathlib.Path(tempfile.mkdtemp())
    if meta["robot"] == "arm3":
        return two_contact_variant(empc, tmp, "ContactModel3D", tuple(meta["gains"]), tuple(meta["gains2"]), integrator=integrator,
                                   cone_on_second=meta["cone_on_second"])[1]
    return arm5_two_contact_variant(empc, tmp, tuple(meta["gains"]), tuple(meta["gains2"]))[1]


def build_problem(spec, integrator):
    if isinstance(spec, dict):
        return build_two_contact_problem(spec, integrator), dict(spec, two_contacts=True)
    if isinstance(spec, tuple):
        _, problem = contact_variant(empc, pathlib.Path(tempfile.mkdtemp()), spec[0], spec[1], integrator=integrator)
        return problem, dict(contact=spec[0], gains=list(spec[1]))
    rel, dt = CONFIGS[spec]
    t = empc.Trajectory()
    t.autoSetup(empc.yaml_path(rel))
    return t.createProblem(dt, True, integrator), dict(yaml=rel, dt_ms=dt)


def main():
    cases, out = (CASES2, OUT2) if "two_contacts" in sys.argv[1:] else (CASES, OUT)
    os.makedirs(out, exist_ok=True)
    for name, (spec, integrator, smooth) in cases.items():
        problem, meta = build_problem(spec, integrator)
        d = problem.desc
        prm = ob.default_params()
        P = nr.Problem(d, prm)
        sets = nr.cost_sets_of(d, prm, smooth)
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        knots, seen = [], set()
        for t in range(d.T + 1):
            si = d.knot_set[t]
            if si not in seen or t == d.T:
                seen.add(si)
                knots.append(t)
        rec = {k: [] for k in KEYS}
        xs, us = [], []
        for t in knots:
            x = np.array(problem.x0)
            x[:3] += rng.normal(size=3) * 0.3
            q = x[3:7] + rng.normal(size=4) * 0.2
            x[3:7] = q / np.linalg.norm(q)
            x[7:] += rng.normal(size=d.nx - 7) * 0.3
            u = rng.uniform(2, 6, size=d.nu)
            u[d.n_rotors:] = rng.normal(size=d.nu - d.n_rotors) * 0.2
            r = nr.node(P, sets[d.knot_set[t]], x, None if t == d.T else u, smooth)
            xs.append(x)
            us.append(u)
            for k in KEYS:
                rec[k].append(np.atleast_1d(r[k]))
        np.savez_compressed(os.path.join(out, name + ".npz"), knots=np.array(knots), xs=np.array(xs), us=np.array(us),
                            smooth=np.array(smooth), integrator=np.array(integrator), meta=np.array(repr(meta)),
                            **{k: np.array(v) for k, v in rec.items()})
        print(name, "knots", knots)


if __name__ == "__main__":
    main()
