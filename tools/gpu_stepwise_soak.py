"""GPU box: the step-wise parity driver (tests/stepwise.py) on initial states the test suite does not use (other seeds, more
rollouts) -- a soak run; raises on the first claim that fails.  usage: python3 tools/gpu_stepwise_soak.py [rollouts] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import empc_loader  # noqa: E402

empc = empc_loader.load()
import oracle_binding as ob  # noqa: E402
import stepwise as sw  # noqa: E402
from conftest import CONFIGS  # noqa: E402
from conftest import arm5_contact_variant, contact_variant, mixed_contact_variant  # noqa: E402
from test_gpu_teacher_forced import factory  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
for name, amp, n in (("eagle_catch", 0.05, B), ("displacement", 0.05, B // 4), ("hover", 0.02, B // 4)):
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    problem = tr.createProblem(CONFIGS[name][1], True, "IntegratedActionModelEuler")
    d = problem.desc
    prm = ob.default_params()
    x0s = empc.perturbed_x0s(problem.x0, n, nq=d.model.nq, amplitude=amp, seed=seed)
    t0 = time.time()
    rep = sw.stepwise_parity(factory(empc, problem, prm), d, prm, x0s, tape_every=97, tight=1e-12, tight_maxiter=500)
    keep = {k: v for k, v in rep.items() if not isinstance(v, (list, dict))}
    print(json.dumps({"problem": name, "rollouts": n, "seed": seed, "seconds": round(time.time() - t0, 1), **keep}), flush=True)

# the option variants (other contact types, the box solvers, RK4 nodes) on a few rollouts each
import pathlib  # noqa: E402
import tempfile  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
nv = max(4, B // 16)


def run(tag, problem, prm, cls=None, **kw):
    d = problem.desc
    x0s = empc.perturbed_x0s(problem.x0, nv, nq=d.model.nq, amplitude=0.02, seed=seed)
    t0 = time.time()
    rep = sw.stepwise_parity(factory(empc, problem, prm, cls), d, prm, x0s, tape_every=53, **kw)
    keep = {k: v for k, v in rep.items() if not isinstance(v, (list, dict))}
    print(json.dumps({"problem": tag, "rollouts": nv, "seed": seed, "seconds": round(time.time() - t0, 1), **keep}), flush=True)


run("eagle_catch/ContactModel6D", contact_variant(empc, tmp, "ContactModel6D", (7.0, 2.0))[1], ob.default_params())
run("eagle_catch/mixed 3D+6D", mixed_contact_variant(empc, tmp, (5.0, 1.0))[1], ob.default_params())
run("arm5/ContactModel3D", arm5_contact_variant(empc, tmp, "ContactModel3D", (4.0, 2.0))[1], ob.default_params(), maxiter=60)
tr = empc.Trajectory()
tr.autoSetup(empc.yaml_path(CONFIGS["displacement"][0]))
run("displacement/RK4", tr.createProblem(80, True, "IntegratedActionModelRK4"), ob.default_params(), tol_tape=1e-8)
for st, cls in ((1, empc.SolverBoxFDDP), (2, empc.SolverBoxDDP)):
    prm = ob.default_params()
    prm.solver_type = st
    run("displacement/box solver %d" % st, tr.createProblem(80, False, "IntegratedActionModelEuler"), prm, cls, maxiter=30, do_same_minimum=False)
