"""GPU box: per-kernel launch times of the option variants that have no bench configuration (RK4 nodes, box solvers,
ContactModel6D, the mixed-contact problem, contact on the 11-DoF arm), B = 1024.  usage: python3 tools/gpu_variant_times.py"""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import empc_loader  # noqa: E402

empc = empc_loader.load()
from conftest import CONFIGS, arm5_contact_variant, contact_variant, mixed_contact_variant  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
B = 1024


def run(tag, problem, cls=None, maxiter=100):
    d = problem.desc
    x0s = empc.perturbed_x0s(problem.x0, B, nq=d.model.nq)
    s = (cls or empc.SolverSbFDDP)(problem, batch=B)
    s.solve([], [], maxiter, x0s=x0s)  # warm-up
    t0 = time.perf_counter()
    s.solve([], [], maxiter, x0s=x0s)
    dt = time.perf_counter() - t0
    st = s.stats()
    n = {k: max(1, st["n_" + k]) for k in ("linearize", "backward", "rollout")}
    out = {"variant": tag, "T": d.T, "sweeps": st["sweeps"], "ms_per_solve": 1e3 * dt,
           "batched_iters_per_s": st["total_iters"] / B / dt,
           "ms_per_launch": {k: st["ms_" + k] / n[k] for k in n}}
    print(json.dumps(out), flush=True)


def traj(name):
    tr = empc.Trajectory()
    tr.autoSetup(empc.yaml_path(CONFIGS[name][0]))
    return tr


for name in ("displacement", "eagle_catch"):
    tr = traj(name)
    run(name + "/Euler", tr.createProblem(CONFIGS[name][1], True, "IntegratedActionModelEuler"))
    run(name + "/RK4", tr.createProblem(CONFIGS[name][1], True, "IntegratedActionModelRK4"))
tr = traj("displacement")
box = tr.createProblem(80, False, "IntegratedActionModelEuler")
run("displacement/SolverBoxFDDP", box, empc.SolverBoxFDDP, 30)
run("displacement/SolverBoxDDP", box, empc.SolverBoxDDP, 30)
run("eagle_catch/ContactModel6D", contact_variant(empc, tmp, "ContactModel6D", (0.0, 0.0))[1])
run("eagle_catch/mixed 3D+6D", mixed_contact_variant(empc, tmp)[1])
run("arm5 push + ContactModel3D", arm5_contact_variant(empc, tmp, "ContactModel3D")[1])
run("push_slide", traj("push_slide").createProblem(13, True, "IntegratedActionModelEuler"))
