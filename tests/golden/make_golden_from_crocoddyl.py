#!/usr/bin/env python3
"""Re-pin the golden vectors on the REAL reference: regenerate tests/golden/solutions/*.npz and the first-iteration tapes with
the reference's own `eagle_mpc.SolverSbFDDP` on Crocoddyl + Pinocchio.

It CANNOT run in the build container or on the GPU box (no network, none of the packages below exists there): it is the
one-command recipe for the day an environment with the reference installed is reachable, and it is never imported by the
product, the tests, bench.py or smoke().  Requirements (reference README.md:61-95):

  * pinocchio (2.x) and its Python bindings,
  * crocoddyl built FROM SOURCE from the fork the reference names, branch `sbfddp`
    (https://github.com/PepMS/crocoddyl, README.md:73-85: SolverFDDP with the modified stopping criteria),
  * eagle_mpc itself (this reference, `make install`) with its Python bindings (`import eagle_mpc`),
  * the robot descriptions the reference's YAML files point to (`example-robot-data` + the eagle_mpc robots package).
    NOTE: the URDFs under eagle-mpc_amd/data/robots/ are synthetic stand-ins written for this repository (tests/golden/README.md);
    pass --urdf-root to make the reference load THOSE files, so that both sides solve the same robot.

What it writes (same keys and shapes as make_golden.py, so every consumer keeps working):
  tests/golden/solutions/<name>.npz        x0s, xs, us, us_squash, cost, iter, status, dt_ms, yaml   + source="crocoddyl"
  tests/golden/crocoddyl_tapes/<name>.npz  first calcDiff at the zero guess: per knot Fx, Fu, Lx, Lu, Lxx, Lxu, Luu, cost, xnext
                                           (the quantities tests/test_gpu_second_restatement.py compares the HIP tape with)
and a manifest (package versions, git revisions it can find, command line) next to them.

After it ran:  python -m pytest tests/test_golden.py tests/test_second_restatement.py   (the oracle against the new vectors:
every disagreement is a misreading of SURVEY.md Appendix A to fix in oracle/), then the GPU suite.  DESIGN.md's
"parity unpinned" notice goes away only when those pass.

    python tests/golden/make_golden_from_crocoddyl.py --yaml-root /path/to/eagle_mpc/yaml [--urdf-root eagle-mpc_amd/data/robots]
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# the BASELINE configurations (same table as make_golden.py): name -> (yaml below the yaml root, dt ms, rollouts, amplitude)
CONFIGS = {
    "hover": ("hexacopter370/trajectories/hover.yaml", 40, 1, 0.0),
    "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80, 3, 0.05),
    "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32, 1, 0.0),
    "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13, 2, 0.05),
}


def perturbed_x0s(x0, batch, nq, seed=0, amplitude=0.05):
    """The perturbation of eagle-mpc_amd/__init__.py::perturbed_x0s, restated so that this script needs nothing of the product:
    rollout 0 keeps the file's state; rollout b >= 1 adds amplitude * U(-1, 1) to every entry and renormalises the quaternion
    (generator seeded with seed * 1000003 + b: the same numbers as the product's helper)."""
    x0 = np.asarray(x0, dtype=np.float64)
    out = np.tile(x0, (batch, 1))
    for b in range(1, batch):
        rng = np.random.default_rng(seed * 1000003 + b)
        out[b] += amplitude * rng.uniform(-1, 1, size=x0.shape)
        out[b, 3:7] /= np.linalg.norm(out[b, 3:7])
    return out


def require_reference():
    try:
        import crocoddyl  # noqa: F401
        import eagle_mpc  # noqa: F401
        import pinocchio  # noqa: F401
    except ImportError as e:
        sys.exit("make_golden_from_crocoddyl.py needs the reference installed (crocoddyl@sbfddp, pinocchio, eagle_mpc): %s\n"
                 "See the header of this file and the reference's README.md:61-95." % e)
    return crocoddyl, eagle_mpc, pinocchio


def manifest(mods, argv):
    info = {"argv": argv}
    for m in mods:
        info[m.__name__] = {"version": getattr(m, "__version__", "?"), "file": getattr(m, "__file__", "?")}
    return info


def first_tape(problem):
    """calc + calcDiff of the shooting problem at (x0 everywhere, zero controls): what SolverSbFDDP's first computeDirection sees"""
    xs = [problem.x0.copy() for _ in range(problem.T + 1)]
    us = [np.zeros(m.nu) for m in problem.runningModels]
    problem.calc(xs, us)
    problem.calcDiff(xs, us)
    datas = list(problem.runningDatas) + [problem.terminalData]
    keys = ("Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu")
    tape = {k: np.array([np.asarray(getattr(d, k)) for d in datas[:-1]]) for k in keys}
    tape["cost"] = np.array([d.cost for d in datas])
    tape["xnext"] = np.array([np.asarray(d.xnext) for d in datas])
    tape["Lx_terminal"] = np.asarray(datas[-1].Lx)
    tape["Lxx_terminal"] = np.asarray(datas[-1].Lxx)
    return tape


def yaml_tree_with_absolute_urdfs(yaml_root, urdf_root):
    """Copy of the YAML tree whose `urdf: "<relative path>"` entries point below urdf_root.  NOTE: the reference resolves the
    `follow:` entries against ITS compiled-in yaml directory (getYamlPath); build it with that directory set to the copy, or
    run it with the copy mounted there."""
    import re
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="empc_yaml_")
    shutil.copytree(yaml_root, dst, dirs_exist_ok=True)
    pat = re.compile(r'^(\s*urdf:\s*")([^/"][^"]*)(")', re.M)
    for d, _, files in os.walk(dst):
        for f in files:
            if f.endswith(".yaml"):
                p = os.path.join(d, f)
                txt = open(p).read()
                new = pat.sub(lambda m: m.group(1) + os.path.join(urdf_root, m.group(2)) + m.group(3), txt)
                if new != txt:
                    open(p, "w").write(new)
    return dst


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--yaml-root", required=True, help="the reference's yaml directory (eagle_mpc/yaml)")
    ap.add_argument("--urdf-root", default=None,
                    help="make the reference load the robot descriptions of THIS repository (eagle-mpc_amd/data/robots): the YAML tree is "
                         "copied to a scratch directory with every relative `urdf:` entry rewritten to an absolute path below this root "
                         "(the reference takes absolute paths as they are, src/utils/parser_yaml.cpp:165-168)")
    ap.add_argument("--maxiter", type=int, default=100)
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    crocoddyl, eagle_mpc, pinocchio = require_reference()
    if a.urdf_root:
        a.yaml_root = yaml_tree_with_absolute_urdfs(a.yaml_root, os.path.abspath(a.urdf_root))
    sol_dir = os.path.join(a.out, "solutions")
    tape_dir = os.path.join(a.out, "crocoddyl_tapes")
    os.makedirs(sol_dir, exist_ok=True)
    os.makedirs(tape_dir, exist_ok=True)
    for name, (rel, dt, n, amp) in CONFIGS.items():
        trajectory = eagle_mpc.Trajectory()
        trajectory.autoSetup(os.path.join(a.yaml_root, rel))
        nq = trajectory.robot_model.nq
        x0s = perturbed_x0s(np.asarray(trajectory.initial_state), n, nq, amplitude=amp)
        rows = {k: [] for k in ("xs", "us", "us_squash", "cost", "iter", "status")}
        for b in range(n):
            trajectory.initial_state = x0s[b]
            problem = trajectory.createProblem(dt, True, "IntegratedActionModelEuler")
            solver = eagle_mpc.SolverSbFDDP(problem, trajectory.squash)
            converged = solver.solve([], [], a.maxiter)
            rows["xs"].append(np.array(solver.xs))
            rows["us"].append(np.array(solver.us))
            rows["us_squash"].append(np.array(solver.us_squash))
            rows["cost"].append(solver.cost)
            rows["iter"].append(solver.iter)
            rows["status"].append(1 if converged else 0)
            if b == 0:
                np.savez_compressed(os.path.join(tape_dir, name + ".npz"), dt_ms=dt, yaml=rel, **first_tape(problem))
        np.savez_compressed(os.path.join(sol_dir, name + ".npz"), x0s=x0s, xs=np.array(rows["xs"]), us=np.array(rows["us"]),
                            us_squash=np.array(rows["us_squash"]), cost=np.array(rows["cost"]), iter=np.array(rows["iter"]),
                            status=np.array(rows["status"]), dt_ms=dt, yaml=rel, source="crocoddyl")
        print(name, "iters", rows["iter"], "cost", rows["cost"])
    with open(os.path.join(a.out, "crocoddyl_manifest.json"), "w") as f:
        json.dump(manifest([crocoddyl, eagle_mpc, pinocchio, np], sys.argv), f, indent=1)
    print("wrote", sol_dir, "and", tape_dir, "-- now: python -m pytest tests/test_golden.py tests/test_second_restatement.py")


if __name__ == "__main__":
    main()
