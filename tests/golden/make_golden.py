#!/usr/bin/env python3
"""Generate tests/golden/solutions/*.npz from the CPU oracle (oracle/liboracle.so).

These are regression vectors of THIS repository's restatement of the algorithm, not outputs of the reference (which cannot be
built or imported here): they pin the oracle against accidental change and let the GPU parity tests run against
committed data.  For every BASELINE config: the unperturbed rollout (b = 0) and two perturbed ones where the problem is well
conditioned, `solve([], [], maxiter=100)`: xs, us, us_squash, cost, iter, status, and the x0s used.

    python tests/golden/make_golden.py            # rewrites the files (run only when the oracle changes on purpose)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import empc_loader  # noqa: E402
import oracle_binding as ob  # noqa: E402

CONFIGS = {  # name: (yaml, dt_ms, number of rollouts, perturbation amplitude)
    "hover": ("hexacopter370/trajectories/hover.yaml", 40, 1, 0.0),
    "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80, 3, 0.05),
    "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32, 1, 0.0),
    "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13, 2, 0.05),
}


def main():
    empc = empc_loader.load()
    out_dir = os.path.join(HERE, "solutions")
    os.makedirs(out_dir, exist_ok=True)
    for name, (rel, dt, n, amp) in CONFIGS.items():
        traj = empc.Trajectory()
        traj.autoSetup(empc.yaml_path(rel))
        problem = traj.createProblem(dt, True, "IntegratedActionModelEuler")
        d = problem.desc
        x0s = empc.perturbed_x0s(problem.x0, n, nq=d.model.nq, amplitude=amp)
        r = ob.solve_batch(d, x0s, 100, nthreads=min(n, 8), want_traj=True)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), x0s=x0s, xs=r["xs"], us=r["us"], us_squash=r["us_squash"],
                            cost=r["cost"], iter=r["iter"], status=r["status"], dt_ms=dt, yaml=rel)
        print(name, "iters", r["iter"], "cost", r["cost"])


if __name__ == "__main__":
    main()
