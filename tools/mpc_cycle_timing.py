#!/usr/bin/env python3
"""Where one controller cycle of the Carrot MPC loop (256 plants) spends its wall time: updateProblem (host rules +
re-prepared table + upload), solve, plant step.  GPU box: python3 tools/mpc_cycle_timing.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import empc_loader
empc = empc_loader.load()
traj = empc.Trajectory(); traj.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
prob = traj.createProblem(80, True, "IntegratedActionModelEuler")
pl = empc.SolverSbFDDP(prob, batch=1); pl.solve([], [], 100)
xs, us = np.array(pl.xs), np.array(pl.us)
yaml50 = os.path.join(os.path.dirname(empc.YAML_DIR), "mpc", "carrot_50knots.yaml")
B = 256
mpc = empc.CarrotMpc(traj, xs, 80, yaml50, batch=B)
mpc.updateProblem(0)
s = mpc.solver
d = mpc.problem.desc
s.plant_states = empc.perturbed_x0s(xs[0], B, nq=d.model.nq, amplitude=0.02)
s.solve(xs[:d.T + 1], us[:d.T], 100, x0s="plant")
s.convergence_init = 1e-3
t = 0
acc = {"updateProblem(host rules)": 0.0, "update_problem(prepare+upload)": 0.0, "solve": 0.0, "stats": 0.0, "plant_step": 0.0}
L = empc.lib()
import ctypes as C
N = 200
for it in range(N + 20):
    if it == 20:
        acc = {k: 0.0 for k in acc}
    a = time.perf_counter()
    L.empc_carrot_mpc_update_problem(mpc._h, int(t))
    b = time.perf_counter()
    s.update_problem()
    c = time.perf_counter()
    s.solve("previous", "previous", mpc.iters, x0s="plant")
    e = time.perf_counter()
    st = s.stats()
    f = time.perf_counter()
    s.plant_step(2)
    g = time.perf_counter()
    acc["updateProblem(host rules)"] += b - a; acc["update_problem(prepare+upload)"] += c - b; acc["solve"] += e - c
    acc["stats"] += f - e; acc["plant_step"] += g - f
    t += 2
tot = sum(acc.values())
print("iters per cycle", mpc.iters, "knots", d.T, "sets", d.n_sets)
for k, v in acc.items():
    print("%-34s %7.1f us per cycle  %4.1f %%" % (k, 1e6 * v / N, 100 * v / tot))
print("total %.1f us per cycle; kernel time per solve (stats): %s" % (1e6 * tot / N, {k: round(v, 3) for k, v in st.items() if k.startswith("ms_")}))
