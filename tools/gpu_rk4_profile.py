"""GPU box: two RK4-node solves of displacement, B = 1024 (the program rocprofv3 wraps for the RK4 kernel breakdown).
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/rk4prof -- python3 tools/gpu_rk4_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import empc_loader  # noqa: E402

empc = empc_loader.load()
tr = empc.Trajectory()
tr.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
problem = tr.createProblem(80, True, "IntegratedActionModelRK4")
x0s = empc.perturbed_x0s(problem.x0, 1024, nq=problem.desc.model.nq)
s = empc.SolverSbFDDP(problem, batch=1024)
for _ in range(2):
    s.solve([], [], 100, x0s=x0s)
print(s.stats())
