import os, sys, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, empc_loader, numpy as np
empc = empc_loader.load()
t = empc.Trajectory(); t.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
p = t.createProblem(80, True, "IntegratedActionModelEuler")
torch.cuda.init()
def free(): return torch.cuda.mem_get_info()[0] / 2**20
f0 = free()
for i in range(30):
    s = empc.SolverSbFDDP(p, batch=256)
    s.solve([], [], 3)
    del s; gc.collect()
    if i in (0, 9, 29): print("after", i + 1, "create/solve/destroy cycles: free MiB", round(free(), 1), "delta", round(free() - f0, 1))
# streamed solves: the queue and the result rows are reallocated per stream and released with the solver
x0s = empc.perturbed_x0s(p.x0, 600, nq=p.desc.model.nq)
f1 = free()
for i in range(12):
    s = empc.SolverSbFDDP(p, batch=128)
    for _ in range(3):
        s.solve_stream(x0s, 3)
    del s; gc.collect()
    if i in (0, 11): print("after", i + 1, "create / 3 streams / destroy cycles: free MiB", round(free(), 1), "delta", round(free() - f1, 1))
