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
