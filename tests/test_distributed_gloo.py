"""N > 1 path on CPU: world_size-2 gloo processes shard the global batch exactly as bench.py does, 'solve' their shard
(through the oracle here -- there is no GPU in this container) and gather the results on rank 0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, global_batch, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    empc = ob.empc
    sh = __import__("importlib").import_module("eagle_mpc_amd.sharding")
    traj = empc.Trajectory()
    traj.autoSetup(empc.yaml_path("hexacopter370_flying_arm_3/trajectories/displacement.yaml"))
    problem = traj.createProblem(400, True, "IntegratedActionModelEuler")  # T = 20: quick
    d = problem.desc
    x0_all = empc.perturbed_x0s(problem.x0, global_batch, nq=d.model.nq)
    x0s = sh.shard(x0_all, world, rank)
    r = ob.solve_batch(d, x0s, 30, nthreads=1)
    rows = sh.pack_results(r["xs"], r["us_squash"], r["cost"], r["iter"])
    allrows = sh.gather_results(dist, rows, world, rank, global_batch=global_batch)
    dist.barrier()
    if rank == 0:
        xs, usq, cost, iters = sh.unpack_results(allrows, d.T, d.nx, d.nu)
        ref = ob.solve_batch(d, x0_all, 30, nthreads=1)
        np.savez(tmp, ok=np.array([np.array_equal(xs, ref["xs"]) and np.array_equal(usq, ref["us_squash"]) and
                                   np.array_equal(cost, ref["cost"]) and np.array_equal(iters, ref["iter"])]),
                 n=np.array([xs.shape[0]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [6, 5])
def test_world_size_2_shard_and_gather(empc, tmp_path, global_batch):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(2, port, global_batch, out), nprocs=2, join=True)
    r = np.load(out)
    assert r["n"][0] == global_batch and bool(r["ok"][0])


def test_shard_bounds(empc):
    import importlib
    sh = importlib.import_module("eagle_mpc_amd.sharding")
    for n in (1, 7, 8, 1024, 4096):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                lo, hi = sh.shard_bounds(n, w, r)
                cover += list(range(lo, hi))
            assert cover == list(range(n))
