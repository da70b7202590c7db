"""The N > 1 path of bench.py on the GPU box, started exactly as the driver starts it (`python3 bench.py --gpus 2 ...`: the
parent spawns two ranks through torch.distributed.run) with both ranks sharing the one visible device and gloo as the
collective backend (RCCL refuses two ranks on one GPU).  Rank 0's gathered rows must be, bit for bit, a single-rank solve
of the global batch: sharding, per-rank solve, row packing and the gather change nothing."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["stream", "batch"])
def test_two_ranks_equal_one_rank(empc, problems, tmp_path, mode):
    B, world = 48, 2
    dump = str(tmp_path / "rows.npy")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--batch", str(B),
                        "--steps", "1", "--warmup", "0", "--mode", mode, "--config", "displacement", "--dump-rows", dump],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["ranks_seen"] == world
    assert out["ranks_matching_golden_vector"] == world  # every rank reproduced the committed eagle_catch vector on its device
    assert out["config"]["mode"] == mode and out["value"] > 0
    rows = np.load(dump)
    # the same global batch on one rank
    _, problem = problems["displacement"]
    d = problem.desc
    x0_all = empc.perturbed_x0s(problem.x0, B * world, nq=d.model.nq)
    s = empc.SolverSbFDDP(problem, batch=B * world)
    s.solve([], [], 100, x0s=x0_all)
    sharding = importlib.import_module("eagle_mpc_amd.sharding")
    ref = sharding.pack_results(s.xs_batch, s.us_squash_batch, s.cost_batch, s.iter_batch)
    assert rows.shape == ref.shape
    assert np.array_equal(rows, ref, equal_nan=True)
