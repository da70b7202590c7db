"""Batch sharding across ranks (one process per GPU) and the single collective of the algorithm: the gather of results.

Trajectories are independent (no cross-trajectory term anywhere in src/sbfddp.cpp), so rank r simply owns the rollouts
[r * per_rank, (r + 1) * per_rank) of the global batch and nothing is exchanged during a solve.
"""
import numpy as np


def shard_bounds(global_batch, world_size, rank):
    """Contiguous shards; the first `global_batch % world_size` ranks hold one extra rollout."""
    base, extra = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(array, world_size, rank):
    lo, hi = shard_bounds(array.shape[0], world_size, rank)
    return np.ascontiguousarray(array[lo:hi])


def pack_results(xs, us_squash, cost, iters):
    """One row per rollout: xs | us_squash | cost | iters (float64)."""
    B = xs.shape[0]
    return np.ascontiguousarray(np.concatenate([xs.reshape(B, -1), us_squash.reshape(B, -1), cost.reshape(B, 1),
                                                iters.reshape(B, 1).astype(np.float64)], axis=1))


def unpack_results(rows, T, nx, nu):
    B = rows.shape[0]
    a = (T + 1) * nx
    b = a + T * nu
    return (rows[:, :a].reshape(B, T + 1, nx), rows[:, a:b].reshape(B, T, nu), rows[:, b], rows[:, b + 1].astype(np.int64))


def gather_rows_device(dist, rows_dev, world_size, rank):
    """Gather equally sized device-resident result rows (a torch CUDA tensor per rank) on rank 0 over RCCL; the gathered
    rows stay on rank 0's GPU (list of per-rank tensors) -- nothing passes through host memory."""
    import torch
    out = [torch.empty_like(rows_dev) for _ in range(world_size)] if rank == 0 else None
    dist.gather(rows_dev, out, dst=0)
    return out


def gather_results(dist, rows, world_size, rank, device=None, global_batch=None):
    """Gather per-rank result rows on rank 0 (torch.distributed gather; RCCL on GPUs, gloo in the CPU tests).
    Shards may differ in size by one row, so rows are padded to the largest shard."""
    import torch
    n = global_batch if global_batch is not None else rows.shape[0] * world_size
    sizes = [shard_bounds(n, world_size, r)[1] - shard_bounds(n, world_size, r)[0] for r in range(world_size)]
    mx = max(sizes)
    buf = np.zeros((mx, rows.shape[1]))
    buf[:rows.shape[0]] = rows
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world_size)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return None
    return np.concatenate([o.cpu().numpy()[:sizes[r]] for r, o in enumerate(out)], axis=0)
