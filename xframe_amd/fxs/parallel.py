"""Restart sharding across ranks (one process per GPU) and the end-of-run gather.

The reference fans restarts out as OS processes and gathers their result dicts through a Manager Queue
(``xframe/projects/fxs/reconstruct.py:151-155``, ``xframe/Multiprocessing.py:360-437``); restarts never
communicate.  Here restart ``i`` runs on rank ``i % world_size`` (the reference maps clients to GPUs with
``(pid // n_control_workers) % n_gpus``, ``Multiprocessing.py:1275-1277``); the only collective is the final
gather: per-restart scalars and the rotation-invariant B_l via all_gather / all_reduce (RCCL on GPUs, gloo on
CPU for tests), result dicts via gather_object to rank 0.
"""
import numpy as np


def shard_restarts(n_total, rank, world_size):
    """indices of the restarts owned by `rank` (round robin)."""
    return list(range(rank, n_total, world_size))


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def gather_results(local_results, local_ids, n_total, rank, world_size):
    """Gather the per-restart result dicts on rank 0 (other ranks keep only their own)."""
    dist = _dist()
    if world_size == 1 or dist is None:
        return local_results
    payload = [(i, r) for i, r in zip(local_ids, local_results)]
    gathered = [None] * world_size if rank == 0 else None
    dist.gather_object(payload, gathered, dst=0)
    if rank != 0:
        return local_results
    out = np.empty(n_total, dtype=object)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out


def gather_scalars(values, device=None):
    """all_gather of a per-rank 1-D float64 array (equal length on every rank) -> (world, n) array."""
    import torch
    dist = _dist()
    v = np.ascontiguousarray(values, dtype=np.float64)
    if dist is None or dist.get_world_size() == 1:
        return v[None]
    t = torch.from_numpy(v.copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return np.stack([o.cpu().numpy() for o in outs])


def average_invariants(local_sum, local_count, device=None):
    """all_reduce(sum) of the rotation-invariant B_l accumulated over a rank's restarts -> global mean.

    Densities of different restarts are only comparable after SO(3) x inversion alignment
    (reference: ``xframe/projects/fxs/average.py``, a "next" row), so the averaged quantity is B_l."""
    import torch
    dist = _dist()
    s = np.ascontiguousarray(local_sum)
    if dist is None or dist.get_world_size() == 1:
        return s / max(local_count, 1)
    t = torch.view_as_real(torch.from_numpy(s.astype(np.complex128).copy())).contiguous()
    n = torch.tensor([float(local_count)], dtype=torch.float64)
    if device is not None:
        t, n = t.to(device), n.to(device)
    dist.all_reduce(t)
    dist.all_reduce(n)
    out = torch.view_as_complex(t.cpu()).numpy()
    return out / max(float(n.cpu()[0]), 1.0)
