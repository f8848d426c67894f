"""Restart sharding across ranks (one process per GPU) and the end-of-run gather.

The reference fans restarts out as OS processes and gathers their result dicts through a Manager Queue
(``xframe/projects/fxs/reconstruct.py:151-155``, ``xframe/Multiprocessing.py:360-437``); restarts never
communicate.  Here restart ``i`` runs on rank ``i % world_size`` (the reference maps clients to GPUs with
``(pid // n_control_workers) % n_gpus``, ``Multiprocessing.py:1275-1277``); the only collective is the final
gather: per-restart scalars via all_gather, the rotation-invariant B_l via all_reduce, the grid-sized arrays of the best
restarts as tensors to rank 0 (RCCL on GPUs, gloo on CPU for tests); only the light parts of the result dicts travel as
pickled objects.
"""
import numpy as np


def shard_restarts(n_total, rank, world_size):
    """indices of the restarts owned by `rank` (round robin)."""
    return list(range(rank, n_total, world_size))


def _dist():
    """torch.distributed when this process is part of an initialised process group, else None.  A process that has not imported
    torch cannot be in one -- and importing it just to ask costs ~0.9 s, more than the whole tutorial schedule on the device."""
    import sys
    if 'torch' not in sys.modules:
        return None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _is_heavy(v):
    """grid-shaped (Nq, n_theta, n_phi[, 3]) or B_l-shaped (L+1, Nq, Nq) arrays: these only travel for the selected restarts"""
    return isinstance(v, np.ndarray) and v.dtype != object and v.ndim >= 3


def _light(res):
    """result dict without its grid-sized arrays (scalars, error histories, unknowns stay)"""
    out = {}
    for k, v in res.items():
        if _is_heavy(v):
            continue
        if isinstance(v, dict):
            v = {kk: vv for kk, vv in v.items() if not _is_heavy(vv)}
        elif isinstance(v, (list, tuple)) and any(_is_heavy(x) for x in v):
            continue
        out[k] = v
    return out


def _heavy_keys(res):
    return sorted(k for k, v in res.items() if _is_heavy(v))


SENT_FROM_DEVICE = 0          # arrays the last gathers of this process sent out of engine buffers (diagnostic, tests)


def gather_results(local_results, local_ids, n_total, rank, world_size, n_full=8, device=None, device_source=None):
    """End-of-run gather (reference: Manager Queue + argsort of the final errors, reconstruct.py:160-177).

    1. all_gather of the per-restart scalars (last main error) -> every rank knows the global ranking;
    2. the light parts of the result dicts (scalars, error histories, unknowns: a few KB each) go to rank 0 as objects;
    3. the grid-sized arrays (densities, supports, B_l: ~7 x 16 MiB per restart at 128 x L32) travel only for the ``n_full``
       best restarts, as tensors point to point to rank 0 (RCCL send / recv over xGMI with the nccl backend; gloo on CPU in the
       tests).  With the nccl backend the owner sends the arrays that are engine state -- the four densities and the two support
       masks, 6 of the ~7 grids -- straight out of HBM: ``device_source[restart id](key)`` hands back a device tensor (a
       device-to-device copy out of the engine's slot arrays, Engine.t_state), so nothing of grid size crosses PCIe on the sending
       side; what the engine does not hold in that form (B_l, the initial density, arrays an output modifier has changed) is
       staged from the host copy of the result dict.  The receiver copies each tensor to the host once (the result dicts are host
       arrays, as the reference's).
    Rank 0 returns an object array of all restarts -- full dicts for its own and for the selected ones, light dicts (flagged
    ``'gathered': 'light'``) for the rest; other ranks return their own results unchanged."""
    global SENT_FROM_DEVICE
    if world_size == 1:
        return local_results
    dist = _dist()
    if dist is None:
        return local_results
    import torch
    n_max = -(-n_total // world_size)
    err = np.full(n_max, np.inf)
    for j, r in enumerate(local_results):
        err[j] = float(np.asarray(r['error_dict']['main'])[-1])
    table = gather_scalars(err, device)                                    # (world, n_max)
    final = np.full(n_total, np.inf)
    for rk in range(world_size):
        ids = shard_restarts(n_total, rk, world_size)
        final[ids] = table[rk, :len(ids)]
    selected = [int(i) for i in np.argsort(final, kind='stable')[:max(int(n_full), 0)]]
    local_ids = list(local_ids)
    payload = [(i, _light(r), [(k, str(r[k].dtype), tuple(r[k].shape)) for k in _heavy_keys(r)])
               for i, r in zip(local_ids, local_results)]
    gathered = [None] * world_size if rank == 0 else None
    dist.gather_object(payload, gathered, dst=0)
    out = None
    if rank == 0:
        out = np.empty(n_total, dtype=object)
        metas = {}
        for part in gathered:
            for i, r, meta in part:
                r['gathered'] = 'light'
                out[i] = r
                metas[i] = meta
        for i, r in zip(local_ids, local_results):
            out[i] = r
    # heavy arrays of the selected restarts, in a fixed order on both ends
    # tensors of `device` travel as they are: the GPU with nccl (= RCCL); a CPU `device` (gloo rehearsal of the same code path)
    use_dev = device is not None and (dist.get_backend() == 'nccl' or getattr(device, 'type', None) == 'cpu')
    for i in selected:
        owner = i % world_size
        if owner == 0:
            continue
        if rank == owner:
            res = local_results[local_ids.index(i)]
            src = (device_source or {}).get(i) if use_dev else None
            for k in _heavy_keys(res):
                t = src(k) if src is not None else None
                if t is not None and tuple(t.shape) == tuple(res[k].shape) and t.element_size() == res[k].dtype.itemsize:
                    dist.send(t.contiguous().view(torch.uint8).reshape(-1), dst=0)           # from the engine's HBM
                    SENT_FROM_DEVICE += 1
                    continue
                a = np.ascontiguousarray(res[k])
                t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
                dist.send(t.to(device) if use_dev else t, dst=0)
        elif rank == 0:
            for k, dt, shape in metas[i]:
                nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
                t = torch.empty(nbytes, dtype=torch.uint8, device=device if use_dev else 'cpu')
                dist.recv(t, src=owner)
                out[i][k] = t.cpu().numpy().view(np.dtype(dt)).reshape(shape)
            out[i]['gathered'] = 'full'
    return out if rank == 0 else local_results


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
