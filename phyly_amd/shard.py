"""
phyly_amd.shard -- site sharding across the GPUs of one node.

Sites are independent given (tree, Q, rates), so the path shards
by contiguous blocks of site patterns, one process per GPU, with no data-path
collective.  The only exchange step is the final reduction of the aggregated
outputs: one all-reduce (RCCL over xGMI when the backend is "nccl") of the
double-double partial sums -- 2 doubles for ll, 2E for edge derivatives,
2Nk for site-summed marginals.  Payloads are <= tens of KB, i.e. latency bound.

The (hi, lo) words are summed independently by the collective and renormalised
on every rank; this keeps ~1e-30 relative accuracy of the partial sums and makes
the result independent of the rank count up to fp64 rounding of the final value.
"""
import numpy as np


def shard_range(S, rank, world):
    """Contiguous block [s0, s1) of rank `rank`: ceil(S / world) sites per rank."""
    per = -(-S // world)
    s0 = min(S, rank * per)
    return s0, min(S, s0 + per)


def allreduce_dd(pairs, device=None):
    """Sum double-double partials over all ranks.

    pairs: array-like [..., 2] of (hi, lo).  Returns a float64 numpy array of the
    totals hi+lo (shape [...]).  Uses torch.distributed when it is initialised
    (backend nccl => RCCL, tensors on `device`; gloo => CPU tensors); with a
    single process it just adds the words."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(pairs, dtype=np.float64)
    if dist.is_available() and dist.is_initialized():      # also with one rank: the collective path is then exercised
        t = torch.from_numpy(a.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        a = t.cpu().numpy()
    hi, lo = a[..., 0].astype(np.longdouble), a[..., 1].astype(np.longdouble)
    return (hi + lo).astype(np.float64)


def sharded_sum(local_fn, S, rank, world, device=None, shape=None):
    """Evaluate local_fn(s0, s1) -> [..., 2] double-double partial sums on this rank's
    block of sites and all-reduce them.  A rank whose block is empty (more ranks than
    sites) contributes zeros: it learns the shape of the partials from the other ranks
    (one MAX all-reduce of the dimensions), or from `shape` when given."""
    import torch
    import torch.distributed as dist
    s0, s1 = shard_range(S, rank, world)
    part = np.asarray(local_fn(s0, s1), dtype=np.float64) if s1 > s0 else None
    if part is None and shape is not None:
        part = np.zeros(tuple(shape), dtype=np.float64)
    if dist.is_available() and dist.is_initialized() and shape is None:
        dims = torch.zeros(8, dtype=torch.int64)
        if part is not None:
            dims[0] = part.ndim
            dims[1:1 + part.ndim] = torch.tensor(part.shape, dtype=torch.int64)
        if device is not None:
            dims = dims.to(device)
        dist.all_reduce(dims, op=dist.ReduceOp.MAX)
        dims = dims.cpu().tolist()
        if part is None:
            part = np.zeros(tuple(int(d) for d in dims[1:1 + int(dims[0])]), dtype=np.float64)
    if part is None:
        raise ValueError("sharded_sum: no rank has sites and no shape was given (S=%d)" % S)
    return allreduce_dd(part, device)
