"""Multi-GPU counting: reads shard over ranks, one all-reduce of the per-guide
count vector (and one MIN all-reduce of first-seen indices) at the end.

Replaces the Pool fan-out and Counter merge of heuristicount.py:703-737.  One
process per GPU; torch.distributed's "nccl" backend is RCCL over xGMI on ROCm,
"gloo" is used by the CPU tests.  There is no data-path collective besides the
final reduction: shards are independent."""
from __future__ import annotations

import os

import numpy as np

I64_MAX = np.iinfo(np.int64).max


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous range [a, b) of ceil(n/world)-sized shards (last may be short)."""
    per = -(-n_total // world)
    a = min(n_total, rank * per)
    return a, min(n_total, a + per)


def init_from_env(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:  # BC_DIST_BACKEND=gloo lets one GPU rehearse the N > 1 path
            backend = os.environ.get("BC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)
    return rank, world, local


def reduce_counts(counts, first_seen=None, device=None):
    """All-reduce SUM of int64 counts[G] and MIN of int64 first_seen[G] (I64_MAX
    = unseen) across the group; returns numpy arrays on every rank.  Accepts
    numpy arrays or torch tensors (tensors are reduced in place on their device)."""
    import torch
    import torch.distributed as dist
    is_np = isinstance(counts, np.ndarray)
    if is_np:
        c = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64))
        f = None if first_seen is None else torch.from_numpy(
            np.minimum(np.ascontiguousarray(first_seen), np.uint64(I64_MAX)).astype(np.int64))
        if device is not None:
            c = c.to(device)
            f = None if f is None else f.to(device)
    else:
        c, f = counts, first_seen
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        if f is not None:
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
    if is_np:
        return c.cpu().numpy(), (None if f is None else f.cpu().numpy())
    return c, f


def count_sharded(n_total: int, match_shard, n_guides: int, rank: int, world: int, device=None):
    """Run `match_shard(a, b) -> (counts uint64[G], first_seen uint64[G])` on this
    rank's read range and combine over ranks.  `first_seen` carries GLOBAL read
    indices, so MIN over ranks is the job-wide first occurrence."""
    a, b = shard_range(n_total, rank, world)
    if b > a:
        counts, first_seen = match_shard(a, b)
    else:
        counts = np.zeros(n_guides, np.uint64)
        first_seen = np.full(n_guides, np.iinfo(np.uint64).max, np.uint64)
    return reduce_counts(counts, first_seen, device=device)
