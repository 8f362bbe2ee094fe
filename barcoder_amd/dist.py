"""Multi-GPU counting: reads shard over ranks, one all-reduce of the per-guide
count vector (and one MIN all-reduce of first-seen indices) at the end.

Replaces the Pool fan-out and Counter merge of heuristicount.py:703-737.  One
process per GPU; torch.distributed's "nccl" backend is RCCL over xGMI on ROCm,
"gloo" is used by the CPU tests (and by BC_DIST_BACKEND=gloo, which lets one GPU
rehearse the N > 1 path).  There is no data-path collective besides the final
reduction: shards are independent.

Product entry points that use this module: bench.py (`--gpus N`) and
`python -m barcoder_amd.heuristicount --gpus N`.  Both start their own ranks with
`launch_workers` when no launcher (torchrun) has exported WORLD_SIZE; the parent
process never touches the GPU, the children are fresh interpreters.

Importing this module does not import torch or the native library."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np

I64_MAX = np.iinfo(np.int64).max
U64_MAX = np.iinfo(np.uint64).max


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous range [a, b) of ceil(n/world)-sized shards (last may be short)."""
    per = -(-n_total // world)
    a = min(n_total, rank * per)
    return a, min(n_total, a + per)


def under_launcher() -> bool:
    """True when a launcher (torchrun or launch_workers) has described the process group."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


DEFAULT_LAUNCH_TIMEOUT = 24 * 3600.0  # seconds; BC_LAUNCH_TIMEOUT overrides


def launch_workers(argv, nproc: int, env=None, timeout=-1) -> int:
    """Start `nproc` ranks of `python argv...` on this node (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment) and wait for
    them.  Rank 0 inherits stdout; the other ranks' stdout goes to stderr, so a
    caller that prints one result line on rank 0 keeps a clean stdout.  Returns
    the first non-zero exit status (the remaining ranks are terminated), else 0.

    timeout (seconds; default BC_LAUNCH_TIMEOUT or 24 h — always finite): a rank that
    HANGS (in RCCL initialisation, in a collective a peer never joins) does not fail, so
    nothing else would end the job: when the time is up every rank still alive is
    terminated, killed five seconds later if it ignores that, and the status is 124.

    The calling process must not have initialised the GPU: the children are
    fresh interpreters (no fork of a HIP context, no exec from a GPU process)."""
    if nproc < 1:
        raise ValueError("launch_workers: nproc must be >= 1")
    if timeout is None or timeout < 0:
        timeout = float(os.environ.get("BC_LAUNCH_TIMEOUT", DEFAULT_LAUNCH_TIMEOUT))
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(nproc)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    procs = []
    try:
        for r in range(nproc):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            out = None if r == 0 else sys.stderr
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=out))
        rc = 0
        import time
        t0 = time.monotonic()
        kill_at = None
        alive = list(procs)
        while alive:
            for p in list(alive):
                st = p.poll()
                if st is None:
                    continue
                alive.remove(p)
                if st != 0 and rc == 0:
                    rc = st
                    for q in alive:  # a failed rank would leave the others waiting in a collective
                        q.terminate()
            if alive:
                now = time.monotonic()
                if kill_at is None and now - t0 > timeout:
                    rc = rc or 124
                    print(f"launch_workers: {len(alive)} of {nproc} ranks still running after {timeout:.0f} s: terminating them",
                          file=sys.stderr)
                    for q in alive:
                        q.terminate()
                    kill_at = now + 5.0
                elif kill_at is not None and now > kill_at:
                    for q in alive:
                        q.kill()
                time.sleep(0.05)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


class stdout_to_stderr:
    """Send file descriptor 1 to stderr for the duration (native libraries — gloo's "[Gloo] Rank 0 is
    connected ..." — print there), so that a rank's stdout carries only what it prints afterwards."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def init_from_env(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun
    or launch_workers).  Returns (rank, world, local device index)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        # fewer devices than ranks (BC_DIST_BACKEND=gloo rehearsal on one GPU): ranks share devices
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = os.environ.get("BC_DIST_BACKEND") or ("nccl" if have_gpu else "gloo")
        if backend == "nccl" and have_gpu and torch.cuda.device_count() < world:
            raise RuntimeError(f"{world} ranks over RCCL need {world} GPUs, {torch.cuda.device_count()} visible "
                               "(set BC_DIST_BACKEND=gloo to rehearse on fewer)")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _group_backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


class Pending:
    """Collectives started by reduce_counts(..., async_op=True): `wait()` before the tensors are read or reused
    (on RCCL that orders the caller's stream behind the collective, it does not block the host)."""
    def __init__(self, works=()):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def reduce_counts(counts, first_seen=None, device=None, async_op=False):
    """All-reduce SUM of int64 counts[G] and MIN of int64 first_seen[G] (I64_MAX
    = unseen) across the group.  numpy arrays in -> numpy arrays out on every
    rank; torch tensors are reduced in place on their device (staged through the
    host when the group's backend is gloo and the tensor lives on a GPU).

    async_op (torch tensors only): start both collectives and return
    (counts, first_seen, Pending) so that the next shard's matching overlaps them
    (RCCL runs them on its own stream); the caller waits before touching the tensors."""
    import torch
    import torch.distributed as dist
    is_np = isinstance(counts, np.ndarray)
    if is_np:
        c = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64))
        f = None if first_seen is None else torch.from_numpy(
            np.minimum(np.ascontiguousarray(first_seen).astype(np.uint64), np.uint64(I64_MAX)).astype(np.int64))
        if device is not None and _group_backend() == "nccl":  # RCCL reduces device tensors
            device = torch.device("cuda", device) if isinstance(device, int) else device
            c = c.to(device)
            f = None if f is None else f.to(device)
    else:
        c, f = counts, first_seen
    pending = Pending()
    # (a group of ONE still runs its collectives: a one-rank "nccl" group is how the RCCL branch is exercised on a
    # one-GPU box, tests/test_dist_gpu.py; without a group nothing is reduced)
    if dist.is_available() and dist.is_initialized():
        gloo_on_gpu = _group_backend() == "gloo" and c.is_cuda
        for t, op in ((c, dist.ReduceOp.SUM), (f, dist.ReduceOp.MIN)):
            if t is None:
                continue
            if gloo_on_gpu:
                h = t.cpu()
                dist.all_reduce(h, op=op)
                t.copy_(h)
            elif async_op and not is_np:
                pending.works.append(dist.all_reduce(t, op=op, async_op=True))
            else:
                dist.all_reduce(t, op=op)
    if is_np:
        return c.cpu().numpy(), (None if f is None else f.cpu().numpy())
    if async_op:
        return c, f, pending
    return c, f


def reduce_scalar_max(x: float, device=None) -> float:
    """MAX over ranks of one float (bench timing)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return float(x)
    dev = device if (device is not None and _group_backend() == "nccl") else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_objects(obj):
    """Every rank's picklable `obj` as a list on every rank (small, variable-length results:
    undocumented k-mer tables, odd string events)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def count_sharded(n_total: int, match_shard, n_guides: int, rank: int, world: int, device=None, async_op=False):
    """Run `match_shard(a, b) -> (counts[G], first_seen[G])` on this rank's read
    range and combine over ranks.  `first_seen` carries GLOBAL read indices, so
    MIN over ranks is the job-wide first occurrence.  numpy arrays (uint64,
    U64_MAX = unseen) or torch int64 tensors (reduced in place)."""
    a, b = shard_range(n_total, rank, world)
    if b > a:
        counts, first_seen = match_shard(a, b)
    else:
        counts = np.zeros(n_guides, np.uint64)
        first_seen = np.full(n_guides, U64_MAX, np.uint64)
    if async_op and isinstance(counts, np.ndarray):  # numpy results (an empty shard) reduce synchronously:
        c, f = reduce_counts(counts, first_seen, device=device)  # the caller still gets the three values it asked for
        return c, f, Pending()
    return reduce_counts(counts, first_seen, device=device, async_op=async_op)


def gather_tables(arrays, device=None):
    """Every rank's tuple of equally long uint64 arrays (an undocumented table: words, counts, first keys) as a list of
    tuples on every rank: one size exchange, then one all_gather of a padded int64 tensor per array — on the device
    for RCCL, on the host for gloo — instead of pickling hundreds of MB through all_gather_object."""
    import torch
    import torch.distributed as dist
    arrays = tuple(np.ascontiguousarray(a, np.uint64) for a in arrays)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [arrays]
    world = dist.get_world_size()
    dev = "cpu"
    if device is not None and _group_backend() == "nccl":
        dev = torch.device("cuda", device) if isinstance(device, int) else device
    n = arrays[0].size
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
    sizes = [int(t.item()) for t in sizes]
    cap = max(max(sizes), 1)
    cols = []
    for a in arrays:
        mine = torch.zeros(cap, dtype=torch.int64, device=dev)
        if n:
            mine[:n] = torch.from_numpy(a.view(np.int64)).to(dev)
        parts = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine)
        cols.append([p[:m].cpu().numpy().view(np.uint64) for p, m in zip(parts, sizes)])
    return [tuple(col[r] for col in cols) for r in range(world)]


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
