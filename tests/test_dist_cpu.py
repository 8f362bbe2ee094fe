"""CPU (gloo, world_size 2): the N > 1 path — contiguous read shards, one SUM
all-reduce of the count vector, one MIN all-reduce of first-seen indices.
The per-shard match is supplied by the oracle here (no GPU in this container);
on the GPU box bench.py / GpuMatcher supply it."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, G, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from barcoder_amd import dist as D, encoding as E, synth as S
    from oracle import oracle as orc
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lib = S.library(G, 20)
    ls = E.unpack_kmers(lib, 20)

    def match_shard(a, b):
        reads = S.reads(lib, 20, 5, a, b - a, 1)
        res = orc.best_hit(ls, E.unpack_kmers(reads, 20), 2, base=a)
        return res["counts"], res["first_seen"]

    counts, first = D.count_sharded(n_total, match_shard, G, r, w)
    # the same reduction on torch tensors, started asynchronously and awaited (bench.py runs it under the next pass)
    import torch
    a, b = D.shard_range(n_total, r, w)
    if b > a:
        lc, lf = match_shard(a, b)
    else:
        lc, lf = np.zeros(G, np.uint64), np.full(G, D.U64_MAX, np.uint64)
    tc = torch.from_numpy(np.asarray(lc).astype(np.int64))
    tf = torch.from_numpy(np.minimum(np.asarray(lf, np.uint64), np.uint64(D.I64_MAX)).astype(np.int64))
    tc, tf, pend = D.reduce_counts(tc, tf, async_op=True)
    assert len(pend.works) == 2
    pend.wait()
    assert np.array_equal(tc.numpy(), np.asarray(counts).astype(np.int64))
    assert np.array_equal(tf.numpy(), np.asarray(first).astype(np.int64))
    # async_op on numpy results (what an empty shard yields): still three values, nothing pending
    ac, af, ap = D.count_sharded(n_total, match_shard, G, r, w, async_op=True)
    assert ap.works == [] and np.array_equal(np.asarray(ac), np.asarray(counts)) and np.array_equal(np.asarray(af), np.asarray(first))
    # undocumented tables of unequal lengths (one of them empty): padded tensors, every rank sees every rank's table
    mine = tuple(np.arange(rank * 5, dtype=np.uint64) + np.uint64(j << 60) for j in range(3))
    got = D.gather_tables(mine)
    assert len(got) == world
    for rr in range(world):
        for j in range(3):
            assert np.array_equal(got[rr][j], np.arange(rr * 5, dtype=np.uint64) + np.uint64(j << 60))
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), counts)
    np.save(os.path.join(out_dir, f"first_{rank}.npy"), first)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5001, 3])
def test_sharded_count_equals_single_process(tmp_path, oracle, n_total):
    import torch.multiprocessing as mp
    from barcoder_amd import dist as D, encoding as E, synth as S
    G, world = 300, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, G, str(tmp_path)), nprocs=world, join=True)
    lib = S.library(G, 20)
    reads = S.reads(lib, 20, 5, 0, n_total, 1)
    ref = oracle.best_hit(E.unpack_kmers(lib, 20), E.unpack_kmers(reads, 20), 2)
    want_first = np.minimum(ref["first_seen"], np.uint64(D.I64_MAX)).astype(np.int64)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"counts_{rank}.npy"), ref["counts"].astype(np.int64))
        np.testing.assert_array_equal(np.load(tmp_path / f"first_{rank}.npy"), want_first)


def test_shard_ranges_cover_everything():
    from barcoder_amd.dist import shard_range
    for n in (0, 1, 7, 8, 9, 10**9 + 7):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) <= -(-n // w)
