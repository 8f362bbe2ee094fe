"""GPU parity: every match algorithm vs the string-based CPU oracle, bit-exact."""
import numpy as np
import pytest

from barcoder_amd import DeviceLibrary, GpuMatcher
from barcoder_amd import encoding as E
from barcoder_amd import synth as S

pytestmark = pytest.mark.gpu


def _case(G, n, k, mixture, seed, frac_skip=0.02):
    lib = S.library(G, k, seed=S.LIB_SEED + k)
    reads = S.reads(lib, k, seed, 0, n, mixture)
    flags = (S.draw(seed + 99, np.arange(n, dtype=np.uint64), 7) % np.uint64(1000) < np.uint64(int(frac_skip * 1000))).astype(np.uint8)
    return lib, reads, flags


def _check(res, ref):
    np.testing.assert_array_equal(res.best_idx, ref["best_idx"])
    np.testing.assert_array_equal(res.best_dist, ref["best_dist"])
    np.testing.assert_array_equal(res.n_best, ref["n_best"])
    np.testing.assert_array_equal(res.counts, ref["counts"])
    np.testing.assert_array_equal(res.first_seen, ref["first_seen"])


@pytest.mark.parametrize("k", [20, 32, 12])
@pytest.mark.parametrize("algo,max_mm", [("hash", 0), ("seed", 0), ("seed", 1), ("seed", 2), ("seed", 3),
                                         ("seed_csr", 0), ("seed_csr", 2), ("seed_csr", 3),
                                         ("brute", 0), ("brute", 2), ("brute", 4)])
def test_match_parity(gpu_device, oracle, algo, max_mm, k):
    G, n = 3000, 20011
    lib, reads, flags = _case(G, n, k, 1 if max_mm else 0, seed=7 + max_mm)
    dl = DeviceLibrary(lib, k, gpu_device)
    res = GpuMatcher(dl).match(reads, flags, max_mismatch=max_mm, algo=algo, first_index=5)
    ref = oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k), max_mm, skip=flags, base=5)
    _check(res, ref)
    assert (res.best_idx >= 0).sum() > n // 2


@pytest.mark.parametrize("env", [{"BC_TALLY": "1"}, {"BC_TALLY": "2"}, {"BC_SLAB_LPR": "8", "BC_SLAB_MIN_LOG2S": "4", "BC_TWO_STAGE": "0"},
                                 {"BC_SLAB_LOG2S": "3,3,3,3"}, {"BC_SLAB_LOG2S": "5,5,5,5"}, {"BC_SLAB_LOG2S": "4,3,5,4"},
                                 {"BC_TWO_STAGE": "0"}, {"BC_TWO_STAGE": "2"}, {"BC_TWO_STAGE": "2", "BC_SLAB_LPR": "8", "BC_SLAB_MIN_LOG2S": "4"},
                                 {"BC_HASH_SLOTS_PCT": "120"}, {"BC_HASH_SLOTS_PCT": "3000"}])
def test_tuning_knobs_do_not_change_results(gpu_device, oracle, env, monkeypatch):
    """Tally by global atomics vs LDS histogram, fused vs two-kernel vs single
    slab kernel, 4 vs 8 lanes per read, slab sizes with and without overflow
    chains, hash tables from crowded to sparse: identical outputs."""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    k, G, n = 20, 6000, 30001
    lib, reads, flags = _case(G, n, k, 1, seed=31)
    dl = DeviceLibrary(lib, k, gpu_device)
    ls, rs = E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k)
    for d in (0, 1, 2, 3):
        res = GpuMatcher(dl).match(reads, flags, max_mismatch=d, algo="seed", first_index=11)
        _check(res, oracle.best_hit(ls, rs, d, skip=flags, base=11))


def test_dense_library_ties(gpu_device, oracle):
    """A library that is dense in sequence space: many ties and n_best > 1."""
    k = 6
    lib = np.arange(4 ** k, dtype=np.uint64)
    # plane form of the 2-bit counter value: spread even/odd bits
    codes = (lib[:, None] >> (np.arange(k, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)
    lo = ((codes & np.uint64(1)) << np.arange(k, dtype=np.uint64)).sum(1)
    hi = ((codes >> np.uint64(1)) << np.arange(k, dtype=np.uint64)).sum(1)
    allk = (lo | (hi << np.uint64(32))).astype(np.uint64)
    rng = np.random.default_rng(3)
    lib = allk[rng.permutation(allk.size)[:1500]]
    reads = allk[rng.integers(0, allk.size, 5000)]
    dl = DeviceLibrary(lib, k, gpu_device)
    m = GpuMatcher(dl)
    ls, rs = E.unpack_kmers(lib, k), E.unpack_kmers(reads, k)
    for algo, d in [("seed", 1), ("seed", 2), ("seed", 3), ("seed_csr", 2), ("brute", 2), ("hash", 0)]:
        res = m.match(reads, None, max_mismatch=d, algo=algo)
        _check(res, oracle.best_hit(ls, rs, d))
    assert res is not None


def _mutate(rng, codes, nsub):
    """codes: (n, k) array of 0..3; substitute nsub[i] distinct positions of row i by a different base"""
    out = codes.copy()
    n, k = codes.shape
    for i in range(n):
        pos = rng.choice(k, size=nsub[i], replace=False)
        out[i, pos] = (out[i, pos] + rng.integers(1, 4, size=nsub[i])) % 4
    return out


@pytest.mark.parametrize("slab", ["3,3,3", "3,3,4", "3,3,5"])
@pytest.mark.parametrize("cluster", [0, 3, 12])
def test_wave_kernel_paths(gpu_device, oracle, slab, cluster, monkeypatch):
    """The default d = 1, 2 kernel (k_match_wave) on every table-2 slab size, with libraries from isolated guides
    to tight clusters: isolation flags, tie counting from neighbour lists (cluster = 3: lists of <= 8), the quad
    exact path (cluster = 12: more than 8 close neighbours), overflow chains, skip flags, a ragged last tile —
    against the oracle and against the quad kernels (BC_LANE=0)."""
    k, n = 20, 10007
    rng = np.random.default_rng(17 + cluster)
    if cluster:
        centres = rng.integers(0, 4, size=(400, k))
        fam = np.repeat(centres, cluster, axis=0)
        fam = _mutate(rng, fam, rng.integers(0, 3, size=fam.shape[0]))
        fam = np.unique(fam, axis=0)
        codes = fam[rng.permutation(fam.shape[0])]
    else:
        codes = np.unique(rng.integers(0, 4, size=(6000, k)), axis=0)
        codes = codes[rng.permutation(codes.shape[0])]
    G = codes.shape[0]
    letters = np.array(list("ACGT"))
    lib_s = ["".join(letters[r]) for r in codes]
    src = rng.integers(0, G, size=n)
    nsub = rng.choice([0, 1, 2, 3], size=n, p=[0.4, 0.3, 0.2, 0.1])
    rd = _mutate(rng, codes[src], nsub)
    rd[rng.random(n) < 0.05] = rng.integers(0, 4, size=k)
    reads_s = ["".join(letters[r]) for r in rd]
    lib, _ = E.pack_kmers(lib_s, k)
    reads, _ = E.pack_kmers(reads_s, k)
    flags = (rng.random(n) < 0.02).astype(np.uint8)
    monkeypatch.setenv("BC_SLAB_LOG2S", slab)
    dl = DeviceLibrary(lib, k, gpu_device)
    for d in (1, 2):
        ref = oracle.best_hit(lib_s, reads_s, d, skip=flags, base=3)
        monkeypatch.setenv("BC_LANE", "1")
        _check(GpuMatcher(dl).match(reads, flags, max_mismatch=d, algo="seed", first_index=3), ref)
        monkeypatch.setenv("BC_LANE", "0")
        _check(GpuMatcher(dl).match(reads, flags, max_mismatch=d, algo="seed", first_index=3), ref)
    if cluster:
        assert (ref["n_best"] > 1).sum() > 50


@pytest.mark.parametrize("k,G", [(12, 5000), (16, 20000), (18, 60000), (20, 20000), (20, 150000), (22, 60000)])
def test_wave_kernel_geometries_at_scale(gpu_device, oracle, k, G):
    """Default d = 1, 2 path over k-mer lengths and library sizes (different seed / rest widths, bucket loads from
    1 to 16 per slab, neighbour densities from none to dense), 300 k reads each, every per-read output against the
    indexed oracle; counts and first-seen against numpy."""
    lib = S.library(G, k, seed=S.LIB_SEED + 7 * k + G)
    n = 300_007
    reads = S.reads(lib, k, 11 + k, 0, n, 1)
    sh = np.arange(k, dtype=np.uint64)
    letters = np.frombuffer(b"ACGT", np.uint8)

    def ascii_of(p):
        q = np.ascontiguousarray(p, np.uint64)[:, None]
        return letters[((q >> sh) & np.uint64(1)).astype(np.uint8) | (((q >> (sh + np.uint64(32))) & np.uint64(1)).astype(np.uint8) << 1)]

    la, ra = ascii_of(lib), ascii_of(reads & E.kmer_mask(k))
    dl = DeviceLibrary(lib, k, gpu_device)
    for d in (1, 2):
        res = GpuMatcher(dl).match(reads, None, max_mismatch=d, algo="seed", first_index=0)
        ref = oracle.best_hit_arrays(la, ra, d, indexed=True)
        _check(res, ref)


def test_tally_hot_guides(gpu_device):
    """Per-guide counts in the millions through the LDS histogram (k_hist), first-seen intact."""
    k, G, n = 20, 300, 3_000_003
    lib = S.library(G, k)
    rng = np.random.default_rng(5)
    pick = rng.choice(G, size=n, p=np.r_[0.6, 0.3, np.full(G - 2, 0.1 / (G - 2))])
    pick[:5] = [17, 17, 3, 0, 1]
    reads = lib[pick].copy()
    miss = rng.random(n) < 0.01
    reads[miss] ^= np.uint64(0x5A5A5)  # a few non-matching reads
    dl = DeviceLibrary(lib, k, gpu_device)
    res = GpuMatcher(dl).match(reads, None, max_mismatch=0, algo="hash", first_index=100)
    hit = res.best_idx >= 0
    np.testing.assert_array_equal(res.best_idx[~miss], pick[~miss])
    np.testing.assert_array_equal(res.counts, np.bincount(res.best_idx[hit], minlength=G))
    assert res.counts.max() > 1_000_000
    first = np.full(G, np.iinfo(np.uint64).max, np.uint64)
    idx_hit = np.flatnonzero(hit)
    order = np.argsort(res.best_idx[idx_hit], kind="stable")
    g_sorted = res.best_idx[idx_hit][order]
    starts = np.r_[0, np.flatnonzero(np.diff(g_sorted)) + 1]
    first[g_sorted[starts]] = idx_hit[order][starts].astype(np.uint64) + np.uint64(100)
    np.testing.assert_array_equal(res.first_seen.astype(np.uint64), first)


def test_all_hits(gpu_device, oracle):
    k = 20
    lib, reads, flags = _case(2000, 5000, k, 1, seed=21)
    dl = DeviceLibrary(lib, k, gpu_device)
    for d in (0, 1, 2, 3):
        hits = GpuMatcher(dl).all_hits(reads, flags, max_mismatch=d)
        ref = oracle.all_hits(E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k), d, skip=flags)
        got = np.stack([hits["query"], hits["guide"], hits["dist"]], axis=1)
        np.testing.assert_array_equal(got, ref)


def test_edge_sizes(gpu_device, oracle):
    k = 20
    lib = S.library(50, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    m = GpuMatcher(dl)
    for n in (0, 1, 3, 4, 5, 255, 257):
        reads = S.reads(lib, k, 3, 0, n, 1)
        for algo, d in [("hash", 0), ("seed", 2), ("brute", 2)]:
            res = m.match(reads, None, max_mismatch=d, algo=algo)
            if n:
                _check(res, oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads, k), d))
            else:
                assert res.best_idx.size == 0 and res.counts.sum() == 0


def test_synth_device_matches_numpy(gpu_device):
    import ctypes as C
    from barcoder_amd import _native as N
    k, n = 20, 10007
    lib = S.library(1234, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    L = N.load()
    d = N.vp()
    N.check(L.bc_dev_alloc(C.byref(d), n * 8))
    try:
        for mix in (0, 1):
            GpuMatcher(dl).synth_reads_dev(seed=2 + mix, first=1000, n=n, mixture=mix, d_reads=d)
            out = np.empty(n, np.uint64)
            N.check(L.bc_dev_d2h(N.ptr(out), d, n * 8, None))
            np.testing.assert_array_equal(out, S.reads(lib, k, 2 + mix, 1000, n, mix))
    finally:
        L.bc_dev_free(d)


def test_large_library_and_misaligned_device_pointers(gpu_device):
    """2e6 guides (longer seeds, bigger slabs) and device pointers that are not
    16-byte aligned (vector paths must fall back): fast kernels == brute force."""
    import torch
    k, G, n = 20, 2_000_000, 60_001
    lib = S.library(G, k)
    reads = S.reads(lib, k, 13, 0, n + 1, 1)
    dl = DeviceLibrary(lib, k, gpu_device)
    m = GpuMatcher(dl)
    for d in (0, 2):
        ref = m.match(reads[:n], None, max_mismatch=d, algo="brute")
        got = m.match(reads[:n], None, max_mismatch=d, algo="seed" if d else "hash")
        _check(got, {"best_idx": ref.best_idx, "best_dist": ref.best_dist, "n_best": ref.n_best,
                     "counts": ref.counts, "first_seen": ref.first_seen})
    # device path with every buffer offset by one element
    dev = torch.device("cuda", gpu_device)
    tr = torch.from_numpy(reads.view(np.int64)).to(dev)
    bi = torch.empty(n + 1, dtype=torch.int32, device=dev)
    bd = torch.empty(n + 1, dtype=torch.uint8, device=dev)
    nb = torch.empty(n + 1, dtype=torch.uint8, device=dev)
    for d, algo in ((2, "seed"), (0, "hash")):
        ref = m.match(reads[1:], None, max_mismatch=d, algo="brute")
        m.match_dev(tr[1:].data_ptr(), n, bi[1:].data_ptr(), max_mismatch=d, algo=algo, d_best_dist=bd[1:].data_ptr(),
                    d_n_best=nb[1:].data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(bi[1:].cpu().numpy(), ref.best_idx)
        np.testing.assert_array_equal(bd[1:].cpu().numpy(), ref.best_dist)
        np.testing.assert_array_equal(nb[1:].cpu().numpy(), ref.n_best)
