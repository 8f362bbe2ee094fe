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
    if algo == "seed":  # which kernel took it: the wave matcher for d = 1, 2 at every k (wide entries for k = 32)
        wide = k == 32
        # d = 3 is a cascade: the d <= 2 pass over every read, the 3-mismatch kernels over the reads it left
        want = {0: "k_match_seed" if wide else "k_match_slab", 1: "k_match_wq", 2: "k_match_wq",
                3: "k_match_seed" if wide else "cascade3:d2+k_match_pairs"}[max_mm]
        assert dl.last_kernel == want, (k, max_mm, dl.last_kernel)


@pytest.mark.parametrize("env", [{"BC_TALLY": "1"}, {"BC_TALLY": "2"}, {"BC_TALLY": "3"},
                                 {"BC_SLAB_LPR": "8", "BC_SLAB_MIN_LOG2S": "4", "BC_SINGLE_SLAB": "1"},
                                 {"BC_SLAB_LOG2S": "3,3,3,3"}, {"BC_SLAB_LOG2S": "5,5,5,5"}, {"BC_SLAB_LOG2S": "4,3,5,4"},
                                 {"BC_SINGLE_SLAB": "1"}, {"BC_LANE": "0"}, {"BC_LANE": "0", "BC_PHASE2_SM": "0"},
                                 {"BC_T2_FILTER": "0"}, {"BC_CASCADE": "0"}, {"BC_CASCADE": "2"}, {"BC_PREFETCH": "0"}, {"BC_PREFETCH": "1"}, {"BC_E8": "0"},
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_E8": "0"},
                                 {"BC_HASH_SLOTS_PCT": "120"}, {"BC_HASH_SLOTS_PCT": "3000"},
                                 # the cascades' second stage from a list (large launches take it by default): written by the
                                 # wave matcher's list form, packed, passes over table groups;
                                 # d = 2 as a cascade too (then d = 3 is a cascade over a cascade and gathers its own list)
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_PAIR_GROUP_KB": "64"},
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_PAIR_CHUNK_LOG2": "13"},  # (very long inputs: several launches)
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_CASCADE2_MIN_G": "1"}, {"BC_CASCADE2_MIN_G": "1"},
                                 {"BC_CASCADE2_MIN_G": "1", "BC_PAIR_FOUR": "0"}, {"BC_SEED_WANT_BIAS": "1"}, {"BC_SEED_WANT_BIAS": "-1"},
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_CASCADE2_MIN_G": "1", "BC_PAIR_BUCKET_PCT": "50"},
                                 {"BC_PAIR_PASS_MIN_READS": "0", "BC_CASCADE2_MIN_G": "1", "BC_PAIR_BUCKET_PCT": "6400"}])
def test_tuning_knobs_do_not_change_results(gpu_device, oracle, env, monkeypatch):
    """Tally by global atomics vs LDS histogram vs records, wave vs quad vs single
    slab kernel, 4 vs 8 lanes per read, slab sizes with and without overflow
    chains, hash tables from crowded to sparse: identical outputs.  (The knobs are
    read when the library is created.)"""
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


@pytest.mark.parametrize("k", [20, 32])
@pytest.mark.parametrize("slab", ["3,3,3", "3,3,4", "3,3,5"])
@pytest.mark.parametrize("cluster", [0, 3, 12])
def test_wave_kernel_paths(gpu_device, oracle, slab, cluster, k, monkeypatch):
    """The default d = 1, 2 kernel (k_match_wq) on every table-2 slab size, with libraries from isolated guides
    to tight clusters: isolation flags, tie counting from neighbour lists (cluster = 3: lists of <= 8), the quad
    exact path (cluster = 12: more than 8 close neighbours), overflow chains, skip flags, a ragged last tile —
    against the oracle and against the quad kernels (BC_LANE=0).  k = 32: the wide entry format (two-dword rests,
    128-byte slabs for tables 0/1, the quad exact path slab_slow_wide); BC_LANE=0 then leaves only the CSR kernel."""
    n = 10007
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
    monkeypatch.setenv("BC_LANE", "1")
    dl = DeviceLibrary(lib, k, gpu_device)
    monkeypatch.setenv("BC_LANE", "0")  # knobs are read at creation: a second library for the quad kernels
    dq = DeviceLibrary(lib, k, gpu_device)
    for d in (1, 2):
        ref = oracle.best_hit(lib_s, reads_s, d, skip=flags, base=3)
        _check(GpuMatcher(dl).match(reads, flags, max_mismatch=d, algo="seed", first_index=3), ref)
        assert dl.last_kernel == "k_match_wq"
        _check(GpuMatcher(dq).match(reads, flags, max_mismatch=d, algo="seed", first_index=3), ref)
        assert dq.last_kernel == ("k_match_fused" if k == 20 else "k_match_seed")
    if cluster:
        assert (ref["n_best"] > 1).sum() > 50


@pytest.mark.parametrize("k,G", [(12, 5000), (16, 20000), (18, 60000), (20, 20000), (20, 150000), (22, 60000),
                                 (24, 60000), (28, 30000), (32, 9503), (32, 100000), (20, 600000), (24, 400000)])
def test_wave_kernel_geometries_at_scale(gpu_device, oracle, k, G):
    """Default d = 1, 2 path over k-mer lengths and library sizes (different seed / rest widths, bucket loads from
    1 to 16 per slab, neighbour densities from none to dense), 300 k reads each, every per-read output against the
    indexed oracle; counts and first-seen against numpy."""
    _geometry_case(gpu_device, oracle, k, G, cascade_from=True)


def test_wave_kernel_on_a_large_library_when_the_cascade_is_off(gpu_device, oracle, monkeypatch):
    """From a weighted load of the wave matcher's buckets on (3.6e5 20-mers; never for 24-mers with their three equal
    8-base seeds) d = 2 runs as the cascade by default (d <= 1 pass + four pair tables: faster there);
    BC_CASCADE2_MIN_G=0 keeps the wave matcher on its 128-byte slabs at 37 guides per bucket — the same results."""
    monkeypatch.setenv("BC_CASCADE2_MIN_G", "0")
    _geometry_case(gpu_device, oracle, 20, 600_000, cascade_from=None)


def _geometry_case(gpu_device, oracle, k, G, cascade_from):
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
        # the wave matcher serves 64-byte slabs, 128-byte slabs for 20-mer libraries up to 40 guides per bucket and
        # wide entries above k = 28; short k-mers whose d = 2 seeds leave 16+ guides per bucket, and tables 0 / 1 of
        # unequal slab size (k = 16, d = 2), stay on the quad kernels
        # the default's load rule (launch_match): guides per bucket of the wave matcher's three d = 2 tables
        want = max(4, min(12, int(np.log(max(G // 4, 1)) / np.log(4) + 1e-9)))                      # build_seed_index
        ld = [G / 4.0 ** min(k // 3 + (1 if t < k % 3 else 0), want) for t in range(3)]
        by_load = ld[2] >= 3 * ld[0] and ld[0] + 0.23 * ld[1] + 0.10 * ld[2] >= 36.0
        if d == 2 and cascade_from is not None and by_load:
            assert dl.last_kernel == "cascade2:d1+k_match_pairs", (k, d, dl.last_kernel)
        elif (k, d) not in ((12, 2), (16, 2)):
            assert dl.last_kernel == "k_match_wq", (k, d, dl.last_kernel)
        else:
            assert dl.last_kernel in ("k_match_wq", "k_match_fused", "k_match_slab", "k_match_seed", "cascade2:d1+k_match_pairs")


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


def _tally_from_best_idx(best_idx, G, base):
    hit = best_idx >= 0
    counts = np.bincount(best_idx[hit], minlength=G).astype(np.uint64)
    first = np.full(G, np.iinfo(np.uint64).max, np.uint64)
    idx_hit = np.flatnonzero(hit)
    order = np.argsort(best_idx[idx_hit], kind="stable")
    g_sorted = best_idx[idx_hit][order]
    if g_sorted.size:
        starts = np.r_[0, np.flatnonzero(np.diff(g_sorted)) + 1]
        first[g_sorted[starts]] = idx_hit[order][starts].astype(np.uint64) + np.uint64(base)
    return counts, first


@pytest.mark.parametrize("G", [45_000, 100_000, 163_840])
@pytest.mark.parametrize("skew", ["uniform", "hot_guide", "one_range"])
def test_record_tally(gpu_device, G, skew):
    """The tally from the matcher's range-partitioned records (k_hist_part + k_first_refine): 3, 5 and 8 guide
    ranges; uniform reads, one guide that takes 40 % of them (its range overflows the 64-record line in most tiles:
    overflow list, results taken from best_idx) and reads that all fall into one range; ragged sizes (n % 4, n % 256);
    exact kernel and wave kernel; accumulation over two calls.  Counts / first-seen must equal what best_idx implies
    and what the five-sweep tally (BC_TALLY=2) and global atomics (BC_TALLY=1) give."""
    import os
    import torch
    k, n = 20, 700_003
    lib = S.library(G, k)
    rng = np.random.default_rng(G % 1000 + len(skew))
    if skew == "uniform":
        pick = rng.integers(0, G, size=n)
    elif skew == "hot_guide":
        pick = np.where(rng.random(n) < 0.4, G // 2 + 7, rng.integers(0, G, size=n))
    else:
        pick = rng.integers(G // 3, G // 3 + 9000, size=n)
    reads = lib[pick].copy()
    sub1 = rng.random(n) < 0.3
    reads[sub1] ^= np.uint64(1) << rng.integers(0, k, size=int(sub1.sum())).astype(np.uint64)  # one substitution
    junk = rng.random(n) < 0.05
    reads[junk] = S.draw(77, np.arange(int(junk.sum()), dtype=np.uint64), 0) & E.kmer_mask(k)
    dev = torch.device("cuda", gpu_device)
    tr = torch.from_numpy(reads.view(np.int64)).to(dev)
    results = {}
    for mode in ("3", "2", "1"):
        os.environ["BC_TALLY"] = mode
        try:
            dl = DeviceLibrary(lib, k, gpu_device)
        finally:
            del os.environ["BC_TALLY"]
        m = GpuMatcher(dl)
        for algo, d in (("hash", 0), ("seed", 2)):
            bi = torch.empty(n, dtype=torch.int32, device=dev)
            cnt = torch.zeros(G, dtype=torch.int32, device=dev)
            fs = torch.full((G,), -1, dtype=torch.int64, device=dev)
            # two launches accumulate into the same vectors: [0, n1) then [n1, n)
            n1 = 300_000 if mode != "1" else n
            for a, b in ((0, n1), (n1, n)):
                if b > a:
                    m.match_dev(tr[a:].data_ptr(), b - a, bi[a:].data_ptr(), max_mismatch=d, algo=algo,
                                d_counts=cnt.data_ptr(), d_first_seen=fs.data_ptr(), first_index=1000 + a)
            torch.cuda.synchronize()
            best = bi.cpu().numpy()
            want_c, want_f = _tally_from_best_idx(best, G, 1000)
            np.testing.assert_array_equal(cnt.cpu().numpy().astype(np.uint64), want_c, err_msg=f"{mode} {algo}")
            np.testing.assert_array_equal(fs.cpu().numpy().view(np.uint64), want_f, err_msg=f"{mode} {algo}")
            results[(mode, algo)] = best
            assert dl.last_kernel == ("k_match_hash" if d == 0 else "k_match_wq")
            if d == 0:
                hit = ~junk & ~sub1
                np.testing.assert_array_equal(best[hit], pick[hit])
    for algo in ("hash", "seed"):
        np.testing.assert_array_equal(results[("3", algo)], results[("2", algo)])


def test_max_mismatch_must_be_below_k(gpu_device):
    """k <= max_mismatch breaks the pigeonhole guarantee (no clean segment is left): rejected, not answered wrongly."""
    lib = S.library(50, 3)
    m = GpuMatcher(DeviceLibrary(lib, 3, gpu_device))
    reads = S.reads(lib, 3, 1, 0, 100, 1)
    assert m.match(reads, None, max_mismatch=2, algo="seed").best_idx.size == 100
    for algo in ("seed", "brute", "seed_csr"):
        with pytest.raises(ValueError, match="max_mismatch"):
            m.match(reads, None, max_mismatch=3, algo=algo)


def test_multi_gpu_single_process(gpu_device, oracle):
    """bc_multi_gpu_match: contiguous shards over several library replicas, per-read outputs stitched by range, counts
    summed and first-seen min'ed — equal to one library over all reads.  One device: a group of one (host reduction)
    and a group of three replicas on the same device; with >= 2 GPUs also the RCCL all-reduce over distinct devices."""
    import barcoder_amd
    from barcoder_amd import MultiGpuMatcher
    k, G, n = 20, 3000, 50_001
    lib, reads, flags = _case(G, n, k, 1, seed=5)
    ref = oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k), 2, skip=flags, base=9)
    groups = [[gpu_device], [gpu_device] * 3]
    if barcoder_amd.device_count() >= 2:
        groups.append(list(range(min(barcoder_amd.device_count(), 4))))
    for devs in groups:
        mm = MultiGpuMatcher(lib, k, devs)
        assert mm.n_gpus == len(devs) and mm.uses_rccl == (len(set(devs)) == len(devs) and len(devs) > 1)
        _check(mm.match(reads, flags, max_mismatch=2, algo="seed", first_index=9), ref)
        _check(mm.match(reads[:2], flags[:2], max_mismatch=2, algo="seed", first_index=9),
               oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads[:2] & E.kmer_mask(k), k), 2, skip=flags[:2], base=9))
        mm.close()


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
    # (from 1 024 reads on the wave matcher prefetches a later tile of the read stream: the clamp at the end of the input)
    for n in (0, 1, 3, 4, 5, 255, 257, 1023, 1024, 1025, 1279, 1281, 2047, 2049):
        reads = S.reads(lib, k, 3, 0, n, 1)
        for algo, d in [("hash", 0), ("seed", 2), ("seed", 3), ("brute", 2)]:
            res = m.match(reads, None, max_mismatch=d, algo=algo)
            if n:
                _check(res, oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads, k), d))
            else:
                assert res.best_idx.size == 0 and res.counts.sum() == 0


def test_host_streaming_modes_agree(gpu_device, oracle):
    """bc_match_host over several chunks per worker (two buffer sets each, one chunk in flight while the neighbours are
    copied): reads from pageable memory (staged) and from page-locked memory (DMA source as it is), results into the
    wrapper's page-locked arrays and into plain numpy arrays handed to the C ABI (staged out), with and without skip
    flags, and the counts-only mode — all identical, and equal to the oracle on a prefix.  Counts-only with a per-read
    output is refused."""
    import ctypes as C
    from barcoder_amd import _native as N
    from barcoder_amd.devmem import PinnedBlock
    k, G, n = 20, 4000, 11_300_007
    lib = S.library(G, k)
    reads = np.concatenate([S.reads(lib, k, 9, a, min(4_000_000, n - a), 1) for a in range(0, n, 4_000_000)])
    flags = (S.draw(5, np.arange(n, dtype=np.uint64), 7) % np.uint64(50) == 0).astype(np.uint8)
    dl = DeviceLibrary(lib, k, gpu_device)
    m = GpuMatcher(dl)
    base = m.match(reads, flags, max_mismatch=2, algo="seed", first_index=3)
    ref = oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads[:200_000] & E.kmer_mask(k), k), 2, skip=flags[:200_000], base=3)
    for key in ("best_idx", "best_dist", "n_best"):
        np.testing.assert_array_equal(getattr(base, key)[:200_000], ref[key])
    pin = PinnedBlock(n * 8)
    preads = pin.array(np.uint64, n)
    preads[:] = reads
    for r, f in ((preads, flags), (preads, None), (reads, None)):
        want = base if f is not None else m.match(reads, None, max_mismatch=2, algo="seed", first_index=3)
        got = m.match(r, f, max_mismatch=2, algo="seed", first_index=3)
        cnt = m.match(r, f, max_mismatch=2, algo="seed", first_index=3, per_read=False)
        assert cnt.best_idx is None and cnt.best_dist is None and cnt.n_best is None
        for key in ("best_idx", "best_dist", "n_best", "counts", "first_seen"):
            np.testing.assert_array_equal(getattr(got, key), getattr(want, key))
        np.testing.assert_array_equal(cnt.counts, want.counts)
        np.testing.assert_array_equal(cnt.first_seen, want.first_seen)
    # plain numpy result arrays through the C ABI: staged out of the workers' page-locked buffers
    bi, bd, nb = np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(n, np.uint8)
    c, fs = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
    L = N.load()
    N.check(L.bc_match_host(dl.handle, N.ptr(reads), N.ptr(flags), n, 2, N.ALGOS["seed"], N.ptr(bi), N.ptr(bd), N.ptr(nb),
                            N.ptr(c), N.ptr(fs), 3))
    np.testing.assert_array_equal(bi, base.best_idx)
    np.testing.assert_array_equal(bd, base.best_dist)
    np.testing.assert_array_equal(nb, base.n_best)
    np.testing.assert_array_equal(c, base.counts)
    with pytest.raises(ValueError, match="counts-only"):
        N.check(L.bc_match_host(dl.handle, N.ptr(reads), None, n, 2, N.ALGOS["seed"], None, N.ptr(bd), None, N.ptr(c), N.ptr(fs), 0))
    with pytest.raises(ValueError, match="counts-only"):
        N.check(L.bc_match_host(dl.handle, N.ptr(reads), None, n, 2, N.ALGOS["seed"], None, None, None, None, None, 0))
    probe = GpuMatcher.pcie_probe(gpu_device, 32 << 20, 2)
    assert probe["h2d_GBps"] > 1 and probe["d2h_GBps"] > 1 and probe["duplex_GBps"] > max(probe["h2d_GBps"], probe["d2h_GBps"]) * 0.8


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
        for mix in (0, 1, 105, 150):  # 100 + J: the read-mix sweep's streams (J % unmatched reads)
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


def test_tally_export_and_reset(gpu_device):
    """bc_tally_export_dev: 64-bit sums, INT64_MAX for guides never seen, accumulators reset for the next pass."""
    import torch
    k, G, n = 20, 5000, 200_000
    lib = S.library(G, k)
    reads = S.reads(lib, k, 3, 0, n, 1)
    dl = DeviceLibrary(lib[:G - 7], k, gpu_device)  # the last 7 guides are absent: their reads match nothing
    m = GpuMatcher(dl)
    want = m.match(reads, None, max_mismatch=1, algo="seed", first_index=11)
    dev = torch.device("cuda", gpu_device)
    Gd = G - 7
    d_reads = torch.from_numpy(reads.view(np.int64)).to(dev)
    bi = torch.empty(n, dtype=torch.int32, device=dev)
    c32 = torch.zeros(Gd, dtype=torch.int32, device=dev)
    fs = torch.full((Gd,), -1, dtype=torch.int64, device=dev)
    c64 = torch.empty(Gd, dtype=torch.int64, device=dev)
    fmin = torch.empty(Gd, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):  # the second pass starts from the reset accumulators
        m.match_dev(d_reads.data_ptr(), n, bi.data_ptr(), max_mismatch=1, algo="seed", d_counts=c32.data_ptr(),
                    d_first_seen=fs.data_ptr(), first_index=11, stream=st)
        m.tally_export_dev(c32.data_ptr(), fs.data_ptr(), c64.data_ptr(), fmin.data_ptr(), reset=True, stream=st)
        torch.cuda.synchronize(dev)
        np.testing.assert_array_equal(c64.cpu().numpy(), want.counts.astype(np.int64))
        f = fmin.cpu().numpy()
        seen = want.counts > 0
        np.testing.assert_array_equal(f[seen], want.first_seen[seen].astype(np.int64))
        assert (f[~seen] == np.iinfo(np.int64).max).all()
        assert int(c32.sum().item()) == 0 and bool((fs == -1).all().item())


@pytest.mark.parametrize("k", [8, 11, 16, 20, 25])
@pytest.mark.parametrize("cluster", [0, 6])
def test_three_mismatch_pair_index(gpu_device, oracle, k, cluster, monkeypatch):
    """max_mismatch = 3 through the pair index (bc_match_pairs.hip): as the second stage of the cascade (default)
    and over every read (BC_CASCADE=2), against the oracle and the four-segment kernels (BC_CASCADE=0).  Clustered
    libraries put guides with three, four and five clean pieces next to each other: every one of them must be counted
    once (first clean pair), ties at distance 1..3 included; reads with skip flags and a ragged launch size."""
    n = 12011
    rng = np.random.default_rng(5 + k + cluster)
    if cluster:
        centres = rng.integers(0, 4, size=(300, k))
        fam = np.repeat(centres, cluster, axis=0)
        fam = _mutate(rng, fam, rng.integers(0, 3, size=fam.shape[0]))
        codes = np.unique(fam, axis=0)
    else:
        codes = np.unique(rng.integers(0, 4, size=(3000, k)), axis=0)
    codes = codes[rng.permutation(codes.shape[0])]
    G = codes.shape[0]
    letters = np.array(list("ACGT"))
    lib_s = ["".join(letters[r]) for r in codes]
    nsub = rng.choice([0, 1, 2, 3, 4], size=n, p=[0.2, 0.2, 0.2, 0.3, 0.1])
    rd = _mutate(rng, codes[rng.integers(0, G, size=n)], nsub)
    rd[rng.random(n) < 0.05] = rng.integers(0, 4, size=k)
    reads_s = ["".join(letters[r]) for r in rd]
    lib, _ = E.pack_kmers(lib_s, k)
    reads, _ = E.pack_kmers(reads_s, k)
    flags = (rng.random(n) < 0.02).astype(np.uint8)
    ref = oracle.best_hit(lib_s, reads_s, 3, skip=flags, base=9)
    if k >= 16:  # (short k-mers: the library is dense, nearly every read has a guide within two)
        assert (ref["best_dist"] == 3).sum() > 500
        if cluster:
            assert ((ref["n_best"] > 1) & (ref["best_dist"] == 3)).sum() > 50
    # ("1p", "1g": the second stage as passes over table groups — one table per pass, a few per pass — which large
    # launches on large libraries take by default; the state carried between passes must give the same ties)
    for mode, want in (("1", "cascade3:d2+k_match_pairs"), ("1p", "cascade3:d2+k_match_pairs"), ("1g", "cascade3:d2+k_match_pairs"),
                       ("2", "k_match_pairs"), ("0", None)):
        monkeypatch.setenv("BC_CASCADE", mode[0])
        monkeypatch.setenv("BC_PAIR_PASS_MIN_READS", "0" if len(mode) > 1 else "65536")
        monkeypatch.setenv("BC_PAIR_GROUP_KB", {"1p": "1", "1g": "200"}.get(mode, "3300"))
        dl = DeviceLibrary(lib, k, gpu_device)
        _check(GpuMatcher(dl).match(reads, flags, max_mismatch=3, algo="seed", first_index=9), ref)
        if want:  # (a dense library of short k-mers may leave the second stage nothing to do)
            assert dl.last_kernel == want, (mode, dl.last_kernel)


@pytest.mark.parametrize("which", ["seed0", "seed1", "seed2"])
def test_bucket_size_boundaries(gpu_device, oracle, which):
    """Buckets of exactly 1 ... 40 guides in one table (guides that share one segment's seed): the slab forms' edges —
    7 / 8 entries of a primary slab, 15 of primary + secondary, chained slabs behind them — for the wave matcher
    (eight-entry primaries for tables 0/1, 16-entry slabs for table 2) and the d = 3 cascade, against the oracle."""
    k = 20
    rng = np.random.default_rng({"seed0": 1, "seed1": 2, "seed2": 3}[which])
    lo, hi = {"seed0": (0, 7), "seed1": (7, 14), "seed2": (14, 20)}[which]
    rows = []
    for size in list(range(1, 41)) * 3:
        shared = rng.integers(0, 4, size=hi - lo)
        grp = rng.integers(0, 4, size=(size, k))
        grp[:, lo:hi] = shared
        rows.append(grp)
    codes = np.unique(np.concatenate(rows + [rng.integers(0, 4, size=(2000, k))]), axis=0)
    codes = codes[rng.permutation(codes.shape[0])]
    G = codes.shape[0]
    letters = np.array(list("ACGT"))
    lib_s = ["".join(letters[r]) for r in codes]
    n = 30011
    nsub = rng.choice([0, 1, 2, 3], size=n, p=[0.3, 0.3, 0.3, 0.1])
    rd = _mutate(rng, codes[rng.integers(0, G, size=n)], nsub)
    rd[rng.random(n) < 0.05] = rng.integers(0, 4, size=k)
    reads_s = ["".join(letters[r]) for r in rd]
    lib, _ = E.pack_kmers(lib_s, k)
    reads, _ = E.pack_kmers(reads_s, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    for d in (1, 2, 3):
        _check(GpuMatcher(dl).match(reads, None, max_mismatch=d, algo="seed", first_index=0), oracle.best_hit(lib_s, reads_s, d))
        assert dl.last_kernel == ("k_match_wq" if d < 3 else "cascade3:d2+k_match_pairs"), dl.last_kernel


@pytest.mark.parametrize("e8", ["1", "0"])
def test_empty_slots_of_the_eight_entry_slabs(gpu_device, oracle, e8, monkeypatch):
    """Tables 0/1 of the wave matcher in their eight-entry form mark an empty slot by a rest with the three spare bits of
    both planes set (at least three mismatches against any read) and class 0.  Reads whose rest is a run of A / C / G / T
    (planes all zeros / all ones: the nearest a read's rest gets to the sentinel), behind seeds that lead to partly
    empty, exactly full (8) and chained (9, 17) buckets and to buckets that do not exist, with 0 ... 3 substitutions —
    against the oracle, with the form on and off (BC_E8=0: seven entries + link)."""
    monkeypatch.setenv("BC_E8", e8)
    k = 20
    rng = np.random.default_rng(23)
    letters = np.array(list("ACGT"))
    rows = []
    for size in (1, 2, 7, 8, 9, 16, 17):
        for base in range(4):
            shared = rng.integers(0, 4, size=7)
            grp = rng.integers(0, 4, size=(size, k))
            grp[:, :7] = shared
            grp[0, 7:] = base          # the bucket's first guide ends in a run of one base
            rows.append(grp)
    codes = np.unique(np.concatenate(rows + [rng.integers(0, 4, size=(3000, k))]), axis=0)
    codes = codes[rng.permutation(codes.shape[0])]
    lib_s = ["".join(letters[r]) for r in codes]
    rd = []
    for grp in rows:
        for base in range(4):
            r = grp[0].copy()
            r[7:] = base               # the bucket's seed + a run of `base`
            rd.append(r)
            for nsub in (1, 2, 3):
                rd.append(_mutate(rng, r[None, :], np.array([nsub]))[0])
    for base in range(4):              # runs of one base over the whole read: mostly buckets that hold nothing like them
        rd.append(np.full(k, base))
        for _ in range(40):
            r = np.full(k, base)
            r[:7] = rng.integers(0, 4, size=7)
            rd.append(r)
    rd = np.array(rd)
    rd = np.concatenate([rd, _mutate(rng, codes[rng.integers(0, codes.shape[0], size=5000)], rng.integers(0, 4, size=5000))])
    reads_s = ["".join(letters[r]) for r in rd]
    lib, _ = E.pack_kmers(lib_s, k)
    reads, _ = E.pack_kmers(reads_s, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    for d in (1, 2, 3):
        _check(GpuMatcher(dl).match(reads, None, max_mismatch=d, algo="seed", first_index=0), oracle.best_hit(lib_s, reads_s, d))
        assert dl.last_kernel == ("k_match_wq" if d < 3 else "cascade3:d2+k_match_pairs"), dl.last_kernel


@pytest.mark.parametrize("G,d,algo", [(170_000, 1, "seed"), (300_000, 2, "seed"), (400_000, 1, "seed"), (700_000, 0, "hash"),
                                      (1_200_000, 0, "hash"), (1_200_000, 2, "seed"), (300_000, 3, "seed"), (1_200_000, 3, "seed")])
def test_record_tally_many_ranges(gpu_device, G, d, algo):
    """Libraries of 9 ... 59 tally ranges with at least 4 reads per guide (the size from which the LDS tally runs):
    record lines of 64 / 32 / 16 entries written by k_match_wq / k_match_hash, the sweeps or atomics for kernels
    that write none (1.2e6 guides at d = 2: k_match_fused).  Counts and first-seen must be what best_idx says —
    including a hot guide that overflows its range's line in every tile and a ragged last tile."""
    k = 20
    lib = S.library(G, k, seed=S.LIB_SEED + G)
    n = 4 * G + 4099
    reads = np.concatenate([S.reads(lib, k, 3, a, min(2_000_000, n - a), 1 if d else 0) for a in range(0, n, 2_000_000)])
    reads[1000:1000 + 200_000:2] = lib[G // 3]          # a hot guide: more hits in one range than a line holds
    dl = DeviceLibrary(lib, k, gpu_device)
    res = GpuMatcher(dl).match(reads, None, max_mismatch=d, algo=algo, first_index=7)
    hit = res.best_idx >= 0
    assert 0.75 * n < hit.sum() < (0.97 if d == 3 else 0.95) * n
    if d == 0:  # exact: the answer is known without the oracle
        order = np.argsort(lib)
        pos = np.searchsorted(lib[order], reads)
        pos[pos == G] = 0
        want = np.where(lib[order][pos] == reads, order[pos], -1).astype(np.int32)
        np.testing.assert_array_equal(res.best_idx, want)
    else:  # a prefix against the indexed oracle (1.2e6 guides at d = 2 run as a cascade: d <= 1 pass, then the pair index)
        m = 150_000
        sh = np.arange(k, dtype=np.uint64)
        letters = np.frombuffer(b"ACGT", np.uint8)

        def ascii_of(p):
            q = np.ascontiguousarray(p, np.uint64)[:, None]
            return letters[((q >> sh) & np.uint64(1)).astype(np.uint8) | (((q >> (sh + np.uint64(32))) & np.uint64(1)).astype(np.uint8) << 1)]
        from oracle import oracle as orc
        ref = orc.best_hit_arrays(ascii_of(lib), ascii_of(reads[:m] & E.kmer_mask(k)), d, indexed=True)
        np.testing.assert_array_equal(res.best_idx[:m], ref["best_idx"])
        np.testing.assert_array_equal(res.best_dist[:m], ref["best_dist"])
        np.testing.assert_array_equal(res.n_best[:m], ref["n_best"])
        if (G, d) == (1_200_000, 2):
            assert dl.last_kernel == "cascade2:d1+k_match_pairs"
        if d == 3:  # (at 1.2e6 guides a cascade over a cascade: the outer one gathers its own list)
            assert dl.last_kernel == "cascade3:d2+k_match_pairs"
    np.testing.assert_array_equal(res.counts, np.bincount(res.best_idx[hit], minlength=G).astype(np.uint64))
    first = np.full(G, np.iinfo(np.uint64).max, np.uint64)
    idx = np.flatnonzero(hit)
    np.minimum.at(first, res.best_idx[idx][::-1], (idx[::-1] + 7).astype(np.uint64))
    np.testing.assert_array_equal(res.first_seen, first)
    assert res.counts[G // 3] >= 100_000
