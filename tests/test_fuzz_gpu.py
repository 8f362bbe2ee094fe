"""GPU: a seeded randomised parity sweep of the matcher against the CPU oracle — k, library size, CLUSTERED libraries
(many guides within 1-3 substitutions of each other: ties, tie counts, neighbour lists, the exact path), read mixes
from "every read exact" to "every read random", skip flags, ragged sizes, budgets 0..3, both exact-match algorithms —
through GpuMatcher.match (host buffers -> bc_match_host).  BC_FUZZ_CASES raises the number of cases per seed."""
import os

import numpy as np
import pytest

from barcoder_amd import DeviceLibrary, GpuMatcher, encoding as E, synth as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_random_cases_agree_with_the_oracle(gpu_device, oracle, seed):
    rng = np.random.default_rng(seed)
    n_cases = int(os.environ.get("BC_FUZZ_CASES", "25"))
    bad = []
    for case in range(n_cases):
        k = int(rng.choice([8, 10, 12, 15, 16, 17, 20, 20, 20, 21, 24, 25, 26, 28, 31, 32]))
        G = int(rng.choice([1, 2, 9, 10, 50, 300, 2000, 5000, 20000, 60000]))
        d = min(int(rng.integers(0, 4)), k - 1)
        lib = S.library(G, k, seed=int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.5 and G >= 10:  # clustered: part of the library becomes close relatives of other guides
            m = G // 3
            rel = lib[rng.integers(0, G, m)].copy()
            for _ in range(int(rng.integers(1, 4))):
                pos = rng.integers(0, k, m).astype(np.uint64)
                x = rng.integers(1, 4, m).astype(np.uint64)
                rel ^= ((x & np.uint64(1)) << pos) | ((x >> np.uint64(1)) << (pos + np.uint64(32)))
            lib[rng.permutation(G)[:m]] = rel
            lib = np.unique(lib)        # (a library holds no duplicates; its order is free)
            rng.shuffle(lib)
            G = int(lib.size)
        n = int(rng.choice([1, 3, 255, 256, 257, 1000, 4099, 20000, 65537]))
        mix = int(rng.choice([0, 1, 1, 1, 105, 150, 199, 200]))
        reads = S.reads(lib, k, int(rng.integers(1, 1 << 20)), int(rng.integers(0, 1000)), n, mix)
        if rng.random() < 0.2:
            reads = lib[rng.integers(0, G, n)]   # every read exact
        flags = (rng.random(n) < rng.choice([0.0, 0.02, 0.5])).astype(np.uint8) if rng.random() < 0.7 else None
        base = int(rng.integers(0, 1 << 40))
        algo = "hash" if (d == 0 and rng.random() < 0.5) else "seed"
        res = GpuMatcher(DeviceLibrary(lib, k, gpu_device)).match(reads, flags, max_mismatch=d, algo=algo, first_index=base)
        ref = oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k), d, skip=flags, base=base)
        if not all(np.array_equal(getattr(res, a), ref[a]) for a in ("best_idx", "best_dist", "n_best", "counts", "first_seen")):
            bad.append((case, k, G, d, n, mix, algo, flags is not None))
    assert not bad, bad


@pytest.mark.parametrize("env", [{"BC_CASCADE2_MIN_G": "1"}, {"BC_CASCADE2_MIN_G": "1", "BC_PAIR_PASS_MIN_READS": "0"},
                                 {"BC_CASCADE2_MIN_G": "1", "BC_PAIR_PASS_MIN_READS": "0", "BC_PAIR_GROUP_KB": "1"},
                                 {"BC_CASCADE2_MIN_G": "1", "BC_PAIR_FOUR": "0"}])
def test_two_mismatch_cascade_counts_every_guide_once(gpu_device, oracle, env, monkeypatch):
    """Budget 2 as a cascade (d <= 1 pass, then the pair index) on TIGHTLY clustered libraries: families of guides one to
    four substitutions apart, reads zero to three substitutions from a family member — so that a read has several guides
    within the budget, each with three, four or five clean pieces, found in several of the read's buckets.  For budgets
    <= 2 the second stage walks only the four tables (0,1) (2,3) (2,4) (3,4) with their own count-once rule
    (bc_match_pairs.hip; BC_PAIR_FOUR=0: all ten with the lexicographic rule): best index, distance and the TIE COUNT must
    be the oracle's in the one-kernel form, the list form (one table per pass too) and with all ten tables."""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    rng = np.random.default_rng(41)
    seen = ties = 0
    for case in range(12):
        k = int(rng.choice([8, 11, 16, 20, 20, 20, 23, 25]))
        fam, per = int(rng.choice([3, 40, 400])), int(rng.choice([4, 12, 30]))
        centres = S.library(fam, k, seed=int(rng.integers(1, 1 << 30)))
        lib = np.repeat(centres, per)
        for _ in range(2):              # every member: up to two substitutions from its centre (members up to four apart)
            pos = rng.integers(0, k, lib.size).astype(np.uint64)
            x = rng.integers(0, 4, lib.size).astype(np.uint64)
            lib ^= ((x & np.uint64(1)) << pos) | ((x >> np.uint64(1)) << (pos + np.uint64(32)))
        lib = np.unique(np.concatenate([lib, S.library(int(rng.choice([0, 50, 3000])) + 1, k, seed=case + 5)]))
        rng.shuffle(lib)
        n = int(rng.choice([257, 4099, 30001]))
        reads = lib[rng.integers(0, lib.size, n)].copy()
        for _ in range(3):              # zero to three substitutions (x = 0: none at that turn)
            pos = rng.integers(0, k, n).astype(np.uint64)
            x = rng.integers(0, 4, n).astype(np.uint64) * (rng.random(n) < 0.6)
            reads ^= ((x & np.uint64(1)) << pos) | ((x >> np.uint64(1)) << (pos + np.uint64(32)))
        flags = (rng.random(n) < 0.03).astype(np.uint8)
        dl = DeviceLibrary(lib, k, gpu_device)
        res = GpuMatcher(dl).match(reads, flags, max_mismatch=2, algo="seed", first_index=3)
        seen += dl.last_kernel == "cascade2:d1+k_match_pairs"
        ref = oracle.best_hit(E.unpack_kmers(lib, k), E.unpack_kmers(reads & E.kmer_mask(k), k), 2, skip=flags, base=3)
        ties += int((ref["n_best"] > 1).sum())
        for a in ("best_idx", "best_dist", "n_best", "counts", "first_seen"):
            np.testing.assert_array_equal(getattr(res, a), ref[a], err_msg=f"case {case} k {k} G {lib.size} n {n}: {a}")
    assert seen >= 8 and ties > 5000, (seen, ties)   # (the cascade really ran, and on reads with several best guides)


@pytest.mark.parametrize("seed", [3, 4])
def test_random_pam_scans_agree_with_the_oracle(gpu_device, oracle, seed):
    """Random sequences (ACGT with runs of N and stray lower-case letters, lengths around the 32- and 128-position word and
    lane boundaries and up to 3e5) x random patterns over {A, C, G, T, N} of 1 ... 8 letters: both strands' PAM bitmaps
    (`pam.pam_scan` -> k_pam_scan) against oracle.pam_bitmaps, and the sliding spacer set (`pam.sliding_spacers` ->
    the fused two-launch site pipeline) against oracle.sliding_sites_py, both directions."""
    import random
    from barcoder_amd import pam as P
    rng = random.Random(seed)
    n_cases = int(os.environ.get("BC_FUZZ_CASES", "25"))
    for case in range(n_cases):
        n = rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 4095, 4097, 8191, 8193, 20000, 100003, 300000])
        s = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(rng.randrange(0, 4)):            # runs of N
            if n:
                a = rng.randrange(n)
                for j in range(a, min(n, a + rng.randrange(1, 40))):
                    s[j] = "N"
        for _ in range(rng.randrange(0, 3)):            # stray lower-case letters (not ACGT to the scan)
            if n:
                s[rng.randrange(n)] = rng.choice("acgtn")
        s = "".join(s)
        pam = "".join(rng.choice("ACGTNNN") for _ in range(rng.randrange(1, 9)))
        # (case_sensitive: lower-case letters match nothing, as for the reference's regexes on a raw string; the default of
        # pack_genome folds case, a convenience the drop-in callers do not use)
        fwd, rev = P.pam_scan(P.pack_genome(s, 0, case_sensitive=True), pam, gpu_device)
        f2, r2 = oracle.pam_bitmaps(s, pam)
        assert np.array_equal(fwd, f2.astype(bool)) and np.array_equal(rev, r2.astype(bool)), (case, n, pam)
        if n <= 20000:
            k = rng.choice([4, 6, 12, 20, 32])
            d = rng.choice(["downstream", "upstream"])
            assert P.sliding_spacers(s, k, pam, d, gpu_device) == oracle.sliding_sites_py(s, k, pam, d), (case, n, pam, k, d)
