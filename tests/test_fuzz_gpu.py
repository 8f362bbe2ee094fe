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
