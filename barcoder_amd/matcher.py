"""GpuMatcher — the read -> library search that stands where the reference has
``process_chunk``'s set lookup (heuristicount.py:449-453) and, for d > 0, the
``BowtieRunner`` / ``run_bowtie_and_parse`` seam (BowtieRunner.py:104-141,
targets.py:467-539).  Bowtie is bypassed, not wrapped.

Everything here calls the C ABI; there is no CPU path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .library import DeviceLibrary

U64_MAX = np.iinfo(np.uint64).max


@dataclass
class MatchResult:
    best_idx: np.ndarray    # int32[n]: guide index, -1 miss, -2 skipped
    best_dist: np.ndarray   # uint8[n]: Hamming distance, 255 none
    n_best: np.ndarray      # uint8[n]: guides at that distance (saturating)
    counts: np.ndarray      # uint64[G]
    first_seen: np.ndarray  # uint64[G]: first read index per guide, U64_MAX unseen

    def counter(self, labels):
        """{barcode: count} in first-seen order (heuristicount.py:876-877)."""
        seen = np.flatnonzero(self.counts)
        order = seen[np.argsort(self.first_seen[seen], kind="stable")]
        return {labels[g]: int(self.counts[g]) for g in order}


class GpuMatcher:
    def __init__(self, library: DeviceLibrary):
        self.lib = library

    # -- host buffers ------------------------------------------------------------
    def match(self, reads, flags=None, max_mismatch: int = 0, algo: str = "auto", first_index: int = 0,
              counts=None, first_seen=None, per_read: bool = True) -> MatchResult:
        """bc_match_host.  The per-read result arrays of a large call are views on page-locked memory (pooled,
        devmem.PinnedBlock): the DMA writes them directly — no staging copy, no first-touch page faults — and they go
        back to the pool when the result is dropped.  per_read=False is the counts-only mode: `best_idx`, `best_dist`
        and `n_best` of the result are None, only the reads cross the link."""
        reads = np.ascontiguousarray(reads, np.uint64)
        n, G = reads.size, self.lib.size
        if flags is not None:
            flags = np.ascontiguousarray(flags, np.uint8)
            if flags.size != n:
                raise ValueError("flags and reads differ in length")
        bi = bd = nb = None
        if per_read and n >= (1 << 18):
            from .devmem import PinnedBlock
            pad = (n + 63) & ~63
            blk = PinnedBlock(pad * 6)
            bi, bd, nb = blk.array(np.int32, n, 0), blk.array(np.uint8, n, pad * 4), blk.array(np.uint8, n, pad * 5)
        elif per_read:
            bi, bd, nb = np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(n, np.uint8)
        counts = np.zeros(G, np.uint64) if counts is None else counts
        first_seen = np.full(G, U64_MAX, np.uint64) if first_seen is None else first_seen
        N.check(N.load().bc_match_host(self.lib.handle, N.ptr(reads), N.ptr(flags), n, int(max_mismatch),
                                       N.ALGOS[algo], N.ptr(bi), N.ptr(bd), N.ptr(nb), N.ptr(counts),
                                       N.ptr(first_seen), int(first_index)))
        return MatchResult(bi, bd, nb, counts, first_seen)

    @staticmethod
    def pcie_probe(device: int = 0, nbytes: int = 256 << 20, repeats: int = 4) -> dict:
        """Copy rates of the host link (page-locked memory <-> HBM), GB/s: bc_pcie_probe."""
        up, dn, both = C.c_double(), C.c_double(), C.c_double()
        N.check(N.load().bc_pcie_probe(device, nbytes, repeats, C.byref(up), C.byref(dn), C.byref(both)))
        return {"h2d_GBps": up.value, "d2h_GBps": dn.value, "duplex_GBps": both.value}

    # -- device buffers (torch tensors or raw pointers) ------------------------------
    def match_dev(self, d_reads: int, n: int, d_best_idx: int, max_mismatch: int = 0, algo: str = "auto",
                  d_flags: int = 0, d_best_dist: int = 0, d_n_best: int = 0, d_counts: int = 0,
                  d_first_seen: int = 0, first_index: int = 0, stream: int = 0):
        N.check(N.load().bc_match_dev(self.lib.handle, d_reads, d_flags or None, n, int(max_mismatch),
                                      N.ALGOS[algo], d_best_idx, d_best_dist or None, d_n_best or None,
                                      d_counts or None, d_first_seen or None, int(first_index), stream or None))

    def tally_export_dev(self, d_counts: int, d_first_seen: int, d_counts64: int, d_first_min: int,
                         reset: bool = False, stream: int = 0):
        """counts -> int64 sums, first_seen -> int64 with INT64_MAX for unseen (what the job-wide SUM / MIN
        all-reduce takes); reset leaves the accumulators empty for the next pass (bc_tally_export_dev)."""
        N.check(N.load().bc_tally_export_dev(d_counts or None, d_first_seen or None, d_counts64 or None,
                                             d_first_min or None, self.lib.size, int(bool(reset)), stream or None))

    def time_dev(self, d_reads: int, n: int, d_best_idx: int, max_mismatch: int, algo: str, iters: int,
                 d_best_dist: int = 0, d_n_best: int = 0, d_counts: int = 0, d_first_seen: int = 0,
                 stream: int = 0) -> float:
        """Mean duration (ms) of the match kernel over `iters` launches, by HIP
        events recorded on `stream` (bench.py's roofline.achieved)."""
        ms = C.c_float(0)
        N.check(N.load().bc_match_time_dev(self.lib.handle, d_reads, None, n, int(max_mismatch), N.ALGOS[algo],
                                           d_best_idx, d_best_dist or None, d_n_best or None, d_counts or None,
                                           d_first_seen or None, iters, stream or None, C.byref(ms)))
        return float(ms.value)

    def synth_reads_dev(self, seed: int, first: int, n: int, mixture: int, d_reads: int, stream: int = 0):
        N.check(N.load().bc_synth_reads_dev(self.lib.handle, seed, first, n, mixture, d_reads, stream or None))

    # -- all hits (guide -> genome, the Bowtie bypass) ---------------------------------
    def all_hits(self, queries, flags=None, max_mismatch: int = 0):
        """Every (query, guide, dist) with dist <= max_mismatch, sorted by
        (query, guide).  Host arrays in, structured numpy array out."""
        L = N.load()
        queries = np.ascontiguousarray(queries, np.uint64)
        n = queries.size
        if n == 0:
            return np.zeros(0, N.HIT_DTYPE)
        N.check(L.bc_dev_set(self.lib.device))
        d_q, d_f, d_h, d_n = N.vp(), N.vp(), N.vp(), N.vp()
        cap = max(1024, 2 * n)
        try:
            N.check(L.bc_dev_alloc(C.byref(d_q), n * 8))
            N.check(L.bc_dev_h2d(d_q, N.ptr(queries), n * 8, None))
            if flags is not None:
                flags = np.ascontiguousarray(flags, np.uint8)
                N.check(L.bc_dev_alloc(C.byref(d_f), n))
                N.check(L.bc_dev_h2d(d_f, N.ptr(flags), n, None))
            N.check(L.bc_dev_alloc(C.byref(d_n), 8))
            while True:
                N.check(L.bc_dev_alloc(C.byref(d_h), cap * 12))
                N.check(L.bc_dev_memset(d_n, 0, 8, None))
                N.check(L.bc_match_all_dev(self.lib.handle, d_q, d_f if flags is not None else None, n,
                                           int(max_mismatch), d_h, cap, d_n, None))
                tot = np.zeros(1, np.uint64)
                N.check(L.bc_dev_d2h(N.ptr(tot), d_n, 8, None))
                m = int(tot[0])
                if m <= cap:
                    hits = np.empty(m, N.HIT_DTYPE)
                    if m:
                        N.check(L.bc_dev_d2h(N.ptr(hits), d_h, m * 12, None))
                    break
                L.bc_dev_free(d_h)
                d_h = N.vp()
                cap = m
        finally:
            for p in (d_q, d_f, d_h, d_n):
                if p:
                    L.bc_dev_free(p)
        return np.sort(hits, order=("query", "guide"))


class MultiGpuMatcher:
    """One process, several GPUs: `bc_multi_*` — a replica of the library per device, contiguous
    read shards, one RCCL all-reduce of counts (SUM) and first-seen (MIN) at the end.  The
    single-process form of barcoder_amd.dist (heuristicount.py:703-737, Pool fan-out + merge).
    `devices` may repeat a device (rehearsal on fewer GPUs; the reduction then goes through the host)."""

    def __init__(self, packed, k: int, devices):
        packed = np.ascontiguousarray(packed, np.uint64)
        self.size, self.k = int(packed.size), int(k)
        devs = np.ascontiguousarray(devices, np.int32)
        self.handle = N.vp()
        N.check(N.load().bc_multi_create(N.ptr(packed), packed.size, self.k, N.ptr(devs), devs.size, C.byref(self.handle)))
        ng, ur = C.c_int(0), C.c_int(0)
        N.check(N.load().bc_multi_info(self.handle, C.byref(ng), C.byref(ur)))
        self.n_gpus, self.uses_rccl = ng.value, bool(ur.value)

    def match(self, reads, flags=None, max_mismatch: int = 0, algo: str = "auto", first_index: int = 0) -> MatchResult:
        reads = np.ascontiguousarray(reads, np.uint64)
        n, G = reads.size, self.size
        if flags is not None:
            flags = np.ascontiguousarray(flags, np.uint8)
            if flags.size != n:
                raise ValueError("flags and reads differ in length")
        bi, bd, nb = np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(n, np.uint8)
        counts, first_seen = np.zeros(G, np.uint64), np.full(G, U64_MAX, np.uint64)
        N.check(N.load().bc_multi_gpu_match(self.handle, N.ptr(reads), N.ptr(flags), n, int(max_mismatch), N.ALGOS[algo],
                                            N.ptr(bi), N.ptr(bd), N.ptr(nb), N.ptr(counts), N.ptr(first_seen),
                                            int(first_index)))
        return MatchResult(bi, bd, nb, counts, first_seen)

    def close(self):
        if self.handle:
            N.load().bc_multi_destroy(self.handle)
            self.handle = N.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
