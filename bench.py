#!/usr/bin/env python3
"""bench.py — headline benchmark: Mreads/s through the <=2-mismatch 20-mer matcher.

Workload (BASELINE.json configs[2]): library of 1e5 distinct
20-mers, 1e8 synthetic reads PER GPU (0/1/2/3 substitutions 60/20/10/5 %, 5 %
random), generated on the device by the seeded splitmix64 stream so they are
resident in HBM before the timed region.  A step = one pass of the matcher over
the rank's reads (best index + distance + n_best + per-guide counts + first-seen)
followed, for N > 1, by the all-reduce (RCCL) of the count vector and of the
first-seen indices: barcoder_amd.dist.count_sharded, the same function
`python -m barcoder_amd.heuristicount --gpus N` uses.

`python bench.py --gpus N` from a bare shell starts its own N ranks (fresh child
processes, before anything in this process touches the GPU); under torchrun it
joins the group the launcher describes.

Prints ONE JSON line on rank 0 (see the contract in DESIGN.md "Measurement").
"""
import argparse
import importlib.util
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = 13          # 8 B packed read in + 4 B best index + 1 B distance out
HBM_PEAK_GBPS = 8000.0   # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy peak
K = 20
MAX_MM = 2
G = 100_000


def _dist_module_without_package():
    """barcoder_amd/dist.py loaded by path: the launching parent needs neither torch nor the native library."""
    spec = importlib.util.spec_from_file_location("_bc_dist_launcher", os.path.join(ROOT, "barcoder_amd", "dist.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ascii_from_packed(packed, k):
    """plane-form uint64 k-mers -> (n, k) uint8 ASCII array (what the string-based oracle reads)"""
    import numpy as np
    p = np.ascontiguousarray(packed, np.uint64)
    sh = np.arange(k, dtype=np.uint64)
    out = np.empty((p.size, k), np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    for a in range(0, p.size, 1 << 20):  # bounded temporaries
        q = p[a:a + (1 << 20), None]
        lo = ((q >> sh) & np.uint64(1)).astype(np.uint8)
        hi = ((q >> (sh + np.uint64(32))) & np.uint64(1)).astype(np.uint8)
        out[a:a + q.shape[0]] = letters[lo | (hi << 1)]
    return out


def cpu_baseline(lib_packed, n_sample, seed, max_mm, mixture=1):
    """The CPU port of the same job on the first reads of the same stream: oracle.c's pigeonhole-indexed best-hit
    search (strings, OpenMP, all host cores), sized from a probe to ~10 s of CPU work; the brute-force sweep the
    oracle is defined by runs on a prefix (~5 s) and must agree with it."""
    import numpy as np
    from barcoder_amd import synth as S
    from oracle import oracle as orc
    la = ascii_from_packed(lib_packed, K)
    probe = ascii_from_packed(S.reads(lib_packed, K, seed, 0, 200_000, mixture), K)
    t0 = time.perf_counter()
    orc.best_hit_arrays(la, probe, max_mm, indexed=True)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
    t0 = time.perf_counter()
    orc.best_hit_arrays(la, probe[:2000], max_mm, indexed=False)
    brate = 2000 / max(time.perf_counter() - t0, 1e-6)
    if n_sample <= 0:
        n_sample = int(min(40_000_000, max(200_000, rate * 10)))
    reads = S.reads(lib_packed, K, seed, 0, n_sample, mixture)
    ra = ascii_from_packed(reads, K)
    t0 = time.perf_counter()
    ref = orc.best_hit_arrays(la, ra, max_mm, indexed=True)
    dt = time.perf_counter() - t0
    nb = int(min(n_sample, max(20_000, brate * 5)))
    t0 = time.perf_counter()
    brute = orc.best_hit_arrays(la, ra[:nb], max_mm, indexed=False)
    bdt = time.perf_counter() - t0
    same = all(np.array_equal(brute[k_], ref[k_][:nb]) for k_ in ("best_idx", "best_dist", "n_best"))
    return ref, reads, dict(value=n_sample / dt / 1e6, unit="Mreads/s", cores=orc.threads(), kind="port",
                            sample=f"first {n_sample} reads of the same stream vs the full {la.shape[0]}-guide library, "
                                   f"{dt:.2f} s, oracle/oracle.c pigeonhole-indexed search (strings, OpenMP)",
                            brute_force_value=nb / bdt / 1e6, brute_force_sample=nb,
                            indexed_equals_brute_force_on_prefix=bool(same))


def cpu_baseline_exact(lib_packed, best_idx_dev, n):
    """configs[1] on the CPU: oracle.c's restatement of the reference's exact-match loop (orc_process_chunk_single =
    heuristicount.py:520-538: N-skip, window slice, flank test, string hash-set lookup) over the first reads of the
    same stream as 20-character strings, one thread (the reference runs one such loop per Pool worker)."""
    import ctypes as C
    import numpy as np
    from barcoder_amd import synth as S
    from oracle import oracle as orc
    m = int(min(n, 20_000_000))
    la = ascii_from_packed(lib_packed, K)
    reads = np.concatenate([S.reads(lib_packed, K, 1, a, min(5_000_000, m - a), 0) for a in range(0, m, 5_000_000)])
    text = ascii_from_packed(reads, K)
    off = np.arange(m + 1, dtype=np.uint64) * np.uint64(K)
    out = np.empty(m, np.int32)
    so = np.empty(m, np.int64)
    sl = np.empty(m, np.int64)
    t0 = time.perf_counter()
    orc.lib().orc_process_chunk_single(text.ctypes.data_as(C.c_char_p), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       C.c_uint64(m), la.ctypes.data_as(C.c_char_p), C.c_uint32(la.shape[0]), C.c_int(K),
                                       C.c_int64(0), C.c_char_p(b""), C.c_int(0), C.c_char_p(b""), C.c_int(0), C.c_int(0),
                                       out.ctypes.data_as(C.POINTER(C.c_int32)), so.ctypes.data_as(C.POINTER(C.c_int64)),
                                       sl.ctypes.data_as(C.POINTER(C.c_int64)))
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(best_idx_dev[:m].cpu().numpy(), out))
    return {"value": m / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": f"first {m} reads of the same stream as {K}-character strings, {dt:.2f} s, oracle/oracle.c "
                      "orc_process_chunk_single (heuristicount.py:520-538)",
            "parity_on_sample": same,
            "reference_measured_in_container": {"value": 0.79, "unit": "Mreads/s per core", "marginal_4_workers": 0.73,
                                                "source": "SURVEY.md section 6: heuristicount.process_chunk, 8 vCPU Xeon 2.6 GHz, Pool(4); not this host"}}


def secondary_workloads(torch, dev, local, stream, lib_packed, matcher, n, steps, with_cpu=True):
    """The other BASELINE configurations and the host-inclusive rates, measured after the timed region (rank 0,
    N = 1).  Each entry is bounded to a few seconds and reports its own failure instead of taking the headline down."""
    import numpy as np
    from barcoder_amd import synth as S
    out = {}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 - a secondary number must never hide the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)

    def exact():  # BASELINE configs[1]: exact match, 1e5 guides, 1e8 reads, HBM-resident
        reads = torch.empty(n, dtype=torch.int64, device=dev)
        bi = torch.empty(n, dtype=torch.int32, device=dev)
        bd = torch.empty(n, dtype=torch.uint8, device=dev)
        nb = torch.empty(n, dtype=torch.uint8, device=dev)
        c32 = torch.zeros(lib_packed.size, dtype=torch.int32, device=dev)
        fs = torch.full((lib_packed.size,), -1, dtype=torch.int64, device=dev)
        matcher.synth_reads_dev(1, 0, n, 0, reads.data_ptr(), stream)
        kw = dict(d_best_dist=bd.data_ptr(), d_n_best=nb.data_ptr(), stream=stream)
        matcher.time_dev(reads.data_ptr(), n, bi.data_ptr(), 0, "hash", iters=2, **kw)
        k_ms = matcher.time_dev(reads.data_ptr(), n, bi.data_ptr(), 0, "hash", iters=max(30, steps), **kw)
        p_ms = matcher.time_dev(reads.data_ptr(), n, bi.data_ptr(), 0, "hash", iters=max(30, steps),
                                d_counts=c32.data_ptr(), d_first_seen=fs.data_ptr(), **kw)
        hit = float((bi >= 0).float().mean().item())
        res = {"workload": f"20-mer exact match, {lib_packed.size} guides, {n} reads (BASELINE configs[1])",
               "kernel": "k_match_hash", "kernel_ms": k_ms, "timing": "HIP events on the launch stream (bc_match_time_dev)",
               "pass_ms": p_ms, "Mreads_per_s": n / p_ms / 1e3,
               "frac": n * ALGO_BYTES / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "pass_frac": n * ALGO_BYTES / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "matched_fraction": hit}
        if with_cpu:
            res["cpu_baseline"] = cpu_baseline_exact(lib_packed, bi, n)
        return res

    def pam():  # BASELINE configs[3]: NGG / 20-mer site scan over an E. coli-sized genome
        import ctypes as C
        from barcoder_amd import _native as N, pam as P
        from barcoder_amd.devmem import DevBuf
        from barcoder_amd.encoding import pam_masks
        L = N.load()
        bp = 4_641_652
        seq = S.genome(bp, seed=4)
        g = P.pack_genome(seq, wrap=22)
        masks = pam_masks("NGG")
        nw = g.bases.size

        def scan_ms(d_b, d_i, total, d_f, d_r, iters):
            # launches issued back to back from C, HIP events around them on the launch stream: the old figure (a host
            # timer around a Python loop of asynchronous launches) measured the host's launch rate — 7.7 us per launch
            # on an idle host, 81 us on the driver's box right after the 128-thread OpenMP baseline (its worker threads
            # still spinning) — not the kernel
            ms = C.c_float(0)
            N.check(L.bc_pam_scan_time_dev(d_b, d_i, total, N.ptr(masks), 3, d_f, d_r, iters, stream, C.byref(ms)))
            return float(ms.value)

        with DevBuf.from_numpy(g.bases, local) as d_b, DevBuf.from_numpy(g.invalid, local) as d_i, \
                DevBuf(nw * 4, local) as d_f, DevBuf(nw * 4, local) as d_r:
            scan_ms(d_b.ptr, d_i.ptr, g.total, d_f.ptr, d_r.ptr, 20)
            ms = scan_ms(d_b.ptr, d_i.ptr, g.total, d_f.ptr, d_r.ptr, 500)
            dev_bits = [np.unpackbits(d.to_numpy(np.uint32).view(np.uint8), bitorder="little")[:g.total] for d in (d_f, d_r)]
        P.find_sites(g, "NGG", 20)
        t0 = time.perf_counter()
        st = P.find_sites(g, "NGG", 20)
        pipe = time.perf_counter() - t0
        byts = nw * (8 + 4 + 4 + 4)
        res = {"workload": f"NGG site scan, both strands, {bp} bp synthetic circular genome (BASELINE configs[3])",
               "kernel": "k_pam_scan", "kernel_ms": ms, "us_per_scan": ms * 1e3,
               "timing": "HIP events around 500 back-to-back launches issued from C (bc_pam_scan_time_dev)",
               "Gbp_per_s": bp / (ms * 1e-3) / 1e9, "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "note": "launch-bound at this size (2.9 MB of algorithmic bytes per launch)",
               "find_sites_ms_incl_h2d_d2h": pipe * 1e3, "find_sites_launches": 2, "sites": int(st.start.size)}
        # the same kernel where it is HBM-bound: a 3.1e9-position genome (packed words filled on the device)
        big = 3_100_000_000
        bw = (big + 31) // 32
        tb = torch.randint(-2**62, 2**62, (bw,), dtype=torch.int64, device=dev)
        ti = torch.zeros(bw, dtype=torch.int32, device=dev)
        tf = torch.empty(bw, dtype=torch.int32, device=dev)
        tr = torch.empty(bw, dtype=torch.int32, device=dev)
        scan_ms(tb.data_ptr(), ti.data_ptr(), big, tf.data_ptr(), tr.data_ptr(), 2)
        bms = scan_ms(tb.data_ptr(), ti.data_ptr(), big, tf.data_ptr(), tr.data_ptr(), 10)
        del tb, ti, tf, tr
        res["hbm_bound_3p1Gbp"] = {"bp": big, "kernel_ms": bms, "algorithmic_GB_per_s": bw * 20 / (bms * 1e-3) / 1e9,
                                   "frac": bw * 20 / (bms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   "bytes_per_32bp": "12 in (planes + invalid bits) + 8 out (two bitmaps)"}
        if with_cpu:
            from oracle import oracle as orc
            topo = seq + seq[:22]
            t0 = time.perf_counter()
            of, orv = orc.pam_bitmaps(topo, "NGG")
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": bp / cdt / 1e6, "unit": "Mbp/s", "cores": 1, "kind": "port",
                                   "sample": f"the whole {bp}-bp genome, both strands, oracle/oracle.c orc_pam_scan (characters), {cdt * 1e3:.1f} ms",
                                   "reference_measured_elsewhere": "design_interactive.ipynb:306,344 logs ~6 s for the reference's "
                                                                   "find_sequences_with_barcode_and_pam on E. coli (unknown hardware)",
                                   "parity_on_sample": bool(np.array_equal(dev_bits[0], of) and np.array_equal(dev_bits[1], orv))}
        return res

    def host_pcie():  # bc_match_host: host buffers in, host results out (PCIe included), <= 2 mismatches
        # held against the link's own copy rates, measured here (page-locked memory <-> HBM, bc_pcie_probe): 8 bytes per
        # read go up and 6 come back, so the all-outputs path is priced against the duplex rate and the counts-only path
        # (nothing per read comes back) against the upload rate
        from barcoder_amd.devmem import PinnedBlock
        m = min(n, 20_000_000)
        reads = np.concatenate([S.reads(lib_packed, K, 2, a, min(5_000_000, m - a), 1) for a in range(0, m, 5_000_000)])
        probe = matcher.pcie_probe(local)
        pin = PinnedBlock(m * 8)
        preads = pin.array(np.uint64, m)
        preads[:] = reads

        def timed(fn, reps=3):
            fn()  # untimed: staging buffers, pooled result blocks, first-touch of everything
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                res = fn()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return best, res

        dt_page, res = timed(lambda: matcher.match(reads, None, MAX_MM, "seed"))
        dt_pin, res_pin = timed(lambda: matcher.match(preads, None, MAX_MM, "seed"))
        dt_cnt, res_cnt = timed(lambda: matcher.match(preads, None, MAX_MM, "seed", per_read=False))
        same = bool(np.array_equal(res.best_idx, res_pin.best_idx) and np.array_equal(res.counts, res_pin.counts) and
                    np.array_equal(res.counts, res_cnt.counts) and np.array_equal(res.first_seen, res_cnt.first_seen))
        link = lambda dt, b: m * b / dt / 1e9
        return {"workload": f"bc_match_host, {m} reads, <= {MAX_MM} mismatches; best of 3 calls after one untimed",
                "Mreads_per_s": m / dt_page / 1e6, "GB_per_s_over_pcie": link(dt_page, 14),
                "pageable": {"what": "reads in pageable memory (staged by worker threads), results in the wrapper's page-locked arrays",
                             "Mreads_per_s": m / dt_page / 1e6, "GB_per_s_over_pcie": link(dt_page, 14),
                             "frac_of_duplex": link(dt_page, 14) / probe["duplex_GBps"]},
                "pinned": {"what": "reads and results in page-locked memory (no staging)",
                           "Mreads_per_s": m / dt_pin / 1e6, "GB_per_s_over_pcie": link(dt_pin, 14),
                           "frac_of_duplex": link(dt_pin, 14) / probe["duplex_GBps"]},
                "counts_only": {"what": "page-locked reads up, per-guide counts / first-seen back once (per_read=False)",
                                "Mreads_per_s": m / dt_cnt / 1e6, "GB_per_s_over_pcie": link(dt_cnt, 8),
                                "frac_of_h2d": link(dt_cnt, 8) / probe["h2d_GBps"]},
                "roofline": {"bound": "pcie", "unit": "GB/s", "peak": probe["duplex_GBps"], "h2d": probe["h2d_GBps"],
                             "d2h": probe["d2h_GBps"], "achieved": link(dt_pin, 14), "frac": link(dt_pin, 14) / probe["duplex_GBps"],
                             "source": "bc_pcie_probe in this run: 256 MiB each way between page-locked memory and HBM, 4 repeats"},
                "modes_agree": same, "matched_fraction": float((res.best_idx >= 0).mean())}

    def from_file():  # heuristicount end to end from a .reads file (sampling + bulk pass + merge)
        import shutil
        import tempfile
        from barcoder_amd import encoding as E, heuristicount as H
        m = min(n, 10_000_000)
        d = tempfile.mkdtemp(prefix="bc_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            labels = E.unpack_kmers(lib_packed, K)
            fa, rd = os.path.join(d, "lib.fasta"), os.path.join(d, "reads.reads")
            with open(fa, "w") as f:
                f.write("".join(f">g{i}\n{s}\n" for i, s in enumerate(labels)))
            lut = np.frombuffer(b"ACGT", np.uint8)
            sh = np.arange(K, dtype=np.uint64)
            with open(rd, "wb") as f:
                for a in range(0, m, 2_000_000):
                    c = min(2_000_000, m - a)
                    w = S.reads(lib_packed, K, 2, a, c, 1)
                    codes = (((w[:, None] >> sh) & np.uint64(1)) | (((w[:, None] >> (sh + np.uint64(32))) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
                    pre = lut[((S.draw(9, np.arange(a, a + c, dtype=np.uint64), 0)[:, None] >> (np.arange(5, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)).astype(np.uint8)]
                    row = np.empty((c, 46), np.uint8)
                    row[:, :5] = pre
                    row[:, 5:15] = np.frombuffer(b"CCGGATTCAG", np.uint8)
                    row[:, 15:35] = lut[codes]
                    row[:, 35:45] = np.frombuffer(b"GTTTTAGAGC", np.uint8)
                    row[:, 45] = 10
                    f.write(row.tobytes())
            with open(rd, "rb", buffering=0) as f:  # (a just-written file reads 3-4 x slower the first time: the job sees
                buf = bytearray(64 << 20)          #  one that has been read once — "in the page cache")
                while f.readinto(buf):
                    pass
            t0 = time.perf_counter()
            r = H.count_barcodes(fa, rd, device=local)
            dt = time.perf_counter() - t0
            return {"workload": f"heuristicount.count_barcodes on a {m}-read .reads file (46-byte lines, in the page cache), exact match",
                    "Mreads_per_s": m / dt / 1e6, "seconds": dt, "documented_reads": int(sum(r["doc"].values())),
                    "bulk_Mreads_per_s": r.get("bulk_mreads_per_s")}
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def wider():  # beyond the BASELINE configurations: three mismatches, and a five times larger library
        from barcoder_amd import DeviceLibrary, GpuMatcher
        reads = torch.empty(n, dtype=torch.int64, device=dev)
        bi = torch.empty(n, dtype=torch.int32, device=dev)
        bd = torch.empty(n, dtype=torch.uint8, device=dev)
        nb = torch.empty(n, dtype=torch.uint8, device=dev)
        res = {"workload": f"{n} reads of the standard stream (seed 2), HBM-resident; pass = matcher + tally, HIP events (bc_match_time_dev)"}

        def one(m_, lib_size, d_):
            c32 = torch.zeros(lib_size, dtype=torch.int32, device=dev)
            fs = torch.full((lib_size,), -1, dtype=torch.int64, device=dev)
            m_.synth_reads_dev(2, 0, n, 1, reads.data_ptr(), stream)
            kw = dict(d_best_dist=bd.data_ptr(), d_n_best=nb.data_ptr(), d_counts=c32.data_ptr(), d_first_seen=fs.data_ptr(), stream=stream)
            m_.time_dev(reads.data_ptr(), n, bi.data_ptr(), d_, "seed", iters=3, **kw)
            p_ms = m_.time_dev(reads.data_ptr(), n, bi.data_ptr(), d_, "seed", iters=max(30, steps), **kw)
            return {"guides": lib_size, "max_mismatch": d_, "kernel": m_.lib.last_kernel, "pass_ms": p_ms, "Mreads_per_s": n / p_ms / 1e3,
                    "matched_fraction": float((bi >= 0).float().mean().item())}

        res["d3_1e5_guides"] = one(matcher, int(lib_packed.size), 3)
        big = S.library(500_000, K)
        res["d2_5e5_guides"] = one(GpuMatcher(DeviceLibrary(big, K, local)), int(big.size), 2)
        return res

    guarded("exact_1e8", exact)
    guarded("wider", wider)
    guarded("pam_4p6Mbp", pam)
    guarded("host_pcie", host_pcie)
    guarded("from_file", from_file)
    return out


def main():
    global K
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 timed steps after 20 untimed ones (0.08 s in all).  The part needs ~20 launches of sustained load to
    # settle its clocks after a pause (profiles/r04_clock_ramp.txt: 1.06 ms per kernel at the start of a burst, 0.89-0.93
    # at its end), so a short run measures the ramp; the driver's own --steps / --warmup are taken as given.
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reads", type=float, default=None,
                    help="reads per GPU (default 1e8 = BASELINE configs[2]; with --gpus 8: 1.25e8 = configs[4], 1e9 reads in all)")
    ap.add_argument("--guides", type=int, default=G)
    ap.add_argument("--k", type=int, default=K, help="k-mer length (default 20: BASELINE configs[2]; 32 = the shipped CN-32 geometry)")
    ap.add_argument("--algo", default="seed", choices=["seed", "seed_csr", "brute", "hash", "auto"])
    ap.add_argument("--max-mismatch", type=int, default=MAX_MM)
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0 = ~15 s worth)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads after the timed region")
    ap.add_argument("--mixture", type=int, default=1, help="synthetic stream: 1 = SURVEY cfg3 mismatch mix (default), 0 = cfg2 exact mix, 100 + J = read-mix sweep: J %% random "
                         "(unmatched) reads, the others with 0/1/2 substitutions 6:2:1")
    args = ap.parse_args()

    if args.reads is None:
        args.reads = 1.25e8 if args.gpus == 8 else 1e8
        if args.gpus > 1:  # the children parse the same value
            sys.argv += ["--reads", repr(args.reads)]
    # ---- self-launch: N ranks as fresh child processes, before this process imports torch or touches the GPU
    if args.gpus > 1 and not ("WORLD_SIZE" in os.environ and "RANK" in os.environ):
        launcher = _dist_module_without_package()
        # a rank that hangs (RCCL initialisation, a collective a peer never joins) must not hold the job: bounded wait
        raise SystemExit(launcher.launch_workers([os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                                 timeout=float(os.environ.get("BC_LAUNCH_TIMEOUT", 1800))))

    K = args.k
    from barcoder_amd import dist as D
    with D.stdout_to_stderr():  # stdout carries the one JSON line and nothing else (gloo prints to fd 1)
        line = run(args)
    if line is not None:
        print(line, flush=True)


def run(args):
    import numpy as np
    import torch
    from barcoder_amd import DeviceLibrary, GpuMatcher, dist as D, synth as S

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n = int(args.reads)
    seed = 2
    Gn = args.guides

    lib_packed = S.library(Gn, K)
    dl = DeviceLibrary(lib_packed, K, local)
    m = GpuMatcher(dl)
    stream = torch.cuda.current_stream().cuda_stream

    reads = torch.empty(n, dtype=torch.int64, device=dev)
    best_idx = torch.empty(n, dtype=torch.int32, device=dev)
    best_dist = torch.empty(n, dtype=torch.uint8, device=dev)
    n_best = torch.empty(n, dtype=torch.uint8, device=dev)
    counts32 = torch.zeros(Gn, dtype=torch.int32, device=dev)
    first_seen = torch.full((Gn,), -1, dtype=torch.int64, device=dev)  # 0xFFFF.. = unseen
    # job-wide results, double-buffered: the all-reduce of one step runs (on RCCL's stream) under the next step's
    # matcher; a buffer is reused only after its collectives have been waited for
    out_c = [torch.zeros(Gn, dtype=torch.int64, device=dev) for _ in range(2)]
    out_f = [torch.empty(Gn, dtype=torch.int64, device=dev) for _ in range(2)]
    pending = [D.Pending(), D.Pending()]
    overlap = world > 1 and os.environ.get("BC_BENCH_ASYNC_REDUCE", "1") != "0"
    turn = [0]
    first = rank * n  # global index of this rank's first read (weak scaling: n per GPU)
    m.synth_reads_dev(seed, first, n, args.mixture, reads.data_ptr(), stream)
    torch.cuda.synchronize()

    def match_shard(a, b):
        """This rank's shard [a, b) of the job's world*n reads (resident since before the timed region)."""
        assert (a, b) == (first, first + n)
        m.match_dev(reads.data_ptr(), n, best_idx.data_ptr(), max_mismatch=args.max_mismatch, algo=args.algo,
                    d_best_dist=best_dist.data_ptr(), d_n_best=n_best.data_ptr(), d_counts=counts32.data_ptr(),
                    d_first_seen=first_seen.data_ptr(), first_index=first, stream=stream)
        # one launch: 64-bit sums, unseen -> the identity of MIN over int64, accumulators reset for the next pass
        b_ = turn[0]
        pending[b_].wait()  # the collectives that last used this buffer (two steps ago)
        m.tally_export_dev(counts32.data_ptr(), first_seen.data_ptr(), out_c[b_].data_ptr(), out_f[b_].data_ptr(),
                           reset=True, stream=stream)
        return out_c[b_], out_f[b_]

    def step():
        """One pass over this rank's shard + the job-wide reduction (started here; awaited before its buffer is
        reused, and by finish() at the end of the timed region)."""
        b_ = turn[0]
        if overlap:
            c, f, pending[b_] = D.count_sharded(world * n, match_shard, Gn, rank, world, device=dev, async_op=True)
        else:
            c, f = D.count_sharded(world * n, match_shard, Gn, rank, world, device=dev)
        turn[0] ^= 1
        return c, f

    def finish():
        for p_ in pending:
            p_.wait()

    for _ in range(args.warmup):
        step()
    finish()
    D.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    finish()  # every step's job-wide result is complete inside the timed region
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.reduce_scalar_max(time.perf_counter() - t0, device=dev)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med_ms = D.reduce_scalar_max(statistics.median(step_ms), device=dev)
    job_counts, job_first = step()  # the job-wide result (identical on every rank)
    finish()
    total_hits = int(job_counts.sum().item())

    # dominant kernel alone (the matcher without the tally), HIP events on the launch stream: roofline.achieved;
    # then the whole pass (matcher + tally) the same way.  At least 50 launches each: a burst that follows a pause (the
    # .item() above) starts with the clock ramp, and an average over fewer launches measures mostly that
    # (profiles/r04_clock_ramp.txt)
    k_ms = m.time_dev(reads.data_ptr(), n, best_idx.data_ptr(), args.max_mismatch, args.algo, iters=max(50, args.steps),
                      d_best_dist=best_dist.data_ptr(), d_n_best=n_best.data_ptr(), stream=stream)
    # the same again over 40 launches issued straight after those: the kernel at the clocks a long run settles at (the
    # first ~20 launches after any pause run 5-10 % slower while the power management settles: profiles/r04_clock_ramp.txt);
    # reported beside kernel_ms, which stays what roofline.frac is computed from
    k_ms_steady = m.time_dev(reads.data_ptr(), n, best_idx.data_ptr(), args.max_mismatch, args.algo, iters=40,
                             d_best_dist=best_dist.data_ptr(), d_n_best=n_best.data_ptr(), stream=stream)
    counts32.zero_()
    first_seen.fill_(-1)
    pass_ms = m.time_dev(reads.data_ptr(), n, best_idx.data_ptr(), args.max_mismatch, args.algo, iters=max(50, args.steps),
                         d_best_dist=best_dist.data_ptr(), d_n_best=n_best.data_ptr(), d_counts=counts32.data_ptr(),
                         d_first_seen=first_seen.data_ptr(), stream=stream)
    achieved = n * ALGO_BYTES / (k_ms * 1e-3) / 1e9

    if rank == 0:
        matched = int((best_idx >= 0).sum().item())
        try:  # the metric string is BASELINE.json's own
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "Mreads/s matched, 20-mer <=2 mismatch, at 1/2/4/8 MI355X; % HBM roofline"
        backend = torch.distributed.get_backend() if world > 1 else None
        out = {
            "metric": metric,
            "value": world * n * args.steps / dt / 1e6,
            "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "value_median": world * n / (med_ms * 1e-3) / 1e6,
            "ms_per_step_median": med_ms,
            "config": {"workload": f"{K}-mer <={args.max_mismatch}-mismatch best hit, {Gn}-guide library, "
                                   f"{n} reads per GPU" + (" (BASELINE configs[4]: 1e9 reads over 8 GPUs)" if (K, args.max_mismatch, Gn, n, world) == (20, 2, 100_000, 125_000_000, 8)
                                                           else " (BASELINE configs[2])" if (K, args.max_mismatch, Gn) == (20, 2, 100_000) else ""),
                       "algo": args.algo, "reads_per_gpu": n, "guides": Gn, "k": K,
                       "max_mismatch": args.max_mismatch, "matched_fraction": matched / n,
                       "job_matched_reads": total_hits,
                       "outputs": "best_idx i32 + best_dist u8 + n_best u8 + counts[G] + first_seen[G] (job-wide on every rank)",
                       "parallelism": f"reads sharded x{world}, counts SUM + first-seen MIN all-reduced"
                                      + (f" ({backend})" if backend else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": dl.last_kernel,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_read": ALGO_BYTES,
                         "kernel_ms_steady": k_ms_steady,
                         "frac_steady": n * ALGO_BYTES / (k_ms_steady * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "steady_note": "the next 40 launches after the ones kernel_ms averages: settled clocks; frac uses kernel_ms",
                         "frac_of_measured_copy_peak": achieved / 6290.0,
                         "pass_ms": pass_ms, "pass": "matcher + tally (per-guide counts and first-seen)",
                         "pass_achieved": n * ALGO_BYTES / (pass_ms * 1e-3) / 1e9,
                         "pass_frac": n * ALGO_BYTES / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        # roofline.traffic: HBM bytes per launch of the dominant kernel from the separate FETCH_SIZE / WRITE_SIZE passes
        # (a PMC pass cannot run inside this process).  The tracked profile is quoted only when it was collected on the
        # kernel sources this run was built from and on the kernel that just ran; otherwise the line says "stale".
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        out["roofline"]["traffic_source"] = "none (profiles/traffic.json absent)"
        if os.path.exists(tr):
            try:
                from barcoder_amd.stamp import match_source_stamp
                t = json.load(open(tr))
                here = match_source_stamp()
                same_cfg = t.get("reads") == n and t.get("algo") == args.algo and t.get("max_mismatch") == args.max_mismatch \
                    and Gn == G and K == 20
                same_src = t.get("match_source_stamp") == here
                same_kernel = str(t.get("dominant_kernel", "")).startswith(str(dl.last_kernel))
                if not same_cfg:
                    out["roofline"]["traffic_source"] = "none (profiles/traffic.json is for another configuration)"
                elif not (same_src and same_kernel):
                    out["roofline"]["traffic_source"] = (f"stale: profiles/traffic.json was measured on kernel sources "
                                                         f"{t.get('match_source_stamp')} / {t.get('dominant_kernel')}, this run is "
                                                         f"{here} / {dl.last_kernel} (re-run profiles/refresh.sh)")
                else:
                    # per launch of the dominant kernel, like `achieved`; the whole pass (with the tally) beside it
                    dom = [v["hbm_bytes"] for k_, v in t.get("per_kernel", {}).items() if k_.startswith("k_match")]
                    out["roofline"]["traffic"] = dom[0] if dom else t["hbm_bytes_per_pass"]
                    out["roofline"]["pass_traffic"] = t["hbm_bytes_per_pass"]
                    out["roofline"]["traffic_source"] = (f"profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                         f"{t.get('dominant_kernel')} on kernel sources {here} ({t.get('source')}); "
                                                         "not collected in this run")
                    notes = [v["note"] for k_, v in t.get("per_kernel", {}).items() if k_.startswith("k_match") and v.get("note")]
                    if notes:  # e.g. the read-stream prefetch: part of the stream is fetched twice on purpose
                        out["roofline"]["traffic_note"] = notes[0]
            except Exception as e:  # noqa: BLE001
                out["roofline"]["traffic_source"] = f"none (profiles/traffic.json unreadable: {e})"
        if world == 1 and not args.no_cpu:
            ref, sample_reads, cb = cpu_baseline(lib_packed, args.cpu_sample, seed, args.max_mismatch, args.mixture)
            m_ = len(sample_reads)
            cb["parity_on_sample"] = bool(np.array_equal(best_idx[:m_].cpu().numpy(), ref["best_idx"]) and
                                          np.array_equal(best_dist[:m_].cpu().numpy(), ref["best_dist"]) and
                                          np.array_equal(n_best[:m_].cpu().numpy(), ref["n_best"]))
            out["cpu_baseline"] = cb
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_workloads(torch, dev, local, stream, lib_packed, m, n, args.steps,
                                                   with_cpu=not args.no_cpu)
    D.barrier()
    D.shutdown()
    return json.dumps(out) if rank == 0 else None


if __name__ == "__main__":
    main()
