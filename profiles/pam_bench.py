#!/usr/bin/env python3
"""PAM site scan timing (BASELINE configs[3]): NGG over a synthetic genome of
E. coli size (4,641,652 bp, seed 4, circular) and over a 3.1e9-bp genome to show
the HBM-bound regime.  Device-resident inputs; HIP events around `iters` launches
issued back to back from C (bc_pam_scan_time_dev), and — for comparison — the wall
clock around a Python loop of the same launches (the host's launch rate, which is
what round 2 reported).  usage: python profiles/pam_bench.py"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import _native as N, pam as P, synth as S
from barcoder_amd.devmem import DevBuf
from barcoder_amd.encoding import pam_masks

L = N.load()
masks = pam_masks("NGG")
out = []
for label, n, iters in (("ecoli_size", 4_641_652, 2000), ("3.1e9_bp", 3_100_000_000, 20)):
    if n < 10_000_000:
        g = P.pack_genome(S.genome(n, seed=4), wrap=22)
        bases, inv, total = g.bases, g.invalid, g.total
    else:  # packed words straight from the stream: every uint64 is 32 uniform bases
        nw = (n + 31) // 32
        bases = S.draw(4, np.arange(nw, dtype=np.uint64), 0)
        inv, total = np.zeros(nw, np.uint32), n
    nw = bases.size
    with DevBuf.from_numpy(bases) as d_b, DevBuf.from_numpy(inv) as d_i, DevBuf(nw * 4) as d_f, DevBuf(nw * 4) as d_r:
        for _ in range(3):
            N.check(L.bc_pam_scan_dev(d_b.ptr, d_i.ptr, total, N.ptr(masks), 3, d_f.ptr, d_r.ptr, None))
        N.check(L.bc_dev_sync(None))
        t0 = time.perf_counter()
        for _ in range(iters):
            N.check(L.bc_pam_scan_dev(d_b.ptr, d_i.ptr, total, N.ptr(masks), 3, d_f.ptr, d_r.ptr, None))
        N.check(L.bc_dev_sync(None))
        host_dt = (time.perf_counter() - t0) / iters
        ms = C.c_float(0)
        N.check(L.bc_pam_scan_time_dev(d_b.ptr, d_i.ptr, total, N.ptr(masks), 3, d_f.ptr, d_r.ptr, iters, None, C.byref(ms)))
        dt = ms.value * 1e-3
        fwd = d_f.to_numpy(np.uint32)
        sites = int(np.unpackbits(fwd.view(np.uint8)).sum()) if nw < 10_000_000 else None
    byts = nw * (8 + 4 + 4 + 4)
    out.append({"genome": label, "bp": n, "us_per_scan": round(dt * 1e6, 2), "Gbp_per_s": round(n / dt / 1e9, 2),
                "timing": "HIP events, launches issued from C", "python_loop_us_per_launch": round(host_dt * 1e6, 2),
                "algorithmic_GB_per_s": round(byts / dt / 1e9, 1), "frac_of_8TBps": round(byts / dt / 8e12, 4),
                "fwd_sites": sites})
if True:
    # full site pipeline on the E. coli-size genome (scan + expand + windows + D2H), as pam.find_sites runs it
    g = P.pack_genome(S.genome(4_641_652, seed=4), wrap=22)
    P.find_sites(g, "NGG", 20)
    t0 = time.perf_counter()
    st = P.find_sites(g, "NGG", 20)
    out.append({"pipeline": "find_sites NGG/20 both strands incl. H2D/D2H (two launches: bc_pam_sites_count_dev / _emit_dev)",
                "ms": round((time.perf_counter() - t0) * 1e3, 2), "sites": int(st.start.size)})
    t0 = time.perf_counter()
    st2 = P.find_sites_staged(g, "NGG", 20)
    out.append({"pipeline": "the same from separate stages (six launches, round 2's form)",
                "ms": round((time.perf_counter() - t0) * 1e3, 2), "sites": int(st2.start.size)})
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        P.find_sites(g, "NGG", 20)
    pr.disable()
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(8)
    print(buf.getvalue(), file=sys.stderr)
print(json.dumps(out, indent=1))
