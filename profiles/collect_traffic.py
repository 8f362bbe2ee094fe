#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of profiles/run_prof.sh into
profiles/traffic.json (read by bench.py for roofline.traffic).

Corrections (MI355X_MICROARCH.md, section HBM): on gfx950 FETCH_SIZE reports half
the bytes of a 16-byte-per-lane coalesced streaming read, WRITE_SIZE is exact.
k_hist / k_hist_part (and the stand-alone k_match_hash) stream their input with
16-byte-per-lane loads, so their FETCH_SIZE is doubled (calibration on known byte
counts: k_match_hash raw 436 MB for an 800 MB read array, k_hist raw 1.0 GB for
5 x 400 MB).  k_match_wq / k_match_fused mix the 16-byte-per-lane read stream (8 B
per read, counted at half: + 4 B per read added back) with table pieces served from
L2 / Infinity Cache (FETCH_SIZE counts Infinity-Cache hits as fetches); with the read-stream prefetch the stream comes in
through scalar loads that are counted in full (see below).
k_first_refine reads 4-byte words (uncalibrated width: taken as counted).
usage: python profiles/collect_traffic.py profiles/r01_rocprof_summary.json reads algo max_mm
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd.stamp import match_source_stamp
src, reads, algo, mm = sys.argv[1], int(float(sys.argv[2])), sys.argv[3], int(sys.argv[4])
d = json.load(open(src))["pmc"]
rows, total = {}, 0.0
for name, e in d.items():
    if not name.startswith(("k_match", "k_hist", "k_first_refine")):
        continue
    f = e.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0) * 1024
    # the matcher runs with and without the tally in one bench invocation (kernel-only timing launches write no
    # records): the launches of a counting pass are the larger ones
    ws = e.get("WRITE_SIZE", {})
    w = ws.get("max_per_dispatch", ws.get("mean_per_dispatch", 0.0)) * 1024
    corr = 2.0 if name.startswith(("k_match_hash", "k_hist")) else 1.0
    add = 4.0 * reads if name.startswith(("k_match_wq", "k_match_wave", "k_match_fused")) else 0.0
    note = None
    if name.startswith("k_match_wq"):
        # With the read-stream prefetch (ladder v22) the stream reaches the L2 through 4-byte scalar loads, one per
        # 64-byte sector, which FETCH_SIZE counts in full; only the demand loads that still miss (their sectors were
        # evicted again) are 16-byte-per-lane vector fetches counted at half.  Their share comes from the L2 miss
        # counter: misses = table misses (the count without prefetch, 0.32e7 per 1e8 reads) + one per prefetched sector
        # + the demand misses.  SQ_INSTS_SMEM tells whether the prefetch ran (32 scalar loads per tile more).
        smem = e.get("SQ_INSTS_SMEM", {}).get("mean_per_dispatch", 0.0)
        tiles = reads / 256.0
        if smem > 40.0 * tiles:  # 29.6 scalar instructions per tile without the prefetch, 61.6 with it
            sectors = reads / 8.0
            miss = e.get("TCC_MISS_sum", {}).get("mean_per_dispatch", 0.0)
            demand_miss = min(max(miss - sectors - 0.032 * reads, 0.0), sectors)
            add = 32.0 * demand_miss  # the uncounted half of each 64-byte demand fetch
            note = "read-stream prefetch on: %.0f %% of the stream's sectors still miss on demand" % (100.0 * demand_miss / sectors)
    rows[name] = {"fetch_raw_bytes": f, "fetch_correction": corr, "fetch_added_bytes": add, "write_bytes": w,
                  "hbm_bytes": f * corr + add + w}
    if note:
        rows[name]["note"] = note
    total += f * corr + add + w
out = {"reads": reads, "algo": algo, "max_mismatch": mm, "hbm_bytes_per_pass": total, "per_kernel": rows,
       "algorithmic_bytes_per_pass": 13 * reads, "source": "profiles/" + os.path.basename(src),
       # what the counters were collected on: bench.py quotes the figure only for the same kernel sources and kernel
       "match_source_stamp": match_source_stamp(),
       "dominant_kernel": next((k for k in rows if k.startswith("k_match")), None),
       "note": "FETCH_SIZE counts Infinity-Cache hits too; the match kernel's table misses are included as fetched bytes"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
