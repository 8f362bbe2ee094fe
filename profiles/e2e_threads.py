#!/usr/bin/env python3
"""Bulk-pass rate of the device ingest against the number of reader threads (BC_STREAM_THREADS), one generated file.
usage: python profiles/e2e_threads.py [n_reads] [threads ...]"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import encoding as E, synth as S, heuristicount as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
threads = [int(x) for x in sys.argv[2:]] or [2, 4, 8, 16]
d = tempfile.mkdtemp(prefix="bc_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
lib = S.library(100_000, 20)
fa = os.path.join(d, "lib.fasta")
with open(fa, "w") as f:
    f.write("".join(f">g{i}\n{s}\n" for i, s in enumerate(E.unpack_kmers(lib, 20))))
rd = os.path.join(d, "reads.reads")
lut = np.frombuffer(b"ACGT", np.uint8)
sh = np.arange(20, dtype=np.uint64)
with open(rd, "wb") as f:
    for a in range(0, n, 2_000_000):
        m = min(2_000_000, n - a)
        w = S.reads(lib, 20, 2, a, m, 1)
        codes = ((((w & np.uint64(0xFFFFFFFF))[:, None] >> sh) & np.uint64(1)) | ((((w >> np.uint64(32))[:, None] >> sh) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
        row = np.empty((m, 46), np.uint8)
        row[:, :5] = lut[0]
        row[:, 5:15] = np.frombuffer(b"CCGGATTCAG", np.uint8)
        row[:, 15:35] = lut[codes]
        row[:, 35:45] = np.frombuffer(b"GTTTTAGAGC", np.uint8)
        row[:, 45] = 10
        f.write(row.tobytes())
for t in threads:
    os.environ["BC_STREAM_THREADS"] = str(t)
    t0 = time.perf_counter()
    r = H.count_barcodes(fa, rd)
    print({"threads": t, "seconds": round(time.perf_counter() - t0, 2), "bulk_Mreads_per_s": round(r["bulk_mreads_per_s"], 1),
           "phases": {k_: round(v, 2) for k_, v in r["seconds"].items()}}, flush=True)
shutil.rmtree(d, ignore_errors=True)
