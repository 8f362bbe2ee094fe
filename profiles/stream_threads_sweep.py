#!/usr/bin/env python3
"""Bulk-pass rate of the device ingest against the number of worker threads (BC_STREAM_THREADS) and the chunk size
(BC_STREAM_CHUNK_MB), on one 1e8-read .reads file (4.6 GB, page cache), one process: the file is written once and
count_barcodes runs once per setting (the first, untimed, run takes the start-up costs).
usage: python profiles/stream_threads_sweep.py [n_reads]"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import encoding as E, synth as S, heuristicount as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
G = 100_000
d = tempfile.mkdtemp(prefix="bc_sweep_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    lib = S.library(G, 20)
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
            codes = (((w[:, None] >> sh) & np.uint64(1)) | (((w[:, None] >> (sh + np.uint64(32))) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
            row = np.empty((m, 46), np.uint8)
            row[:, :5] = lut[(S.draw(9, np.arange(a, a + m, dtype=np.uint64), 0)[:, None] >> (np.arange(5, dtype=np.uint64) * np.uint64(2)) & np.uint64(3)).astype(np.uint8)]
            row[:, 5:15] = np.frombuffer(b"CCGGATTCAG", np.uint8)
            row[:, 15:35] = lut[codes]
            row[:, 35:45] = np.frombuffer(b"GTTTTAGAGC", np.uint8)
            row[:, 45] = 10
            f.write(row.tobytes())
    H.count_barcodes(fa, rd)  # start-up
    for mb in ("16", "32"):
        for t in ("4", "6", "8", "12", "16", "24"):
            os.environ["BC_STREAM_THREADS"], os.environ["BC_STREAM_CHUNK_MB"] = t, mb
            best = 0.0
            for _ in range(2):
                r = H.count_barcodes(fa, rd)
                best = max(best, r["bulk_mreads_per_s"])
            print(f"chunk {mb} MiB threads {t}: bulk {best:.0f} Mreads/s = {best * 46e6 / 1e9:.1f} GB/s of file", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
