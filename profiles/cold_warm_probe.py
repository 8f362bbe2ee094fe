import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from barcoder_amd import encoding as E, synth as S, heuristicount as H
n = 100_000_000; G = 100_000
d = tempfile.mkdtemp(prefix="bc_cw_", dir="/dev/shm")
try:
    lib = S.library(G, 20)
    fa = os.path.join(d, "lib.fasta")
    open(fa, "w").write("".join(f">g{i}\n{s}\n" for i, s in enumerate(E.unpack_kmers(lib, 20))))
    rd = os.path.join(d, "reads.reads")
    lut = np.frombuffer(b"ACGT", np.uint8); sh = np.arange(20, dtype=np.uint64)
    with open(rd, "wb") as f:
        for a in range(0, n, 2_000_000):
            m = min(2_000_000, n - a)
            w = S.reads(lib, 20, 2, a, m, 1)
            codes = (((w[:, None] >> sh) & np.uint64(1)) | (((w[:, None] >> (sh + np.uint64(32))) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
            row = np.empty((m, 46), np.uint8)
            row[:, :5] = lut[(S.draw(9, np.arange(a, a + m, dtype=np.uint64), 0)[:, None] >> (np.arange(5, dtype=np.uint64) * np.uint64(2)) & np.uint64(3)).astype(np.uint8)]
            row[:, 5:15] = np.frombuffer(b"CCGGATTCAG", np.uint8); row[:, 15:35] = lut[codes]; row[:, 35:45] = np.frombuffer(b"GTTTTAGAGC", np.uint8); row[:, 45] = 10
            f.write(row.tobytes())
    # a second copy of the file: does the FILE's first read cost, or the process's first run?
    rd2 = os.path.join(d, "reads2.reads"); shutil.copyfile(rd, rd2)
    os.environ["BC_STREAM_PROF"] = "1"
    for label, path in (("run 1 (file A)", rd), ("run 2 (file A)", rd), ("run 3 (file B, never read)", rd2), ("run 4 (file B)", rd2)):
        t0 = time.perf_counter(); r = H.count_barcodes(fa, path); dt = time.perf_counter() - t0
        print(label, "total %.2f s" % dt, {k: round(v, 3) for k, v in r["seconds"].items()}, "bulk %.0f Mreads/s" % r["bulk_mreads_per_s"], flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
