#!/usr/bin/env python3
"""End-to-end timing of barcoder_amd.heuristicount.count_barcodes on a synthetic
.reads file (5 nt + CCGGATTCAG + 20-mer + GTTTTAGAGC, the layout of BASELINE.md
section 2).  usage: python profiles/e2e_bench.py [n_reads] [n_guides] [paired|gz|zst]
`paired` adds a mate file (3 nt + the reverse complement of flank + 20-mer + flank) and counts the pairs;
`gz` / `zst` compress the file first (gzip level 1 / the system libzstd) and also time the bare decoder on it:
the bulk rate of a compressed file is the decoder's."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import encoding as E, synth as S, heuristicount as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
PAIRED = len(sys.argv) > 3 and sys.argv[3] == "paired"
COMP = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] in ("gz", "zst") else None
d = tempfile.mkdtemp(prefix="bc_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
lib = S.library(G, 20)
labels = E.unpack_kmers(lib, 20)
fa = os.path.join(d, "lib.fasta")
with open(fa, "w") as f:
    f.write("".join(f">g{i}\n{s}\n" for i, s in enumerate(labels)))
t0 = time.perf_counter()
rd = os.path.join(d, "reads.reads")
CH = 2_000_000
lut = np.frombuffer(b"ACGT", np.uint8)
rd2 = os.path.join(d, "mates.reads")
f2 = open(rd2, "wb") if PAIRED else None
with open(rd, "wb") as f:
    for a in range(0, n, CH):
        m = min(CH, n - a)
        w = S.reads(lib, 20, 2, a, m, 1)
        lo, hi = (w & np.uint64(0xFFFFFFFF)).astype(np.uint64), (w >> np.uint64(32)).astype(np.uint64)
        sh = np.arange(20, dtype=np.uint64)
        codes = (((lo[:, None] >> sh) & np.uint64(1)) | (((hi[:, None] >> sh) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
        pre = lut[(S.draw(9, np.arange(a, a + m, dtype=np.uint64), 0)[:, None] >> (np.arange(5, dtype=np.uint64) * np.uint64(2)) & np.uint64(3)).astype(np.uint8)]
        row = np.empty((m, 46), np.uint8)
        row[:, :5] = pre
        row[:, 5:15] = np.frombuffer(b"CCGGATTCAG", np.uint8)
        row[:, 15:35] = lut[codes]
        row[:, 35:45] = np.frombuffer(b"GTTTTAGAGC", np.uint8)
        row[:, 45] = 10
        f.write(row.tobytes())
        if f2 is not None:  # the mate: 3 nt, then the reverse complement of flank + barcode + flank
            comp = np.zeros(256, np.uint8)
            comp[list(b"ACGT")] = list(b"TGCA")
            mate = np.empty((m, 44), np.uint8)
            mate[:, :3] = pre[:, :3]
            mate[:, 3:43] = comp[row[:, 5:45][:, ::-1]]
            mate[:, 43] = 10
            f2.write(mate.tobytes())
if f2 is not None:
    f2.close()
print(f"generated {n} reads ({os.path.getsize(rd) / 1e6:.0f} MB) in {time.perf_counter() - t0:.1f} s", file=sys.stderr)
if COMP:
    import ctypes as C
    from barcoder_amd import _native as N
    plain_size = os.path.getsize(rd)
    t0 = time.perf_counter()
    if COMP == "gz":
        import subprocess
        subprocess.check_call(["gzip", "-1", "-k", rd])
        os.remove(rd)
        rd = rd + ".gz"
    else:
        z = C.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        raw = open(rd, "rb").read()
        buf = C.create_string_buffer(z.ZSTD_compressBound(len(raw)))
        m_ = z.ZSTD_compress(buf, len(buf), raw, len(raw), 1)
        os.remove(rd)
        rd = rd + ".zst"
        open(rd, "wb").write(buf.raw[:m_])
        del raw, buf
    print(f"compressed to {os.path.getsize(rd) / 1e6:.0f} MB in {time.perf_counter() - t0:.1f} s", file=sys.stderr)
    # the bare decoder: bc_reader_read_raw into a scratch buffer, nothing else
    L = N.load()
    h = N.vp()
    N.check(L.bc_reader_open(rd.encode(), C.byref(h)))
    scratch = np.empty(16 << 20, np.uint8)
    got, total = C.c_uint64(1), 0
    t0 = time.perf_counter()
    while got.value:
        N.check(L.bc_reader_read_raw(h, N.ptr(scratch), scratch.size, C.byref(got)))
        total += got.value
    dt_dec = time.perf_counter() - t0
    L.bc_reader_close(h)
    assert total == plain_size
    print({"decoder_alone": COMP, "decoded_GB_per_s": round(total / dt_dec / 1e9, 3), "Mreads_per_s": round(n / dt_dec / 1e6, 1)})
# A file that was just written reads 3-4 x slower the FIRST time (tmpfs / page cache: 1.75 s of pread summed over eight
# workers against 0.46-0.55 s on every later pass, profiles/r04_cold_warm_probe.txt) — a property of the box's memory
# management, not of the job.  The timed job therefore sees a file that has been read once ("in the page cache");
# BC_E2E_FIRST_READ=1 times the first read instead.
if not os.environ.get("BC_E2E_FIRST_READ"):
    for path_ in [rd] + ([rd2] if PAIRED else []):
        with open(path_, "rb", buffering=0) as f_:
            buf_ = bytearray(64 << 20)
            while f_.readinto(buf_):
                pass
t0 = time.perf_counter()
r = H.count_barcodes(fa, rd, rd2 if PAIRED else None)
dt = time.perf_counter() - t0
print({"file": "first read after writing" if os.environ.get("BC_E2E_FIRST_READ") else "read once before (page cache)"})
print({"reads": n, "guides": G, "seconds": round(dt, 2), "Mreads_per_s": round(n / dt / 1e6, 2),
       "documented_reads": sum(r["doc"].values()), "undocumented_reads": r["undoc"].total(),
       "offset": r["bc_start1"], "flanks": r["flanks_fwd"]})

print({"bulk_pass": r["bulk_pass"],
       "bulk_Mreads_per_s": round(r["bulk_mreads_per_s"], 1), "bulk_GB_per_s_of_file": round(r["bulk_mreads_per_s"] * 46e6 / 1e9, 2),
       "undocumented_barcodes": len(r["undoc"]), "seconds": {k_: round(v, 2) for k_, v in r["seconds"].items()}})
if os.environ.get("BC_E2E_TWICE"):  # the same job again in this process: what of the first run was first-use cost
    t0 = time.perf_counter()
    r2 = H.count_barcodes(fa, rd, rd2 if PAIRED else None)
    dt2 = time.perf_counter() - t0
    print({"second_run_seconds": round(dt2, 3), "Mreads_per_s": round(n / dt2 / 1e6, 2),
           "seconds": {k_: round(v, 3) for k_, v in r2["seconds"].items()}})
t0 = time.perf_counter()
lines = H.summary_lines(r, fa)
print({"summary_lines_s": round(time.perf_counter() - t0, 2), "top_undocumented": lines[-1].strip()})
import shutil
shutil.rmtree(d, ignore_errors=True)
