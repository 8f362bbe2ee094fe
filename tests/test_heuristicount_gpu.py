"""GPU parity for the process_chunk drop-in and the main() driver against golden
vectors produced by the reference itself."""
import contextlib
import gzip
import io
import json
import os
import types
from collections import Counter

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _load(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        return json.load(f)


def _rc(s):
    return s[::-1].translate(str.maketrans("ATCGN", "TAGCN"))


def test_process_chunk_drop_in(gpu_device):
    """Same 11 positional arguments as heuristicount.py:428-440, same (Counter, int)
    out — key order (first-seen) included."""
    from barcoder_amd.heuristicount import process_chunk
    n = 0
    for c in _load("heuristicount_chunks.json.gz"):
        fwd = {(c["L_fwd"] or "") + b + (c["R_fwd"] or "") for b in c["barcodes"]}
        rev = {(c["L_rev"] or "") + _rc(b) + (c["R_rev"] or "") for b in c["barcodes"]}
        args = (tuple(c["chunk"]), fwd, rev, c["L_fwd_start"], c["L_rev_start"], c["bc_len"], c["L_fwd"], c["R_fwd"],
                c["L_rev"], c["R_rev"], c["need_swap"])
        if "error" in c:
            with pytest.raises(ValueError, match="must be the same"):
                process_chunk(*args)
            continue
        counts, total = process_chunk(*args)
        assert isinstance(counts, Counter) and total == c["n"], c["name"]
        assert list(counts.items()) == [tuple(x) for x in c["counts"]], c["name"]
        n += 1
    assert n >= 21


def test_main_stdout_matches_reference(gpu_device, tmp_path, capsys):
    """End to end: FASTA + FASTQ/.reads files in, `barcode<TAB>count` lines out,
    byte-identical to the reference's stdout (first-seen order)."""
    from barcoder_amd import heuristicount as H
    for s in _load("heuristicount_main.json.gz"):
        if s["name"] == "noflank":
            continue  # the reference's own answer depends on set iteration order here (spurious 1-nt flanks)
        fa = tmp_path / f"{s['name']}.fasta"
        fa.write_text("".join(f">bc{i}\n{b}\n" for i, b in enumerate(s["barcodes"])))

        def write(path, reads):
            with open(path, "w") as f:
                for i, r in enumerate(reads):
                    f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" if s["ext"] == "fastq" else r + "\n")
        f1 = tmp_path / f"{s['name']}_1.{s['ext']}"
        write(f1, s["reads1"])
        f2 = None
        if s["paired"]:
            f2 = tmp_path / f"{s['name']}_2.{s['ext']}"
            write(f2, s["reads2"])
        args = types.SimpleNamespace(fasta_file=str(fa), file1=str(f1), file2=str(f2) if f2 else None)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            H.main(args, device=gpu_device)
        assert buf.getvalue() == s["stdout"], s["name"]
        assert len(s["stdout"].splitlines()) > 50


def test_main_swallows_errors_like_reference(gpu_device, tmp_path, caplog):
    from barcoder_amd import heuristicount as H
    fa = tmp_path / "few.fasta"
    fa.write_text(">a\nACGT\n")
    rd = tmp_path / "r.reads"
    rd.write_text("ACGT\n")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        H.main(types.SimpleNamespace(fasta_file=str(fa), file1=str(rd), file2=None))  # logs, does not raise
    assert buf.getvalue() == ""
    assert "fewer than 10" in caplog.text


def test_config1_cn32_library(gpu_device, oracle):
    """BASELINE configs[0]: the shipped CN-32-zmo library (9,503 unique 32-mers)
    vs 10k synthetic reads with the SURVEY 8d cfg1 layout, against the oracle."""
    import random
    from barcoder_amd.heuristicount import ChunkCounter
    with gzip.open(os.path.join(GOLDEN, "cn32_spacers.txt.gz"), "rt") as f:
        lib = list(dict.fromkeys(f.read().split()))
    assert len(lib) == 9503
    rng = random.Random(1234)
    Lf, Rf = "CCGGATTCAG", "GTTTTAGAGCTAGAAATAGC"
    reads = []
    for _ in range(10_000):
        bc = list(rng.choice(lib))
        u = rng.random()
        for p in rng.sample(range(32), 2 if u > 0.95 else 1 if u > 0.85 else 0):
            bc[p] = rng.choice([c for c in "ACGT" if c != bc[p]])
        r = "".join(rng.choice("ACGT") for _ in range(5)) + Lf + "".join(bc) + Rf
        if rng.random() < 0.01:
            p = rng.randrange(len(r))
            r = r[:p] + "N" + r[p + 1:]
        reads.append(r)
    cc = ChunkCounter(lib, 32, 5, None, Lf, Rf, None, None, False, device=gpu_device)
    got, n = cc.process((reads, None))
    want, n2 = oracle.process_chunk_py((reads, None), lib, 5, None, 32, Lf, Rf, None, None, False)
    assert n == n2 == 10_000
    assert list(got.items()) == list(want.items())
    assert sum(v for k, v in got.items() if not k.endswith("*")) > 8000


def test_cli_twin(gpu_device, tmp_path):
    """`python -m barcoder_amd.heuristicount lib.fasta reads.fastq.gz` prints the reference's stdout."""
    import subprocess
    import sys
    s = next(x for x in _load("heuristicount_main.json.gz") if x["name"] == "fwd_single")
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">bc{i}\n{b}\n" for i, b in enumerate(s["barcodes"])))
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wt") as f:
        for i, r in enumerate(s["reads1"]):
            f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "barcoder_amd.heuristicount", str(fa), str(fq)], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout == s["stdout"]
    assert "Total Reads" in out.stderr and "Documented Barcode Reads" in out.stderr


@pytest.mark.parametrize("k", [40, 120])
def test_barcodes_longer_than_32_nt(gpu_device, oracle, tmp_path, k, monkeypatch):
    """heuristicount.py:75-97 admits barcodes up to 1 000 nt and its set lookup (:449-453) does not care about their
    length; the packed matcher holds 32 bases.  Longer barcodes go through a text hash in the GPU's exact table with
    every hit verified against the text: forward, reverse and paired chunks (N reads, short reads, odd characters,
    unknown barcodes, disagreeing mates) must give the Counter — key order included — of the oracle's restatement
    of process_chunk, and main() the stdout that restatement implies."""
    import random
    from barcoder_amd import heuristicount as H
    rng = random.Random(7 + k)
    lib = list(dict.fromkeys("".join(rng.choice("ACGT") for _ in range(k)) for _ in range(400)))
    Lf, Rf = "CCGGATTCAG", "GTTTTAGAGC"
    Lr, Rr = _rc(Rf), _rc(Lf)
    fwd, mates = [], []
    for i in range(6000):
        bc = rng.choice(lib) if rng.random() > 0.12 else "".join(rng.choice("ACGT") for _ in range(k))
        mid = Lf + bc + Rf
        a = "".join(rng.choice("ACGT") for _ in range(5)) + mid + "".join(rng.choice("ACGT") for _ in range(rng.randrange(6)))
        b = "".join(rng.choice("ACGT") for _ in range(3)) + _rc(mid) + "".join(rng.choice("ACGT") for _ in range(rng.randrange(6)))
        v = rng.random()
        if v < 0.02:
            p = rng.randrange(len(a)); a = a[:p] + "N" + a[p + 1:]
        elif v < 0.04:
            p = rng.randrange(len(b)); b = b[:p] + "N" + b[p + 1:]
        elif v < 0.06:
            a = a[:rng.randrange(len(a))]
        elif v < 0.08:   # an odd character inside the barcode of one mate
            p = 15 + rng.randrange(k); a = a[:p] + "a" + a[p + 1:]
        elif v < 0.10:   # the same odd character mirrored in both mates
            j = rng.randrange(k)
            a = a[:15 + j] + "a" + a[15 + j + 1:]
            q = 3 + len(Rf) + (k - 1 - j); b = b[:q] + "a" + b[q + 1:]
        elif v < 0.12:   # mates that disagree
            b = b[:3] + _rc(Lf + rng.choice(lib) + Rf) + b[3 + len(mid):]
        elif v < 0.13:   # one substitution away from a library entry: undocumented (exact match only)
            p = 15 + rng.randrange(k); a = a[:p] + ("A" if a[p] != "A" else "C") + a[p + 1:]
        fwd.append(a); mates.append(b)
    fset = {Lf + x + Rf for x in lib}
    rset = {Lr + _rc(x) + Rr for x in lib}
    for name, chunk, swap in (("forward", (fwd, None), False), ("reverse", (None, mates), False),
                              ("paired", (fwd, mates), False), ("paired swapped", (mates, fwd), True)):
        got, n = H.process_chunk(chunk, fset, rset, 5, 3, k, Lf, Rf, Lr, Rr, swap, device=gpu_device)
        want, n2 = oracle.process_chunk_py(chunk, lib, 5, 3, k, Lf, Rf, Lr, Rr, swap)
        assert n == n2 == 6000, name
        assert list(got.items()) == list(want.items()), name
        assert sum(v for key, v in got.items() if not key.endswith("*")) > 3500, name
        assert any(key.endswith("*") for key in got), name
    with pytest.raises(ValueError, match="at most 32 nt"):
        H.ChunkCounter(lib, k, 5, 3, Lf, Rf, Lr, Rr, False, device=gpu_device, max_mismatch=1)
    # the driver: sampling finds offset and flanks for long barcodes too, stdout in first-seen order
    fa = tmp_path / "long.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    rd = tmp_path / "long.reads"
    rd.write_text("".join(r + "\n" for r in fwd))
    r = H.count_barcodes(str(fa), str(rd), device=gpu_device)
    assert (r["bc_start1"], r["flanks_fwd"]) == (15, (Lf, Rf))
    want, _ = oracle.process_chunk_py((fwd, None), lib, 5, None, k, Lf, Rf, None, None, False)
    assert list(r["doc"].items()) == [(key, v) for key, v in want.items() if not key.endswith("*")]
    assert dict(r["undoc"].items()) == {key: v for key, v in want.items() if key.endswith("*")}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        H.main(types.SimpleNamespace(fasta_file=str(fa), file1=str(rd), file2=None), device=gpu_device)
    assert buf.getvalue() == "".join(f"{key}\t{v}\n" for key, v in want.items() if not key.endswith("*"))
    # a single uncompressed file goes through the device ingest path (bc_stream_count_file in text-hash mode); the host
    # front half (BC_DEVICE_INGEST=0) and a file of reverse mates (need_swap) must give the same tables, in small chunks too
    rv = tmp_path / "long_rev.reads"
    rv.write_text("".join(m + "\n" for m in mates))
    want_rev, _ = oracle.process_chunk_py((None, mates), lib, 5, 3, k, Lf, Rf, Lr, Rr, False)
    for path, ref in ((rd, want), (rv, want_rev)):
        tables = []
        for env in ("1", "0"):
            monkeypatch.setenv("BC_DEVICE_INGEST", env)
            r = H.count_barcodes(str(fa), str(path), device=gpu_device)
            tables.append((list(r["doc"].items()), dict(r["undoc"].items()), r["total_reads"], r["need_swap"]))
        assert tables[0] == tables[1], path
        assert tables[0][0] == [(key, v) for key, v in ref.items() if not key.endswith("*")], path
        assert tables[0][1] == {key: v for key, v in ref.items() if key.endswith("*")}, path
    assert tables[0][3] is True
    # mate pairs of such barcodes (round 4): both files through the device — the stores keep every window's text beside
    # its hash, pairs need equal hashes and equal texts, pairs with an odd window come back as text — against the host
    # front half and the oracle's process_paired_end, in either file order, plain and gzip-compressed
    import gzip
    want_pair, _ = oracle.process_chunk_py((fwd, mates), lib, 5, 3, k, Lf, Rf, Lr, Rr, False)
    gz1, gz2 = tmp_path / "long.reads.gz", tmp_path / "long_rev.reads.gz"
    for src, dst in ((rd, gz1), (rv, gz2)):
        with gzip.open(dst, "wb", compresslevel=1) as f:
            f.write(open(src, "rb").read())
    for f1, f2 in ((rd, rv), (rv, rd), (gz1, gz2)):
        tables = []
        for env in ("1", "0"):
            monkeypatch.setenv("BC_DEVICE_INGEST", env)
            r = H.count_barcodes(str(fa), str(f1), str(f2), device=gpu_device)
            assert ("device ingest (bc_stream_count_pair)" in r["bulk_pass"]) == (env == "1")
            tables.append((list(r["doc"].items()), dict(r["undoc"].items()), r["total_reads"]))
        assert tables[0] == tables[1], (f1, f2)
        assert tables[0][0] == [(key, v) for key, v in want_pair.items() if not key.endswith("*")], (f1, f2)
        assert tables[0][1] == {key: v for key, v in want_pair.items() if key.endswith("*")}, (f1, f2)
    # several 16 MiB chunks on several worker threads, the miss list grown past its (shrunken) first estimate
    big = tmp_path / "long_big.reads"
    big.write_text("".join(r + "\n" for r in fwd) * (60 if k == 40 else 30))
    monkeypatch.setenv("BC_STREAM_EST_PCT", "1")
    tables = []
    for env in ("1", "0"):
        monkeypatch.setenv("BC_DEVICE_INGEST", env)
        r = H.count_barcodes(str(fa), str(big), device=gpu_device)
        tables.append((list(r["doc"].items()), dict(r["undoc"].items()), r["total_reads"]))
    assert tables[0] == tables[1]
    assert tables[0][2] == 6000 * (60 if k == 40 else 30)


def test_long_barcode_entry_points_refuse_what_they_cannot_do(gpu_device, tmp_path):
    """bc_lib_attach_text needs a k = 32 library of text hashes and texts of 33 ... 1000 characters; the device ingest
    path matches such a library exactly only (heuristicount.py:449-453 is a set lookup): anything else is an error
    through the C ABI, never a silent fallback."""
    import ctypes as C
    from barcoder_amd import _native as N, library as Lb, synth as S
    from barcoder_amd.ingest import stream_count_file
    L = N.load()
    short = Lb.DeviceLibrary(S.library(100, 20), 20, gpu_device)
    text = np.frombuffer(b"ACGT" * 10 * 100, np.uint8)
    with pytest.raises(ValueError, match="bc_lib_attach_text"):
        N.check(L.bc_lib_attach_text(short.handle, N.ptr(text), 40))
    seqs = ["".join("ACGT"[(i * 7 + j * 3 + (i >> j % 5)) % 4] for j in range(40)) for i in range(50)]
    seqs = list(dict.fromkeys(seqs))
    long_lib = Lb.DeviceLibrary.from_strings(seqs, device=gpu_device)
    assert long_lib.long and long_lib.text_k == 40
    with pytest.raises(ValueError, match="bc_lib_attach_text"):
        N.check(L.bc_lib_attach_text(long_lib.handle, N.ptr(text), 32))
    rd = tmp_path / "x.reads"
    rd.write_text("".join("AAAAA" + s + "CC\n" for s in seqs))
    G = long_lib.size
    c64, f64 = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
    with pytest.raises(ValueError, match="matched exactly"):
        stream_count_file(long_lib.handle, str(rd), 0, os.path.getsize(rd), 5, "", "", False, 1, "auto", 0, c64, f64)
    sr = stream_count_file(long_lib.handle, str(rd), 0, os.path.getsize(rd), 5, "", "", False, 0, "auto", 0, c64, f64)
    assert sr.n_reads == len(seqs) and int(c64.sum()) == len(seqs) and sr.u_words.size == 0
    # the paired entry point takes such a library too (texts kept beside the hashes), exact match only
    from barcoder_amd.ingest import stream_count_pair
    with pytest.raises(ValueError, match="matched exactly"):
        stream_count_pair(long_lib.handle, str(rd), str(rd), 5, "", "", 2, "", "", 1, "auto", 0, c64, f64)
