"""GPU: bc_stream_count_file (raw file bytes -> HBM -> counts) against the host front half (bc_reader_* +
bc_pack_windows + per-chunk grouping, itself pinned by the reference's golden vectors): same documented counts
in the same first-seen order, same undocumented table, same read total — on files that span several device
chunks, with ragged / blank / CRLF lines, N reads, short reads, odd characters inside the window, a last line
without newline, FASTQ quality lines that look like headers, forward and reverse geometry, byte ranges."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L_FLANK, R_FLANK = "CCGGATTCAG", "GTTTTAGAGC"


def _make_file(tmp_path, name, n, k, fastq, rng, lib_s, reverse=False, trailing_newline=True):
    from barcoder_amd.heuristicount import rev_comp
    lines = []
    for i in range(n):
        u = rng.random()
        bc = lib_s[int(rng.integers(0, len(lib_s)))]
        if u < 0.10:                                    # not in the library (usually)
            bc = "".join(rng.choice(list("ACGT"), size=k))
        mid = L_FLANK + bc + R_FLANK
        if reverse:
            mid = rev_comp(mid)
        pre = "".join(rng.choice(list("ACGT"), size=5))
        r = pre + mid + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 9))))
        v = rng.random()
        if v < 0.01:
            p = int(rng.integers(0, len(r)))
            r = r[:p] + "N" + r[p + 1:]
        elif v < 0.02:
            r = r[:int(rng.integers(0, len(r)))]            # cut short: window clamped, flanks usually gone
        elif v < 0.03:
            p = 15 + int(rng.integers(0, k))
            r = r[:p] + "a" + r[p + 1:]                     # odd character inside the window
        elif v < 0.035:
            r = ""                                          # blank line
        elif v < 0.045:
            r = "  " + r + " \t"                            # stripped by the reader
        elif v < 0.05:
            r = r[:len(pre) + len(mid) - 3]                 # right flank cut: fails the flank test
        lines.append(r)
    path = tmp_path / name
    nl = "\n"
    with open(path, "w", newline="") as f:
        for i, r in enumerate(lines):
            end = nl if (trailing_newline or i < n - 1) else ""
            eol = "\r\n" if (i % 97 == 0 and end) else end
            if fastq:
                q = "@" * len(r) if i % 50 == 0 else "+" * len(r) if i % 51 == 0 else "I" * len(r)
                f.write(f"@r{i}\n{r}\n+\n{q}{eol}")
            else:
                f.write(r + eol)
    return str(path)


@pytest.mark.parametrize("fastq,reverse,k,trailing", [(False, False, 20, True), (True, False, 20, False), (False, True, 24, False),
                                                      (True, True, 32, True)])
def test_device_ingest_equals_host_front_half(gpu_device, tmp_path, monkeypatch, fastq, reverse, k, trailing):
    from barcoder_amd import encoding as E, heuristicount as H, synth as S
    rng = np.random.default_rng(11 + k + fastq)
    lib = S.library(3000, k)
    lib_s = E.unpack_kmers(lib, k)
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">g{i}\n{s}\n" for i, s in enumerate(lib_s)))
    n = 420_000 if fastq else 900_000                  # > 2 device chunks of 16 MiB either way
    path = _make_file(tmp_path, "r.fastq" if fastq else "r.reads", n, k, fastq, rng, lib_s, reverse, trailing)
    assert os.path.getsize(path) > 34 << 20
    # (three mismatches: the cascade's list form — workspace per worker stream, released with the stream — inside the ingest)
    for mm in (0, 1) + ((3,) if (k == 20 and not fastq) else ()):
        monkeypatch.setenv("BC_DEVICE_INGEST", "1")
        dev = H.count_barcodes(str(fa), path, device=gpu_device, max_mismatch=mm)
        monkeypatch.setenv("BC_DEVICE_INGEST", "0")
        host = H.count_barcodes(str(fa), path, device=gpu_device, max_mismatch=mm, chunk_reads=100_003)
        assert dev["total_reads"] == host["total_reads"] == n
        assert dev["need_swap"] == host["need_swap"] == reverse
        assert list(dev["doc"].items()) == list(host["doc"].items())          # counts AND first-seen order
        assert sum(dev["doc"].values()) > 0.7 * n
        assert dict(dev["undoc"].items()) == dict(host["undoc"].items())
        assert [k_ for k_, _ in dev["undoc"].items()] == [k_ for k_, _ in host["undoc"].items()]
        assert len(dev["undoc"]) > 1000
        assert dev["undoc"].extra == host["undoc"].extra and len(dev["undoc"].extra) > 0   # the odd windows
    assert dev["bulk_mreads_per_s"] > 0


def test_chunk_size_does_not_change_the_result(gpu_device, tmp_path, monkeypatch):
    """run_file cuts a long file into larger chunks (16 or 32 MiB by the file's size; BC_STREAM_CHUNK_MB overrides):
    the same FASTQ as 3 MiB chunks (a dozen, ragged ends everywhere), as the default and as one 48 MiB chunk gives the
    same documented counts, first-seen order, undocumented table and odd windows."""
    from barcoder_amd import encoding as E, heuristicount as H, synth as S
    k = 20
    rng = np.random.default_rng(77)
    lib = S.library(3000, k)
    lib_s = E.unpack_kmers(lib, k)
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">g{i}\n{s}\n" for i, s in enumerate(lib_s)))
    n = 300_000
    path = _make_file(tmp_path, "r.fastq", n, k, True, rng, lib_s, False, True)
    assert 20 << 20 < os.path.getsize(path) < 48 << 20
    got = {}
    for mb in ("3", None, "48"):
        if mb is None:
            monkeypatch.delenv("BC_STREAM_CHUNK_MB", raising=False)
        else:
            monkeypatch.setenv("BC_STREAM_CHUNK_MB", mb)
        got[mb] = H.count_barcodes(str(fa), path, device=gpu_device, max_mismatch=1)
    monkeypatch.delenv("BC_STREAM_CHUNK_MB", raising=False)
    a = got[None]
    assert a["total_reads"] == n and a["bulk_pass"].startswith("device ingest")
    for mb in ("3", "48"):
        b = got[mb]
        assert b["total_reads"] == n
        assert list(a["doc"].items()) == list(b["doc"].items())
        assert list(a["undoc"].items()) == list(b["undoc"].items())
        assert a["undoc"].extra == b["undoc"].extra and len(a["undoc"].extra) > 0


def _zstd_compress(raw: bytes):
    import ctypes
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(raw))
    out = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(out, cap, raw, len(raw), 1)
    return out.raw[:n]


@pytest.mark.parametrize("ext,fastq,reverse", [(".gz", False, False), (".gz", True, True), (".zst", False, True)])
def test_compressed_input_takes_the_device_path(gpu_device, tmp_path, monkeypatch, ext, fastq, reverse):
    """heuristicount.py:60-72, :100-153: .gz / .zst are first-class inputs.  A compressed file goes through the same
    device path as its plain twin — decoded on the host 16 MiB at a time into the upload buffers — and gives the same
    documented counts in the same first-seen order, undocumented table, odd windows and read total as the plain file on
    the device path and as the compressed file on the host front half; split into parts (chunk c -> part c % n: several
    ranks over one compressed file) the parts add up to the whole; a damaged stream is an error, not a short file."""
    import gzip
    from barcoder_amd import DeviceLibrary, encoding as E, heuristicount as H, ingest as I, synth as S
    rng = np.random.default_rng(29 + fastq)
    k = 20
    lib = S.library(2500, k)
    lib_s = E.unpack_kmers(lib, k)
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">g{i}\n{s}\n" for i, s in enumerate(lib_s)))
    n = 330_000 if fastq else 800_000                    # three device chunks of decoded text either way
    plain = _make_file(tmp_path, "c.fastq" if fastq else "c.reads", n, k, fastq, rng, lib_s, reverse, trailing_newline=not fastq)
    raw = open(plain, "rb").read()
    assert len(raw) > 33 << 20
    comp = plain + ext
    if ext == ".gz":
        with gzip.open(comp, "wb", compresslevel=1) as f:
            f.write(raw)
    else:
        z = _zstd_compress(raw)
        if z is None:
            pytest.skip("libzstd not available")
        open(comp, "wb").write(z)
    monkeypatch.setenv("BC_DEVICE_INGEST", "1")
    a = H.count_barcodes(str(fa), plain, device=gpu_device, max_mismatch=1)
    b = H.count_barcodes(str(fa), comp, device=gpu_device, max_mismatch=1)
    monkeypatch.setenv("BC_DEVICE_INGEST", "0")
    c = H.count_barcodes(str(fa), comp, device=gpu_device, max_mismatch=1, chunk_reads=150_001)
    assert "device ingest" in b["bulk_pass"] and "device ingest" not in c["bulk_pass"], (b["bulk_pass"], c["bulk_pass"])
    for other in (b, c):
        assert other["total_reads"] == a["total_reads"] == n
        assert list(other["doc"].items()) == list(a["doc"].items())
        assert dict(other["undoc"].items()) == dict(a["undoc"].items())
        assert other["undoc"].extra == a["undoc"].extra
    assert sum(a["doc"].values()) > 0.7 * n and len(a["undoc"]) > 500 and len(a["undoc"].extra) > 0
    # parts: chunk c of the decoded stream belongs to part c % 3; keys are read indices in the whole file
    dl = DeviceLibrary(lib, k, gpu_device)
    G = lib.size
    start, Lf, Rf = (5, L_FLANK, R_FLANK)
    if reverse:
        from barcoder_amd.heuristicount import rev_comp
        start, Lf, Rf = 5, rev_comp(R_FLANK), rev_comp(L_FLANK)  # (the window starts at its left flank, as in a forward read)
        assert a["bc_start2"] == 15 and a["need_swap"]

    def run(parts):
        cc, ff = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
        tabs, nn, odd = [], 0, []
        for p_ in parts:
            r = I.stream_count_file(dl.handle, comp, 0, 0, start, Lf, Rf, reverse, 1, "auto", 0, cc, ff, n_threads=3, part=p_)
            tabs.append((r.u_words, r.u_counts, r.u_first))
            nn += r.n_reads
            odd += r.odd
        uw, uc, uf = H.merge_undocumented(tabs)
        return nn, cc, ff, uw, uc, uf, sorted(odd)
    whole, split = run([None]), run([(0, 3), (1, 3), (2, 3)])
    assert whole[0] == split[0] == n and int(whole[1].sum()) == sum(a["doc"].values())
    for x, y in zip(whole[1:6], split[1:6]):
        np.testing.assert_array_equal(x, y)
    assert whole[6] == split[6]
    # a stream cut short is an error
    cut = str(tmp_path / ("cut.fastq" if fastq else "cut.reads")) + ext
    data = open(comp, "rb").read()
    open(cut, "wb").write(data[:len(data) * 2 // 3])
    cc, ff = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
    with pytest.raises(ValueError, match="truncated|damaged"):
        I.stream_count_file(dl.handle, cut, 0, 0, start, Lf, Rf, reverse, 0, "auto", 0, cc, ff)
    with pytest.raises(ValueError, match="byte ranges"):
        N_ = __import__("barcoder_amd._native", fromlist=["x"])
        h = N_.vp()
        import ctypes as C
        N_.check(N_.load().bc_stream_count_file(dl.handle, comp.encode(), 10, 1000, start, Lf.encode(), len(Lf), Rf.encode(), len(Rf),
                                                1 if reverse else 0, 0, N_.ALGOS["auto"], 0, 0, N_.ptr(cc), N_.ptr(ff), C.byref(h)))


def test_stream_byte_ranges_tile_the_file(gpu_device, tmp_path):
    """Record-aligned byte ranges (what the ranks of `--gpus N` get) add up to the whole file: counts summed, keys
    min'ed, undocumented tables merged — for .reads and FASTQ."""
    from barcoder_amd import DeviceLibrary, encoding as E, heuristicount as H, ingest as I, synth as S
    rng = np.random.default_rng(5)
    k = 20
    lib = S.library(2000, k)
    lib_s = E.unpack_kmers(lib, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    for fastq in (False, True):
        path = _make_file(tmp_path, "t.fastq" if fastq else "t.reads", 60_000, k, fastq, rng, lib_s)
        size = os.path.getsize(path)

        def run(cuts):
            G = lib.size
            c, f = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
            tabs, n, odd = [], 0, []
            for a, b in zip(cuts, cuts[1:]):
                lo, hi = I.record_boundary(path, a), I.record_boundary(path, b)
                r = I.stream_count_file(dl.handle, path, lo, hi, 5, L_FLANK, R_FLANK, False, 0, "auto", lo, c, f, n_threads=2)
                tabs.append((r.u_words, r.u_counts, r.u_first))
                n += r.n_reads
                odd += [t for _k, t in r.odd]
            uw, uc, uf = H.merge_undocumented(tabs)
            return n, c, np.argsort(f, kind="stable")[:int((c > 0).sum())], uw, uc, odd
        whole = run([0, size])
        assert whole[0] == 60_000
        for cuts in ([0, size // 3, size // 2 + 7, size], [0, 1, 2, 3, size - 2, size], [0, size // 7 * 3, size]):
            part = run(cuts)
            assert part[0] == whole[0]
            np.testing.assert_array_equal(part[1], whole[1])
            np.testing.assert_array_equal(part[2], whole[2])       # first-seen ORDER survives the range keys
            np.testing.assert_array_equal(part[3], whole[3])
            np.testing.assert_array_equal(part[4], whole[4])
            assert part[5] == whole[5]


def test_undocumented_list_grows_under_concurrent_workers(gpu_device, tmp_path):
    """Bare 20-mers, one per 21-byte line, nearly all of them absent from the library: more undocumented windows
    than the run-wide list is first sized for (span / 24), in a file of several 16 MiB chunks taken by six worker
    threads at once — the list must grow without losing an entry (round 2 grew it under a lock that the other
    workers' launches did not take).  Expected table from numpy; one worker and six must agree with it."""
    from barcoder_amd import DeviceLibrary, encoding as E, ingest as I, synth as S
    k, n = 20, 1_800_000
    lib = S.library(500, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    words = S.draw(77, np.arange(n, dtype=np.uint64), 0) & np.uint64(0x000FFFFF000FFFFF)
    words[::1000] = lib[np.arange(words[::1000].size) % lib.size]      # a few documented reads
    words[5::7] = words[3::7][:words[5::7].size]                       # repeated undocumented words
    sh = np.arange(k, dtype=np.uint64)
    codes = (((words[:, None] >> sh) & np.uint64(1)) | (((words[:, None] >> (sh + np.uint64(32))) & np.uint64(1)) << np.uint64(1))).astype(np.uint8)
    rows = np.empty((n, k + 1), np.uint8)
    rows[:, :k] = np.frombuffer(b"ACGT", np.uint8)[codes]
    rows[:, k] = 10
    path = str(tmp_path / "bare.reads")
    rows.tofile(path)
    assert os.path.getsize(path) == n * 21 > 2 * (16 << 20) and n > os.path.getsize(path) // 24 + 65536
    in_lib = np.isin(words, lib)
    uw, first, cnt = np.unique(words[~in_lib], return_index=True, return_counts=True)
    first_keys = np.flatnonzero(~in_lib)[first]
    for threads in (1, 6):
        c, f = np.zeros(lib.size, np.uint64), np.full(lib.size, np.iinfo(np.uint64).max, np.uint64)
        r = I.stream_count_file(dl.handle, path, 0, os.path.getsize(path), 0, "", "", False, 0, "auto", 0, c, f, n_threads=threads)
        assert r.n_reads == n and int(c.sum()) == int(in_lib.sum())
        order = np.argsort(r.u_words)
        np.testing.assert_array_equal(r.u_words[order], uw)
        np.testing.assert_array_equal(r.u_counts[order], cnt.astype(np.uint64))
        np.testing.assert_array_equal(r.u_first[order], first_keys.astype(np.uint64))


def test_failed_run_releases_its_streams_and_the_library_lives_on(gpu_device, tmp_path):
    """A chunk that fails in the middle of a file (a line longer than the 1 MiB slack, in the third of four chunks, six
    workers) ends the call with the long-line error — and every worker's stream workspace on the library is released on
    the way out, whichever worker failed and whichever left because another one had (round 3 released it on the success
    path only: the entries leaked, counted against the 1 024-stream cap, and a later stream reusing a handle value
    inherited one).  Seventy failed runs later the library holds no workspace and counts a good file as before."""
    import ctypes as C
    from barcoder_amd import DeviceLibrary, _native as N, encoding as E, ingest as I, synth as S
    k = 20
    lib = S.library(2000, k)
    dl = DeviceLibrary(lib, k, gpu_device)
    lib_s = E.unpack_kmers(lib, k)
    rng = np.random.default_rng(5)
    picks = rng.integers(0, lib.size, 60_000)
    good = str(tmp_path / "good.reads")
    with open(good, "w") as f:
        f.write("".join("AC" + lib_s[i] + "GT\n" for i in picks))
    want = np.bincount(picks, minlength=lib.size).astype(np.uint64)
    # five chunks of 16 MiB; the over-long line starts just before the end of the third and runs past that chunk's
    # 1 MiB of slack (a long line that ends inside the slack is an ordinary, if odd, line)
    line = ("AC" + lib_s[0] + "GT\n").encode()
    chunk = 16 << 20
    n_before = (3 * chunk - 100) // len(line)
    bad = str(tmp_path / "bad.reads")
    with open(bad, "wb") as f:
        f.write(line * n_before)
        f.write(b"A" * ((1 << 20) + 4096) + b"\n")
        f.write(line * (chunk // len(line) + 1000))
    assert 4 * chunk < os.path.getsize(bad) < 5 * chunk
    L = N.load()

    def held():
        n = C.c_uint32()
        N.check(L.bc_lib_workspace_count(dl.handle, C.byref(n)))
        return n.value

    def run(path, threads):
        c, f = np.zeros(lib.size, np.uint64), np.full(lib.size, np.iinfo(np.uint64).max, np.uint64)
        r = I.stream_count_file(dl.handle, path, 0, os.path.getsize(path), 2, "", "", False, 1, "auto", 0, c, f, n_threads=threads)
        return r, c

    r, c = run(good, 3)
    np.testing.assert_array_equal(c, want)
    base = held()
    for i in range(70):
        with pytest.raises(ValueError, match="a line longer than"):
            run(bad, 6 if i % 2 else 2)
        assert held() == base, i
    r, c = run(good, 6)
    assert r.n_reads == picks.size and r.u_words.size == 0
    np.testing.assert_array_equal(c, want)
    assert held() == base


def _make_pair_files(tmp_path, n, k, fastq, rng, lib_s, swapped):
    """Mates: read 1 carries L + barcode + R, read 2 the reverse complement of that stretch; with N reads, mates that
    disagree, short reads, odd characters (in one mate, or mirrored in both so that the pair still agrees), blank
    lines.  swapped: the files change roles (file 1 holds the reverse mates)."""
    from barcoder_amd.heuristicount import rev_comp
    r1, r2 = [], []
    for i in range(n):
        bc = lib_s[int(rng.integers(0, len(lib_s)))]
        if rng.random() < 0.10:
            bc = "".join(rng.choice(list("ACGT"), size=k))
        mid = L_FLANK + bc + R_FLANK
        a = "".join(rng.choice(list("ACGT"), size=5)) + mid + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 9))))
        b = "".join(rng.choice(list("ACGT"), size=3)) + rev_comp(mid) + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 7))))
        v = rng.random()
        if v < 0.01:
            p = int(rng.integers(0, len(a)))
            a = a[:p] + "N" + a[p + 1:]
        elif v < 0.02:
            p = int(rng.integers(0, len(b)))
            b = b[:p] + "N" + b[p + 1:]
        elif v < 0.04:   # mates disagree
            other = lib_s[int(rng.integers(0, len(lib_s)))]
            b = b[:3] + rev_comp(L_FLANK + other + R_FLANK) + b[3 + len(mid):]
        elif v < 0.05:
            a = a[:int(rng.integers(0, len(a)))]
        elif v < 0.06:   # odd character inside one mate's window
            p = 15 + int(rng.integers(0, k))
            a = a[:p] + "a" + a[p + 1:]
        elif v < 0.07:   # the same odd character mirrored in both mates: the pair still agrees (undocumented, as a string)
            j = int(rng.integers(0, k))
            a = a[:15 + j] + "a" + a[15 + j + 1:]
            q = 3 + len(R_FLANK) + (k - 1 - j)
            b = b[:q] + "a" + b[q + 1:]
        elif v < 0.075:
            b = ""
        elif v < 0.085:
            a, b = "  " + a + "\t", b + "  "
        r1.append(a)
        r2.append(b)
    if swapped:
        r1, r2 = r2, r1
    paths = []
    for name, reads in (("m1", r1), ("m2", r2)):
        path = tmp_path / (name + (".fastq" if fastq else ".reads"))
        with open(path, "w", newline="") as f:
            for i, r in enumerate(reads):
                f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" if fastq else r + "\n")
        paths.append(str(path))
    return paths


@pytest.mark.parametrize("fastq,swapped,k", [(False, False, 20), (True, True, 20), (False, True, 32)])
def test_paired_device_ingest_equals_host_front_half(gpu_device, tmp_path, monkeypatch, fastq, swapped, k):
    """bc_stream_count_pair (both mate files through the device, pairing and one match over the whole run) against
    the host front half of the same job (bc_reader_* + bc_pack_windows per chunk, itself pinned by the reference's
    paired golden vectors): documented counts in first-seen order, undocumented table, odd pairs, totals."""
    from barcoder_amd import encoding as E, heuristicount as H, synth as S
    rng = np.random.default_rng(23 + k + fastq)
    lib = S.library(3000, k)
    lib_s = E.unpack_kmers(lib, k)
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">g{i}\n{s}\n" for i, s in enumerate(lib_s)))
    n = 260_000 if fastq else 520_000            # more than one 16 MiB chunk per file
    f1, f2 = _make_pair_files(tmp_path, n, k, fastq, rng, lib_s, swapped)
    assert os.path.getsize(f1) > 17 << 20
    for mm in (0, 1):
        monkeypatch.setenv("BC_DEVICE_INGEST", "1")
        # (mm = 1: the per-read window arrays start at a tenth of what the file needs and grow under the workers)
        monkeypatch.setenv("BC_STREAM_EST_PCT", "10" if mm else "125")
        dev = H.count_barcodes(str(fa), f1, f2, device=gpu_device, max_mismatch=mm)
        monkeypatch.setenv("BC_DEVICE_INGEST", "0")
        host = H.count_barcodes(str(fa), f1, f2, device=gpu_device, max_mismatch=mm, chunk_reads=100_003)
        assert dev["total_reads"] == host["total_reads"] == n
        assert dev["need_swap"] == host["need_swap"] == swapped
        assert list(dev["doc"].items()) == list(host["doc"].items())
        assert sum(dev["doc"].values()) > 0.6 * n
        assert dict(dev["undoc"].items()) == dict(host["undoc"].items())
        assert [k_ for k_, _ in dev["undoc"].items()] == [k_ for k_, _ in host["undoc"].items()]
        assert dev["undoc"].extra == host["undoc"].extra and len(dev["undoc"].extra) > 0   # the mirrored odd pairs
    # mate files of different lengths: the reference's error (heuristicount.py:484-486), raised as ValueError
    short = tmp_path / ("short" + (".fastq" if fastq else ".reads"))
    with open(f2) as f, open(short, "w") as g:
        lines = f.readlines()
        g.writelines(lines[:len(lines) - (8 if fastq else 2)])
    monkeypatch.setenv("BC_DEVICE_INGEST", "1")
    with pytest.raises(ValueError, match="Length of reads1 and reads2 must be the same"):
        H.count_barcodes(str(fa), f1, str(short), device=gpu_device)
