"""CPU: host-side logic of the package (no GPU, no compute calls into the
library) against golden vectors from the reference, and the C ABI surface."""
import ctypes as C
import gzip
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _load(name):
    path = os.path.join(GOLDEN, name)
    with (gzip.open(path, "rt") if name.endswith(".gz") else open(path)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def scenarios(tmp_path_factory):
    d = tmp_path_factory.mktemp("hc")
    out = []
    for s in _load("heuristicount_main.json.gz"):
        fa = d / f"{s['name']}.fasta"
        fa.write_text("".join(f">bc{i}\n{b}\n" for i, b in enumerate(s["barcodes"])))

        def write(path, reads):
            with open(path, "w") as f:
                for i, r in enumerate(reads):
                    f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" if s["ext"] == "fastq" else r + "\n")
        f1 = d / f"{s['name']}_1.{s['ext']}"
        write(f1, s["reads1"])
        f2 = None
        if s["paired"]:
            f2 = d / f"{s['name']}_2.{s['ext']}"
            write(f2, s["reads2"])
        out.append((s, str(fa), str(f1), str(f2) if f2 else None))
    return out


def test_header_symbols_exported():
    """Every function include/barcoder.h declares is exported and bound."""
    from barcoder_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "barcoder.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?(bc_\w+)\(", hdr, re.M))
    assert len(declared) >= 25
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    L = N.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.bc_version() == 100


def test_no_gpu_fails_loudly():
    import barcoder_amd
    from barcoder_amd import synth as S
    if barcoder_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(barcoder_amd.BarcoderNativeError, match="no CPU path"):
        barcoder_amd.DeviceLibrary(S.library(100, 20), 20)


def test_pack_roundtrip_and_revcomp(oracle):
    from barcoder_amd import encoding as E
    rng = np.random.default_rng(1)
    for k in (1, 12, 20, 31, 32):
        seqs = ["".join(rng.choice(list("ACGT"), k)) for _ in range(200)]
        p, f = E.pack_kmers(seqs, k)
        assert not f.any()
        assert E.unpack_kmers(p, k) == seqs
        assert E.unpack_kmers(E.revcomp_kmers(p, k), k) == [oracle.rev_comp(s) for s in seqs]
    p, f = E.pack_kmers(["ACGN", "acgt", "ACGT"], 4)
    assert f.tolist() == [1, 1, 0] and (p[:2] >> np.uint64(63)).tolist() == [1, 1]
    with pytest.raises(ValueError, match="same length"):
        E.pack_kmers(["ACG", "ACGT"])


def test_pack_windows_matches_oracle_statuses(oracle):
    """bc_pack_windows (host C++) vs oracle.c's window rules on the golden chunk inputs."""
    from barcoder_amd import encoding as E, _native as N
    done = 0
    for c in _load("heuristicount_chunks.json.gz"):
        if "error" in c or (c["chunk"][0] and c["chunk"][1]):
            continue
        a, b = c["chunk"]
        r1, r2 = (b, a) if c["need_swap"] else (a, b)
        reads, start, L, R, rev = (r1, c["L_fwd_start"], c["L_fwd"], c["R_fwd"], False) if r1 else \
                                  (r2, c["L_rev_start"], c["L_rev"], c["R_rev"], True)
        text, off = E.reads_to_text(reads)
        packed, status = E.pack_windows(text, off, start, L or "", R or "", c["bc_len"], rev)
        _, out = oracle.process_chunk_single_c(reads, c["barcodes"], start, L, R, rev)
        np.testing.assert_array_equal(status == N.WIN_SKIP, out == -2)
        ok = status == N.WIN_OK
        assert ((packed >> np.uint64(63)) == 1)[~ok].all() or c["bc_len"] == 32
        # packed OK windows decode to the barcode slice the oracle reports
        k = c["bc_len"]
        seqs = E.unpack_kmers(packed[ok] & E.kmer_mask(k), k)
        for i, s in zip(np.flatnonzero(ok)[:50], seqs[:50]):
            w = reads[i][start:start + len(L or "") + k + len(R or "")]
            q = w[len(L or ""):len(w) - len(R or "")]
            assert s == (oracle.rev_comp(q) if rev else q)
        done += 1
    assert done >= 12


def test_hash_windows_follow_the_packed_window_rules(oracle):
    """bc_hash_windows (barcodes of more than 32 nt) applies bc_pack_windows' N-skip / slice / flank rules: on the
    golden chunk inputs (k <= 32, where both exist) the statuses are identical, the window text is the barcode slice
    the oracle reports (reverse-complemented for the reverse orientation), equal texts hash equally and bc_hash_kmers
    gives the library the same hashes."""
    from barcoder_amd import encoding as E, _native as N
    done = 0
    for c in _load("heuristicount_chunks.json.gz"):
        if "error" in c or (c["chunk"][0] and c["chunk"][1]):
            continue
        a, b = c["chunk"]
        r1, r2 = (b, a) if c["need_swap"] else (a, b)
        reads, start, L, R, rev = (r1, c["L_fwd_start"], c["L_fwd"], c["R_fwd"], False) if r1 else \
                                  (r2, c["L_rev_start"], c["L_rev"], c["R_rev"], True)
        k = c["bc_len"]
        text, off = E.reads_to_text(reads)
        _packed, status = E.pack_windows(text, off, start, L or "", R or "", k, rev)
        h, st2, wins = E.hash_windows(text, off, start, L or "", R or "", k, rev)
        np.testing.assert_array_equal(status, st2)
        ok = np.flatnonzero(st2 == N.WIN_OK)
        assert ok.size > 0
        texts = [bytes(wins[i]).decode() for i in ok]
        for i, s_ in zip(ok[:80], texts[:80]):
            w = reads[i][start:start + len(L or "") + k + len(R or "")]
            q = w[len(L or ""):len(w) - len(R or "")]
            assert s_ == (oracle.rev_comp(q) if rev else q)
        np.testing.assert_array_equal(h[ok], E.hash_kmers(texts, k))
        assert (h[st2 != N.WIN_OK] == 0).all()
        done += 1
    assert done >= 12
    # longer than the packed form holds: distinct texts, distinct hashes; the hash depends on every character
    import random
    rng = random.Random(3)
    seqs = ["".join(rng.choice("ACGT") for _ in range(120)) for _ in range(2000)]
    hs = E.hash_kmers(seqs, 120)
    assert np.unique(hs).size == 2000
    one = list(seqs[0])
    for p_ in (0, 7, 8, 63, 64, 119):
        t = one.copy()
        t[p_] = "A" if t[p_] != "A" else "C"
        assert E.hash_kmers(["".join(t)], 120)[0] != hs[0]


def test_read_fasta_and_chunks(tmp_path, scenarios):
    from barcoder_amd import heuristicount as H
    s, fa, f1, f2 = scenarios[0]
    assert H.read_fasta(fa) == set(s["barcodes"])
    chunks = list(H.read_in_chunks(f1, None, chunk_size=1000))
    assert [len(c[0]) for c in chunks] == [1000, 1000, 1000] and chunks[0][1] is None
    assert sum((c[0] for c in chunks), []) == s["reads1"]
    with pytest.raises(ValueError):
        H.read_fasta(str(tmp_path / "x.txt"))
    with pytest.raises(ValueError):
        list(H.read_in_chunks(str(tmp_path / "x.fq")))
    with pytest.raises(ValueError, match="fewer than 10"):
        H.validate_barcodes({"ACGT"})
    gz = tmp_path / "r.reads.gz"
    with gzip.open(gz, "wt") as f:
        f.write("ACGT\nTTTT\n")
    assert list(H.read_in_chunks(str(gz))) == [(["ACGT", "TTTT"], None)]


def test_sample_data_and_flanks_match_reference(scenarios):
    from barcoder_amd import heuristicount as H
    for s, fa, f1, f2 in scenarios:
        lib = H.read_fasta(fa)
        n_novel, off1, off2, s1, s2, observed, need_swap, n_chunks = H.sample_data(f1, f2, lib, s["paired"])
        want = s["sample_data"]
        got = dict(n_novel=n_novel, offset1=off1, offset2=off2, n_valid1=len(s1 or ()), n_valid2=len(s2 or ()),
                   n_observed=len(observed), need_swap=need_swap, num_chunks=n_chunks)
        assert got == want, s["name"]
        if s["flanks1"] is not None and s["name"] != "noflank":  # noflank: ties depend on set order
            assert list(H.find_flanks(s1, off1, s["k"])) == s["flanks1"], s["name"]
        if s["flanks2"] is not None:
            assert list(H.find_flanks(s2, off2, s["k"])) == s["flanks2"], s["name"]


def test_barcode_library_api(tmp_path):
    from barcoder_amd import BarCodeLibrary, BarCodeLibraryError
    with gzip.open(os.path.join(GOLDEN, "cn32_spacers.txt.gz"), "rt") as f:
        spacers = f.read().split()
    tsv = tmp_path / "lib.tsv"
    tsv.write_text("name\tspacer\n" + "".join(f"g{i}\t{s}\n" for i, s in enumerate(spacers)))
    lib = BarCodeLibrary(str(tsv), column="spacer")
    assert lib.size == 9503 == len(lib.barcodes) and len(lib.ordered) == 9503
    assert lib.ordered[0] == spacers[0]
    lib.add("A" * 32)
    assert lib.size == 9504
    lib.remove("A" * 32)
    with pytest.raises(KeyError):
        lib.remove("A" * 32)
    with pytest.raises(BarCodeLibraryError):
        BarCodeLibrary(str(tsv))  # no column
    with pytest.raises(BarCodeLibraryError):
        BarCodeLibrary(str(tmp_path / "x.csv"))
    fa = tmp_path / "l.fasta"
    fa.write_text(">a\nACGT\nAC\n>b\nTTTT\n")
    assert BarCodeLibrary(str(fa)).ordered == ["ACGTAC", "TTTT"]
    assert BarCodeLibrary(barcodes=["AC", "AC", "GT"]).ordered == ["AC", "GT"]


def test_synth_streams_are_stable():
    """Pin the seeded streams (bench inputs) so CPU, GPU and later rounds agree."""
    from barcoder_amd import synth as S
    lib = S.library(1000, 20)
    assert lib[:2].tolist() == [4144102275729808, 593813589130682] and np.unique(lib).size == 1000
    r = S.reads(lib, 20, 2, 0, 5, 1)
    assert r.tolist() == [804503209419294, 3293299318415417, 1716754263489503, 60735132603916, 1130907839570119]
    assert S.reads(lib, 20, 2, 3, 2, 1).tolist() == r[3:].tolist()
    # the mixture is what SURVEY 8d says: ~60/20/10/5/5
    from barcoder_amd import encoding as E
    n = 40000
    rr = S.reads(lib, 20, 2, 0, n, 1)
    g = lib[S.guide_index(2, np.arange(n, dtype=np.uint64), 1000).astype(np.int64)]
    x = rr ^ g
    d = np.array([bin(int((v & np.uint64(0xFFFFFFFF)) | (v >> np.uint64(32)))).count("1") for v in x])
    frac = [(d == i).mean() for i in range(4)]
    assert abs(frac[0] - 0.60) < 0.02 and abs(frac[1] - 0.20) < 0.02 and abs(frac[2] - 0.10) < 0.01
    assert len(S.genome(1000)) == 1000 and set(S.genome(1000)) <= set("ACGT")


def test_mismatch_module_matches_reference(capsys):
    """mismatch.py mirror: y_pred values (float64) and the generated variant tables."""
    from barcoder_amd import mismatch as M
    m = _load("mismatch.json")
    params = m["params"]
    for o, v, want in m["pairs"]:
        got = M.calculate_y_pred(o, v, params["GC_content"], params)
        assert (got is None and want is None) or abs(got - want) < 1e-12
        mask = sum(1 << i for i, (a, b) in enumerate(zip(o, v)) if a != b)
        got2 = M.y_pred_from_mask(o, v, mask, params)
        assert (got2 is None and want is None) or abs(got2 - want) < 1e-12
    for t in m["tables"]:
        M.generate_mismatches([t["spacer"]], 0.0, 1.0, 0.1, params, t["spacer"])
        assert capsys.readouterr().out == t["tsv"]


def test_targets_helpers_match_oracle(oracle):
    from barcoder_amd import targets as T
    assert T.get_diff("ACGT", "ACGT") is None and T.get_diff("ACGT", "ACCA") == "C3G,A4T"
    assert T.diff_from_mask("ACGT", "ACCA", 0b1100) == "C3G,A4T"
    for a, b, n in [(0, 32, 100), (90, 122, 100), (-10, 22, 100), (68, 100, 100)]:
        assert T.get_coords(a, b, n) == oracle.get_coords(a, b, n)
    assert T.pam_matches("NGG", "AGG") and not T.pam_matches("NGG", "AGA") and not T.pam_matches("NGG", None)
    assert T.pam_matches("NNN", "xyz") and T.get_overlap(0, 10, 5, 20) == 5 and T.get_offset("R", 1, 9, 0, 20) == 11


def test_native_reader_matches_python_reader(tmp_path):
    """bc_reader_* vs heuristicount.read_in_chunks (itself golden-tested) on FASTQ /
    .reads, plain / gz / zst, CRLF, missing final newline, truncated record, blank lines."""
    import ctypes
    import random
    from barcoder_amd import heuristicount as H, ingest
    rng = random.Random(1)
    reads = ["".join(rng.choice("ACGTN") for _ in range(rng.randrange(0, 120))) for _ in range(3000)]

    def write(path, kind, trailing, crlf):
        nl = "\r\n" if crlf else "\n"
        op = gzip.open if path.endswith(".gz") else open
        with op(path, "wt", newline="") as f:
            for i, r in enumerate(reads):
                end = "" if (i == len(reads) - 1 and not trailing) else nl
                f.write(f"@r{i}{nl}{r}{nl}+{nl}{'I' * len(r)}{end}" if kind == "fastq" else r + end)

    for kind in ("fastq", "reads"):
        for gz in (False, True):
            for trailing in (True, False):
                for crlf in (False, True):
                    p = str(tmp_path / (f"x.{kind}" + (".gz" if gz else "")))
                    write(p, kind, trailing, crlf)
                    assert ingest.read_all(p) == sum((c[0] for c in H.read_in_chunks(p)), [])
    p = str(tmp_path / "t.fastq")
    open(p, "w").write("@a\nACGT\n+\nIIII\n@b\nGGCC")
    assert ingest.read_all(p) == ["ACGT", "GGCC"]
    with pytest.raises(ValueError, match="Unsupported file type"):
        ingest.NativeReader(str(tmp_path / "x.fq"))
    with pytest.raises(ValueError):
        ingest.NativeReader(str(tmp_path / "missing.reads"))
    # tiny text buffer: stitched pieces, exact read counts
    r = ingest.NativeReader(str(tmp_path / "x.reads"), max_reads=700, bytes_per_read=8)
    t, o = r.next(700)
    assert o.size - 1 == 700 and t.size == int(o[-1]) == sum(len(x) for x in reads[:700])
    assert sum(o2.size - 1 for _t, o2 in r) == len(reads) - 700
    # .zst through the system libzstd, when it is there
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return
    raw = ("\n".join(reads) + "\n").encode()
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(cap)
    m = z.ZSTD_compress(dst, cap, raw, len(raw), 3)
    pz = tmp_path / "x.reads.zst"
    pz.write_bytes(dst.raw[:m])
    assert ingest.read_all(str(pz)) == reads


def test_native_sampling_matches_reference(scenarios):
    """bc_sample_data / bc_sample_flanks vs the golden sample_data / find_flanks results."""
    from barcoder_amd import heuristicount as H
    for s, fa, f1, f2 in scenarios:
        smp = H.NativeSample(f1, f2, H.read_fasta_ordered(fa), s["paired"])
        got = dict(n_novel=smp.n_novel, offset1=smp.offset1, offset2=smp.offset2, n_valid1=smp.n_valid1,
                   n_valid2=smp.n_valid2, n_observed=smp.n_observed, need_swap=smp.need_swap,
                   num_chunks=smp.num_chunks)
        assert got == s["sample_data"], s["name"]
        if s["name"] != "noflank":  # ties there depend on set order in the reference itself
            f_1 = smp.flanks(0, smp.offset1, s["k"])
            f_2 = smp.flanks(1, smp.offset2, s["k"])
            assert list(f_1) == (s["flanks1"] or [None, None]), s["name"]
            assert list(f_2) == (s["flanks2"] or [None, None]), s["name"]
        smp.close()


def test_c_abi_error_mapping():
    """Status codes -> the exception classes the reference's Python code raises."""
    import ctypes as C
    from barcoder_amd import _native as N, encoding as E
    L = N.load()
    out = np.zeros(4, np.uint64)
    assert L.bc_pack_kmers(b"ACGT", 1, 0, 4, N.ptr(out), None) == N.BC_EINVAL
    assert b"bc_pack_kmers" in L.bc_last_error()
    with pytest.raises(ValueError, match="bc_pack_kmers"):
        N.check(L.bc_pack_kmers(b"ACGT", 1, 33, 33, N.ptr(out), None))
    with pytest.raises(ValueError):
        N.check(L.bc_lib_create(None, 0, 20, 0, C.byref(N.vp())))
    with pytest.raises(ValueError, match="never match"):
        E.pam_masks("NGR")
    assert E.pam_masks("NGR", iupac=True).tolist() == [15, 4, 5]
    off = np.array([0, 4], np.uint64)
    with pytest.raises(ValueError, match="bc_pack_windows"):
        E.pack_windows(b"ACGT", off, -1, "", "", 4)
    # windows: N anywhere skips; flanks checked on the clamped python slice; odd characters flagged
    text, off = E.reads_to_text(["TTACGTGG", "TTACNTGG", "TTACGT", "TTAcGTGG", "GGACGTGG", ""])
    packed, status = E.pack_windows(text, off, 0, "TT", "GG", 4)
    assert status.tolist() == [N.WIN_OK, N.WIN_SKIP, N.WIN_SKIP, N.WIN_ODD, N.WIN_SKIP, N.WIN_SKIP]
    assert E.unpack_kmers(packed[:1], 4) == ["ACGT"]
    packed_r, _ = E.pack_windows(text, off, 0, "TT", "GG", 4, reverse=True)
    assert E.unpack_kmers(packed_r[:1], 4) == ["ACGT"]  # ACGT is its own reverse complement
    text, off = E.reads_to_text(["TTAAGTGG"])
    assert E.unpack_kmers(E.pack_windows(text, off, 0, "TT", "GG", 4, reverse=True)[0], 4) == ["ACTT"]


def test_design_filters():
    """design_guides.py:108-310 on a hand-made table: every filter, the NaN rules of the pandas round trip
    (None == x False, None != x True, groupby drops None, sort puts None last, numeric None -> 0)."""
    from barcoder_amd import design as D
    cols = ["spacer", "locus_tag", "gene", "chr", "mismatches", "target", "tar_start", "tar_end", "offset", "overlap",
            "sp_dir", "tar_dir", "note"]

    def row(sp, tag, chr_, a, off, ov, sd, td, note, mm=0, target="acgt"):
        return dict(spacer=sp, locus_tag=tag, gene=tag and tag.lower(), chr=chr_, mismatches=mm, target=target,
                    tar_start=a, tar_end=None if a is None else a + 4, offset=off, overlap=ov, sp_dir=sd, tar_dir=td,
                    note=note)
    rows = [
        row("AAAA", "G1", "c1", 10, 0, 4, "F", "F", "1 site, 1 gene"),
        row("CCCC", "G1", "c1", 14, 4, 4, "R", "F", "1 site, 1 gene"),
        row("GGGG", "G1", "c1", 18, 8, 3, "F", "F", "2 sites, 1 gene, 1 intergenic"),
        row("GGGG", None, "c1", 90, None, None, "F", None, "2 sites, 1 gene, 1 intergenic"),
        row("TTTT", "G2", "c2", 5, 2, 4, "F", "F", "1 site, 2 genes"),
        row("TTTT", "G3", "c2", 5, 30, 4, "F", "R", "1 site, 2 genes"),
        row("ACAC", "G2", "c2", 25, 22, 4, "F", "F", "1 site, 1 gene", mm=1),
        row("NONE", None, None, None, None, None, None, None, "0 sites", target=None),
    ]
    sp = lambda out: [r["spacer"] for r in out[1]]
    base = D.filter_guides(cols, rows, 4)
    assert base[1][0]["target"] == "ACGT"                                   # upper-cased (:110)
    assert sp(base) == ["AAAA", "CCCC", "GGGG", "GGGG", "TTTT", "TTTT", "ACAC", "NONE"]  # chr, tar_start, ..., None last
    assert base[1][3]["offset"] == 0 and base[1][3]["locus_tag"] is None     # numeric None -> 0, strings stay None
    assert base[1][7]["tar_start"] == 0 and base[1][7]["chr"] is None
    assert sp(D.filter_guides(cols, rows, 4, orientation="forward")) == ["AAAA", "GGGG", "TTTT", "ACAC"]
    assert sp(D.filter_guides(cols, rows, 4, orientation="reverse")) == ["CCCC", "GGGG", "TTTT", "NONE"]
    assert sp(D.filter_guides(cols, rows, 4, omit_offtargets=True)) == ["AAAA", "CCCC", "TTTT", "TTTT", "ACAC"]
    assert sp(D.filter_guides(cols, rows, 4, mismatches=1)) == ["AAAA", "CCCC", "GGGG", "GGGG", "TTTT", "TTTT", "NONE"]
    assert sp(D.filter_guides(cols, rows, 4, omit_ambiguous=True)) == ["AAAA", "CCCC", "ACAC"]
    assert sp(D.filter_guides(cols, rows, 4, omit_intergenic=True)) == ["AAAA", "CCCC", "TTTT", "TTTT", "ACAC", "NONE"]
    assert sp(D.filter_guides(cols, rows, 4, full_overlap=True)) == ["AAAA", "CCCC", "TTTT", "TTTT", "ACAC"]
    # tiles of 8 along each gene: G1 offsets 0,4,8 -> 0 and 8; G2 offsets 2,22 -> both; G3 -> 30
    assert sp(D.filter_guides(cols, rows, 4, tile_size=8)) == ["AAAA", "GGGG", "GGGG", "TTTT", "TTTT", "ACAC"]
    # with full overlap the walk starts at the first fully overlapping guide and GGGG (overlap 3) is gone already
    assert sp(D.filter_guides(cols, rows, 4, full_overlap=True, tile_size=8)) == ["AAAA", "TTTT", "TTTT", "ACAC"]
    assert sp(D.filter_guides(cols, rows, 4, keep_top=1)) == ["AAAA", "TTTT", "TTTT"]
    assert sp(D.filter_guides(cols, rows, 4, keep_top=2)) == ["AAAA", "CCCC", "TTTT", "TTTT", "ACAC"]


def _zstd_compress(raw: bytes):
    import ctypes
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(cap)
    m = z.ZSTD_compress(dst, cap, raw, len(raw), 3)
    return dst.raw[:m]


def test_damaged_compressed_stream_is_an_error(tmp_path):
    """A truncated or corrupt .gz / .zst reads file must fail (the reference's gzip / zstandard readers
    raise; main() then prints no counts, heuristicount.py:882-888) — never a silently shorter stream."""
    import random
    from barcoder_amd import ingest
    rng = random.Random(7)
    reads = ["".join(rng.choice("ACGT") for _ in range(60)) for _ in range(40000)]
    raw = ("\n".join(reads) + "\n").encode()
    good = gzip.compress(raw)
    (tmp_path / "ok.reads.gz").write_bytes(good)
    assert ingest.read_all(str(tmp_path / "ok.reads.gz")) == reads
    (tmp_path / "cut.reads.gz").write_bytes(good[:len(good) // 2])
    with pytest.raises(ValueError, match="gzip stream"):
        ingest.read_all(str(tmp_path / "cut.reads.gz"))
    bad = bytearray(good)
    for i in range(len(bad) // 2, len(bad) // 2 + 64):
        bad[i] ^= 0x5A
    (tmp_path / "bad.reads.gz").write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="gzip stream"):
        ingest.read_all(str(tmp_path / "bad.reads.gz"))
    z = _zstd_compress(raw)
    if z is not None:
        (tmp_path / "ok.reads.zst").write_bytes(z)
        assert ingest.read_all(str(tmp_path / "ok.reads.zst")) == reads
        (tmp_path / "cut.reads.zst").write_bytes(z[:len(z) // 2])
        with pytest.raises(ValueError, match="zstd stream"):
            ingest.read_all(str(tmp_path / "cut.reads.zst"))
        zb = bytearray(z)
        for i in range(len(zb) // 2, len(zb) // 2 + 64):
            zb[i] ^= 0x5A
        (tmp_path / "bad.reads.zst").write_bytes(bytes(zb))
        with pytest.raises(ValueError, match="zstd stream"):
            ingest.read_all(str(tmp_path / "bad.reads.zst"))


def test_reader_byte_ranges_tile_the_file(tmp_path):
    """bc_reader_open_range: ranges that tile an uncompressed file deliver every record exactly once, in
    order, for .reads and for FASTQ (quality lines that begin with '@' or '+' included); the tell() keys of
    consecutive chunks order reads like their global index."""
    import random
    from barcoder_amd import ingest
    rng = random.Random(3)
    reads = ["".join(rng.choice("ACGTN") for _ in range(rng.randrange(1, 90))) for _ in range(2500)]
    pr = tmp_path / "x.reads"
    pr.write_text("\n".join(reads) + "\n")
    pq = tmp_path / "x.fastq"
    with open(pq, "w") as f:
        for i, r in enumerate(reads):
            q = "".join(rng.choice("@+IF#") for _ in r)
            f.write(f"@r{i} x\n{r}\n+\n{q}\n")
    pt = tmp_path / "t.fastq"  # last record cut off after its sequence line
    pt.write_text("".join(f"@r{i}\n{r}\n+\n{'@' * len(r)}\n" for i, r in enumerate(reads[:50])) + "@last\nACGT")

    def ranged(path, cuts):
        got, keys = [], []
        for a, b in zip(cuts, cuts[1:]):
            with ingest.NativeReader(str(path), max_reads=97, byte_range=(a, b)) as r:
                while True:
                    key = r.tell()
                    text, off = r.next()
                    if off.size <= 1:
                        break
                    bts = text.tobytes()
                    o = off.astype(np.int64)
                    got += [bts[o[i]:o[i + 1]].decode() for i in range(o.size - 1)]
                    keys += [key + i for i in range(o.size - 1)]
        return got, keys

    for path, want in ((pr, reads), (pq, reads), (pt, reads[:50] + ["ACGT"])):
        size = os.path.getsize(path)
        for world in (1, 2, 3, 7, 64):
            per = -(-size // world)
            cuts = [min(size, r * per) for r in range(world)] + [size]
            got, keys = ranged(path, cuts)
            assert got == want, (path.name, world)
            assert all(x < y for x, y in zip(keys, keys[1:])), (path.name, world)
        for _ in range(20):  # arbitrary cut points, including inside records and at line starts
            cuts = [0] + sorted(rng.randrange(0, size + 1) for _ in range(4)) + [size]
            assert ranged(path, cuts)[0] == want
    with pytest.raises(ValueError, match="compressed"):
        (tmp_path / "x.reads.gz").write_bytes(gzip.compress(b"ACGT\n"))
        ingest.NativeReader(str(tmp_path / "x.reads.gz"), byte_range=(0, 5))


@pytest.mark.parametrize("fastq", [False, True])
def test_paired_read_ranges_hold_the_same_reads(tmp_path, fastq):
    """ingest.paired_read_ranges: for every rank of 1, 2, 3 and 5 the two byte ranges start at record boundaries,
    tile their files, and hold the same read indices although the mates' lines differ in length (and the last line
    of one file has no newline); files with different numbers of reads raise the reference's error."""
    import random
    from barcoder_amd import ingest as I
    rng = random.Random(5)
    n = 1237
    a = ["".join(rng.choice("ACGT") for _ in range(rng.randrange(20, 60))) for _ in range(n)]
    b = ["".join(rng.choice("ACGT") for _ in range(rng.randrange(5, 90))) for _ in range(n)]
    ext = ".fastq" if fastq else ".reads"

    def write(path, reads, trailing=True):
        rec = [f"@r{i}\n{r}\n+\n{'I' * len(r)}" if fastq else r for i, r in enumerate(reads)]
        with open(path, "w") as f:
            f.write("\n".join(rec) + ("\n" if trailing else ""))
    fa, fb = str(tmp_path / ("a" + ext)), str(tmp_path / ("b" + ext))
    write(fa, a)
    write(fb, b, trailing=False)
    per = 4 if fastq else 1
    for world in (1, 2, 3, 5):
        calls = {"i": 0}

        def gather(obj):  # what dist.gather_objects returns: every rank's (newlines of its slice, last byte)
            path = (fa, fb)[calls["i"] % 2]
            calls["i"] += 1
            size = os.path.getsize(path)
            cuts = [size * r // world for r in range(world + 1)]
            last = I.file_count_newlines(path, 0, 0)[2]
            return [(I.file_count_newlines(path, cuts[r], cuts[r + 1])[0], last) for r in range(world)]
        got = [I.paired_read_ranges(fa, fb, r, world, gather) for r in range(world)]
        for path, reads, which in ((fa, a, 1), (fb, b, 2)):
            data = open(path, "rb").read()
            assert got[0][which][0] == 0 and got[-1][which][1] == len(data)
            for r in range(world):
                r0, (lo, hi) = got[r][0], got[r][which]
                if r + 1 < world:
                    assert hi == got[r + 1][which][0]
                assert r0 == n * r // world
                lines = data[lo:hi].decode().split("\n")
                if lines and lines[-1] == "":
                    lines.pop()
                seqs = lines[1::4] if fastq else lines
                assert seqs == reads[r0:n * (r + 1) // world], (world, r, which)
                assert lo == 0 or data[lo - 1:lo] == b"\n"
    write(fb, b[:-1])
    calls = {"i": 0}
    world = 2
    with pytest.raises(ValueError, match="Length of reads1 and reads2 must be the same"):
        I.paired_read_ranges(fa, fb, 0, world, gather)
    assert I.file_seek_newline(fa, 0, 0) == 0 and I.file_seek_newline(fa, 0, 10**9) == os.path.getsize(fa)


def test_merge_undocumented_any_order():
    from barcoder_amd import heuristicount as H
    u = np.uint64
    a = (np.array([5, 9, 7], u), np.array([2, 1, 1], u), np.array([10, 3, 8], u))
    b = (np.array([9, 5, 1], u), np.array([4, 1, 6], u), np.array([1, 30, 2], u))
    for tables in ([a, b], [b, a], [a, (None, None, None), b]):
        w, c, f = H.merge_undocumented(tables)
        assert w.tolist() == [1, 5, 7, 9] and c.tolist() == [6, 3, 1, 5] and f.tolist() == [2, 10, 8, 1]
    w, c, f = H.merge_undocumented([])
    assert w.size == c.size == f.size == 0
    # barcodes of more than 32 nt: the empty table has text rows of that width (a rank that owned no chunk), it merges
    # with a filled one, and nothing treats it as packed k-mers
    w, c, f = H.merge_undocumented([], 40)
    assert w.size == 0 and w.dtype == np.dtype((np.void, 40)) and H.window_labels(w, 40) == []
    rows = np.frombuffer(b"A" * 40 + b"C" * 40, np.dtype((np.void, 40)))
    filled = (rows, np.array([2, 3], u), np.array([7, 5], u))
    w2, c2, f2 = H.merge_undocumented([(w, c, f), filled, filled], 40)
    assert H.window_labels(w2, 40) == ["A" * 40, "C" * 40] and c2.tolist() == [4, 6] and f2.tolist() == [7, 5]
    assert H.window_labels(np.zeros(0, u), 20) == []
    from barcoder_amd.encoding import unpack_kmers
    assert unpack_kmers(np.zeros(0, u), 40) == []
    assert H.shard_plan("a.reads", None, 1) == "stride" and H.shard_plan("a.reads", None, 4) == "bytes"
    assert H.shard_plan("a.reads.gz", None, 4) == "stride" and H.shard_plan("a.fastq", "b.fastq", 4) == "stride"


def test_launch_workers_starts_ranks_and_reports_failure(tmp_path):
    """dist.launch_workers: fresh child interpreters with the torchrun environment; rank 0 owns stdout; a
    failing rank ends the job with its status."""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "print(f'rank {r} of {w}')\n"
        "if len(sys.argv) > 1 and r == int(sys.argv[1]): sys.exit(7)\n")
    drv = ("import sys; sys.path.insert(0, %r); import importlib.util as u; "
           "s = u.spec_from_file_location('d', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
           "assert 'torch' not in sys.modules; "
           "sys.exit(m.launch_workers([%r] + sys.argv[1:], 3, timeout=60))") % (
        ROOT, os.path.join(ROOT, "barcoder_amd", "dist.py"), str(script))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, env=env)
    assert ok.returncode == 0 and ok.stdout.strip() == "rank 0 of 3"
    assert "rank 1 of 3" in ok.stderr and "rank 2 of 3" in ok.stderr
    bad = subprocess.run([sys.executable, "-c", drv, "2"], capture_output=True, text=True, env=env)
    assert bad.returncode == 7


def test_launch_workers_ends_a_rank_that_hangs(tmp_path):
    """A rank that hangs (it does not fail: RCCL initialisation, a collective a peer never joins) must not hold the job
    for ever: launch_workers' wait is bounded — by default too — and ends the ranks still alive with status 124; one
    that ignores SIGTERM is killed."""
    import subprocess
    import sys
    import time
    script = tmp_path / "h.py"
    script.write_text("\n".join([
        "import os, signal, sys, time",
        "r = int(os.environ['RANK'])",
        "if r == 1:",
        "    signal.signal(signal.SIGTERM, signal.SIG_IGN)",
        "    time.sleep(600)",
        "print('rank', r, 'done')", ""]))
    drv = ("import sys; sys.path.insert(0, %r); import importlib.util as u; "
           "s = u.spec_from_file_location('d', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
           "assert 0 < m.DEFAULT_LAUNCH_TIMEOUT < float('inf'); "
           "sys.exit(m.launch_workers([%r], 2, timeout=float(sys.argv[1]) if len(sys.argv) > 1 else -1))") % (
        ROOT, os.path.join(ROOT, "barcoder_amd", "dist.py"), str(script))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BC_LAUNCH_TIMEOUT")}
    t0 = time.monotonic()
    hung = subprocess.run([sys.executable, "-c", drv, "1.5"], capture_output=True, text=True, env=env, timeout=120)
    assert hung.returncode == 124 and "rank 0 done" in hung.stdout and "still running" in hung.stderr
    assert time.monotonic() - t0 < 60
    # the default comes from BC_LAUNCH_TIMEOUT (24 h without it): finite either way
    hung = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, env=dict(env, BC_LAUNCH_TIMEOUT="1"), timeout=120)
    assert hung.returncode == 124


def test_native_flanks_on_worker_threads(tmp_path):
    """bc_sample_flanks counts a large sample on worker threads and merges their counters: same flanks as the Python
    restatement of find_flanks (heuristicount.py:382-425) on a sample big enough for several threads — a majority
    flank, a minority variant of it, reads that end inside the right flank."""
    import random
    from barcoder_amd import heuristicount as H
    rng = random.Random(11)
    k = 18
    lib = list(dict.fromkeys("".join(rng.choice("ACGT") for _ in range(k)) for _ in range(8000)))
    reads = []
    for i in range(90000):
        L = "CCGGATTCAG" if rng.random() < 0.7 else "CCGGTTTCAG"
        R = "GTTTTAGAGC" if rng.random() < 0.8 else "GTTTAAGAGC"
        r = "".join(rng.choice("ACGT") for _ in range(4)) + L + rng.choice(lib) + R + "AC"
        if rng.random() < 0.1:
            r = r[:4 + 10 + k + rng.randrange(10)]
        reads.append(r)
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    rd = tmp_path / "r.reads"
    rd.write_text("".join(r + "\n" for r in reads))
    ordered = H.read_fasta_ordered(str(fa))
    smp = H.NativeSample(str(rd), None, ordered, False)
    want = H.sample_data(str(rd), None, set(ordered), False)
    assert smp.offset1 == want[1] == 14 and smp.n_valid1 == len(want[3]) > 24576
    assert tuple(smp.flanks(0, smp.offset1, k)) == tuple(H.find_flanks(want[3], want[1], k)) == ("CCGGATTCAG", "GTTTTAGAGC")
    smp.close()


def test_crispri_views_match_reference_class():
    """`crispri.CRISPRiLibrary` + `pam.PAMFinder` + `crispri.joined_frame` against the views the reference's own
    CRISPRiLibrary (CRISPRiLibrary.py:4-120) and PAMFinder (PAMProcessor.py:60-97) computed from the same hit x
    feature frame (tests/golden/crispri.json.gz, oracle/gen_golden.py).  The hits come from the fixture here; the
    GPU twin (tests/test_targets_gpu.py) gets them from `targets.find_hits`."""
    import gzip
    import json
    from barcoder_amd import targets as T
    from barcoder_amd.crispri import CRISPRiLibrary, joined_frame
    from barcoder_amd.library import read_fasta_records
    from barcoder_amd.pam import PAMFinder
    with gzip.open(os.path.join(GOLDEN, "crispri.json.gz"), "rt") as f:
        gold = json.load(f)
    names = ["CP023716.1", "CP023717.1", "CP023718.1", "CP023719.1"]
    seqs = dict(zip(names, read_fasta_records(os.path.join(GOLDEN, "plasmids.fasta"))))
    lens = {k: len(v) for k, v in seqs.items()}
    parts = T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"))
    hits = {}
    for fr in gold["frame"]:
        hits[(fr["Chromosome"], fr["Start"], fr["Strand"], fr["Barcode"])] = {
            "chr": fr["Chromosome"], "tar_start": fr["Start"], "len": fr["End"] - fr["Start"],
            "sp_dir": "F" if fr["Strand"] == "+" else "R", "spacer": fr["Barcode"], "mismatches": fr["Mismatches"]}
    rows = list(hits.values())[::-1] + [{"name": "x", "spacer": "ACGT", "len": 4}]  # any order; bare rows join nothing
    df = joined_frame(rows, parts, lens)
    rec = lambda d: json.loads(d.to_json(orient="records"))
    assert rec(df) == gold["frame"]
    for case in gold["cases"]:
        lib = CRISPRiLibrary(df.copy(), PAMFinder(seqs, case["pam"], case["direction"]))
        assert list(lib.targets_df["PAM"]) == case["PAM"]
        assert [bool(x) for x in lib.targets_df["Targeting"]] == case["Targeting"]
        for view in ("source_unique_targets", "mapped_targets", "unique_targets", "unambiguous_targets"):
            assert rec(getattr(lib, view)) == case[view], (case["pam"], view)
        assert [int(i) for i in lib.unambiguous_targets.index] == case["unambiguous_index"]


def test_reader_raw_stream(tmp_path):
    """bc_reader_read_raw: the decoded byte stream of a plain / .gz / .zst reads file, sequentially, in pieces of any
    size (what the device ingest uploads of a compressed file); a stream cut short is an error, never a short file."""
    import ctypes as Ct
    from barcoder_amd import _native as N
    raw = b"".join(b"ACGT" * 9 + bytes([65 + i % 4]) + b"\n" for i in range(50_000))
    files = {"a.reads": raw, "a.reads.gz": gzip.compress(raw, 1)}
    z = _zstd_compress(raw)
    if z is not None:
        files["a.reads.zst"] = z
    L = N.load()
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        for piece in (1 << 20, 70_001):
            h = N.vp()
            N.check(L.bc_reader_open(str(p).encode(), Ct.byref(h)))
            buf, got, out = np.empty(piece, np.uint8), Ct.c_uint64(1), []
            while True:
                N.check(L.bc_reader_read_raw(h, N.ptr(buf), piece, Ct.byref(got)))
                if not got.value:
                    break
                out.append(buf[:got.value].tobytes())
            L.bc_reader_close(h)
            assert b"".join(out) == raw, (name, piece)
        if name != "a.reads":
            cut = tmp_path / ("cut_" + name)
            cut.write_bytes(data[:len(data) // 2])
            h = N.vp()
            N.check(L.bc_reader_open(str(cut).encode(), Ct.byref(h)))
            buf, got = np.empty(1 << 20, np.uint8), Ct.c_uint64(1)
            with pytest.raises(ValueError, match="truncated|damaged"):
                while got.value:
                    N.check(L.bc_reader_read_raw(h, N.ptr(buf), buf.size, Ct.byref(got)))
            L.bc_reader_close(h)


def test_native_sampling_fast_path_equals_loop_form(tmp_path, monkeypatch):
    """The single-end sampler counts a chunk's library k-mers from a table of first meetings filled by worker threads
    (bc_sample.cpp) instead of walking the reads in order; a chunk that holds a read with more library k-mers than the
    look-ahead keeps takes the loop form.  Both must give what the golden-pinned Python statement of
    heuristicount.py:156-379 gives, on input built to hit the corners: repeated reads, reverse-complement barcodes,
    entries that are their own reverse complement or each other's, reads with two barcodes, a read with eight (third
    chunk only), short reads, an N inside a window."""
    import random
    from barcoder_amd import heuristicount as H
    rng = random.Random(5)
    k, G = 16, 12000
    rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    lib = ["ACGTACGTACGTACGT", "AAAACCCCGGGGTTTT"]            # their own reverse complements
    lib += ["ACGGTCATGCATGGCA", rc("ACGGTCATGCATGGCA")]         # each other's
    while len(lib) < G:
        lib.append("".join(rng.choice("ACGT") for _ in range(k)))
    lib = list(dict.fromkeys(lib))
    L, R = "CCGGATTCAG", "GTTTTAGAGC"
    reads = []
    n = 6 * len(lib)
    for i in range(n):
        u = rng.random()
        pre = "".join(rng.choice("ACGT") for _ in range(5))
        if u < 0.10 and reads:
            r = reads[rng.randrange(len(reads))]                  # an earlier read again
        elif u < 0.20:
            r = pre + L + rc(rng.choice(lib)) + R
        elif u < 0.22:
            r = pre + L + rng.choice(lib) + rng.choice(lib) + R   # two barcodes (and whatever straddles them)
        elif u < 0.24:
            r = pre + L + rng.choice(lib)[:10]                    # short
        elif u < 0.26:
            b = rng.choice(lib)
            r = pre + L + b[:7] + "N" + b[8:] + R
        elif u < 0.30:
            r = pre + L + lib[rng.randrange(4)] + R               # the palindromic corner, often
        else:
            r = pre + L + rng.choice(lib) + R
        reads.append(r)
    reads[2 * len(lib) + 17] = "".join(rng.choice(lib) for _ in range(8))   # eight barcodes: the third chunk's loop form
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    rd = tmp_path / "r.reads"
    rd.write_text("".join(r + "\n" for r in reads))
    ordered = H.read_fasta_ordered(str(fa))
    want = H.sample_data(str(rd), None, set(ordered), False)
    # (the sampled chunks' text is kept and the sets hold views into it — up to BC_SAMPLE_KEEP_MB, 1 GiB by default; past
    # that, here from the second chunk on, the sets copy what they keep: both forms, and several thread counts)
    for keep_mb, threads in ((None, None), ("1", "3"), ("0", "7")):
        for name, val in (("BC_SAMPLE_KEEP_MB", keep_mb), ("BC_SAMPLE_THREADS", threads)):
            if val is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, val)
        smp = H.NativeSample(str(rd), None, ordered, False)
        got = (smp.n_novel, smp.offset1, smp.offset2, smp.n_valid1, smp.n_valid2, smp.n_observed, smp.need_swap, smp.num_chunks)
        assert got == (want[0], want[1], want[2], len(want[3]), len(want[4]), len(want[5]), want[6], want[7]), (keep_mb, threads)
        assert tuple(smp.flanks(0, smp.offset1, k)) == tuple(H.find_flanks(want[3], want[1], k))
        smp.close()
    assert want[7] >= 4 and want[1] == 15


def test_native_sampling_mate_pairs_equal_loop_form(tmp_path):
    """Mate pairs with the pre-filter: the worker threads find the windows of both mates that are library entries and the
    reference's loop runs over those positions alone; the distinct-read count is taken afterwards from the pairs the loop
    did not skip.  Held against the golden-pinned Python statement of heuristicount.py:156-379 on pairs built to hit the
    coupling between the mates: a first-mate window already met skips its mate's, a pair is skipped when EITHER mate
    equals an earlier valid read, mates shorter than the first mate, barcodes past the first mate's last window, a pair
    with more library k-mers than the look-ahead keeps."""
    import random
    from barcoder_amd import heuristicount as H
    rng = random.Random(9)
    k, G = 16, 9000
    rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    lib = ["ACGTACGTACGTACGT"]
    while len(lib) < G:
        lib.append("".join(rng.choice("ACGT") for _ in range(k)))
    lib = list(dict.fromkeys(lib))
    L, R = "CCGGATTCAG", "GTTTTAGAGC"
    r1s, r2s = [], []
    n = 7 * len(lib)
    for i in range(n):
        u = rng.random()
        b = rng.choice(lib)
        pre = "".join(rng.choice("ACGT") for _ in range(5))
        r1 = pre + L + b + R
        r2 = pre[:3] + rc(L + b + R)
        if u < 0.08 and r1s:
            j = rng.randrange(len(r1s))
            r1 = r1s[j]                                           # first mate seen before, another mate
        elif u < 0.16 and r2s:
            r2 = r2s[rng.randrange(len(r2s))]                     # mate seen before
        elif u < 0.20:
            r2 = r2[:rng.randrange(0, 25)]                        # short (or empty) mate
        elif u < 0.24:
            r1 = r1[:20]                                          # short first mate: few windows, the mate's barcode lies past them
        elif u < 0.28:
            r2 = "AC" + L + b + R                                 # both mates carry the SAME k-mer at the same position
            r1 = "AC" + L + b + R
        elif u < 0.32:
            r2 = pre + L + rng.choice(lib) + R                    # forward barcode in the mate
        elif u < 0.34:
            r1 = pre + L + "N" + b[1:] + R
        r1s.append(r1); r2s.append(r2)
    j = 2 * len(lib) + 5
    r1s[j] = "".join(rng.choice(lib) for _ in range(6))
    r2s[j] = "".join(rng.choice(lib) for _ in range(6))          # twelve library k-mers in one pair
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    f1, f2 = tmp_path / "a.reads", tmp_path / "b.reads"
    f1.write_text("".join(r + "\n" for r in r1s))
    f2.write_text("".join(r + "\n" for r in r2s))
    ordered = H.read_fasta_ordered(str(fa))
    want = H.sample_data(str(f1), str(f2), set(ordered), True)
    smp = H.NativeSample(str(f1), str(f2), ordered, True)
    got = (smp.n_novel, smp.offset1, smp.offset2, smp.n_valid1, smp.n_valid2, smp.n_observed, smp.need_swap, smp.num_chunks)
    assert got == (want[0], want[1], want[2], len(want[3]), len(want[4]), len(want[5]), want[6], want[7])
    assert want[7] >= 3 and len(want[3]) > 1000 and len(want[4]) > 1000
    assert tuple(smp.flanks(0, smp.offset1, k)) == tuple(H.find_flanks(want[3], want[1], k))
    assert tuple(smp.flanks(1, smp.offset2, k)) == tuple(H.find_flanks(want[4], want[2], k))
    smp.close()


def test_four_pair_tables_cover_two_substitutions_and_count_every_guide_once():
    """The combinatorics behind the d <= 2 cascade's second stage (bc_match_pairs.hip, build_pair_index / d_pairs2).  A
    k-mer is cut into five pieces; a guide within two substitutions of a read keeps at least three pieces clean.
    (1) The four pair tables (0,1) (2,3) (2,4) (3,4) hold every such guide in one of the read's buckets (a pair of clean
    pieces), and no three pair tables do — four is the minimum.  (2) The count-once rule the kernel evaluates per
    candidate — pieces a and b clean, and every `other` mask non-zero (= at least one piece of the mask's set dirty) —
    with the masks build_pair_index gives the four tables counts the guide in EXACTLY one of them, for every set of clean
    pieces of size >= 3.  (3) Budget three (two clean pieces at least) needs all ten pairs."""
    from itertools import combinations
    pieces = range(5)
    pairs = list(combinations(pieces, 2))
    triples_plus = [set(c) for r in (3, 4, 5) for c in combinations(pieces, r)]
    four = [(0, 1), (2, 3), (2, 4), (3, 4)]
    covers = lambda tabs, clean_sets: all(any(a in c and b in c for a, b in tabs) for c in clean_sets)
    assert covers(four, triples_plus)
    assert not any(covers(t, triples_plus) for t in combinations(pairs, 3))
    # `other` sets as build_pair_index fills them for the four tables (a mask over several pieces = their union)
    other = {(0, 1): [], (2, 3): [{0, 1}], (2, 4): [{0, 1}, {3}], (3, 4): [{0, 1}, {2}]}
    for clean in triples_plus:
        counted = [t for t in four if t[0] in clean and t[1] in clean and all(m - clean for m in other[t])]
        assert len(counted) == 1, (clean, counted)
    # the ten-table rule (lexicographically first clean pair: every piece below b other than a is dirty) counts once too,
    # and nine tables do not cover two clean pieces
    doubles_plus = [set(c) for r in (2, 3, 4, 5) for c in combinations(pieces, r)]
    for clean in doubles_plus:
        counted = [(a, b) for a, b in pairs if a in clean and b in clean and all(c not in clean for c in range(b) if c != a)]
        assert len(counted) == 1, (clean, counted)
    assert not any(covers(t, doubles_plus) for t in combinations(pairs, 9))
