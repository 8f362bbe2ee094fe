"""GPU parity for the PAM scan kernels and the reference-flavoured site lists."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plasmids():
    from barcoder_amd.library import read_fasta_records
    return read_fasta_records(os.path.join(GOLDEN, "plasmids.fasta"))


@pytest.mark.parametrize("pam", ["NGG", "NGNC", "TTTN", "G", "GGGGGGG", "NNGRRT"])
def test_scan_bitmaps_vs_oracle(gpu_device, oracle, plasmids, pam):
    from barcoder_amd import pam as P
    iupac = "R" in pam
    for s in (plasmids[0], plasmids[3][:1000], "ACGTNNGGCCGG", "GG", "A" * 31 + "GG", "C" * 33 + "NGG" + "AGG"):
        fwd, rev = P.pam_scan(s, pam, gpu_device, iupac=iupac)
        if iupac:  # oracle knows only N; expand R = A|G by union of concrete patterns
            of = np.zeros(len(s), bool)
            orv = np.zeros(len(s), bool)
            for a in "AG":
                for b in "AG":
                    f2, r2 = oracle.pam_bitmaps(s, f"NNG{a}{b}T")
                    of |= f2.astype(bool)
                    orv |= r2.astype(bool)
        else:
            f2, r2 = oracle.pam_bitmaps(s, pam)
            of, orv = f2.astype(bool), r2.astype(bool)
        np.testing.assert_array_equal(fwd, of)
        np.testing.assert_array_equal(rev, orv)


def test_scan_wrap_equals_doubled_sequence(gpu_device, oracle, plasmids):
    """Circular handling: scanning len+wrap positions of the wrapped layout equals
    scanning seq+seq[:wrap] (the reference's overhang, targets.py:44-46)."""
    from barcoder_amd import pam as P
    s = plasmids[1]
    wrap = 40
    fwd, rev = P.pam_scan(P.pack_genome(s, wrap), "NGG", gpu_device)
    f2, r2 = oracle.pam_bitmaps(s + s[:wrap], "NGG")
    np.testing.assert_array_equal(fwd, f2.astype(bool))
    np.testing.assert_array_equal(rev, r2.astype(bool))


def test_guidefinder_matches_reference(gpu_device, plasmids):
    """GuideFinder.find_guides_from_pam: same list (order, duplicates, clipped
    guides) as PAMProcessor.py:34-57 produced on the 4 shipped plasmids."""
    from barcoder_amd.pam import GuideFinder
    with gzip.open(os.path.join(GOLDEN, "pam_guidefinder.json.gz"), "rt") as f:
        gold = json.load(f)
    names = ["CP023716.1", "CP023717.1", "CP023718.1", "CP023719.1"]
    records = dict(zip(names, plasmids))
    for key, want in gold.items():
        pam, L, direction = key.split("_")
        got = GuideFinder(records, pam, direction, int(L), device=gpu_device).find_guides_from_pam()
        assert got == want, key
    assert len(gold["NGG_20_downstream"]) == 11046


@pytest.mark.parametrize("pam,k", [("NGG", 20), ("NGNC", 32)])
@pytest.mark.parametrize("direction", ["downstream", "upstream"])
def test_sliding_scan_matches_oracle(gpu_device, oracle, plasmids, pam, k, direction):
    """design_guides.py:22-49 flavour on the topological (seq + overhang) strings."""
    from barcoder_amd.pam import sliding_spacers
    total = set()
    for s in plasmids[:2]:
        topo = s + s[:100_000]
        got = sliding_spacers(topo, k, pam, direction, gpu_device)
        assert got == oracle.sliding_sites_py(topo, k, pam, direction)
        total |= got
    assert len(total) > 2000
    # short / degenerate inputs, and non-ACGT characters inside spacers
    for s in ("", "ACGT", "ACGTNNNNACGGTTGGCCAAGGTTNACGGATCGGATTCGGAGGCTTAGG" * 3, "acgtAGGTCCGGTTAACCGGTTAACCGGTTAGG"):
        assert sliding_spacers(s, 6, "NGG", direction, gpu_device) == oracle.sliding_sites_py(s, 6, "NGG", direction)


def test_shipped_library_is_found_by_the_scan(gpu_device, plasmids):
    """All 706 plasmid spacers of CN-32-zmo.tsv are sliding-scan NGNC/32 sites."""
    from barcoder_amd.pam import sliding_spacers
    spacers = set()
    with open(os.path.join(GOLDEN, "cn32_plasmid_rows.tsv")) as f:
        cols = f.readline().rstrip("\n").split("\t")
        for line in f:
            spacers.add(dict(zip(cols, line.rstrip("\n").split("\t")))["spacer"])
    found = set()
    for s in plasmids:
        found |= sliding_spacers(s + s[:100_000], 32, "NGNC", "downstream", gpu_device)
    assert len(spacers) == 706 and spacers <= found


def test_find_sequences_from_fasta(gpu_device, oracle, tmp_path, plasmids):
    from barcoder_amd.pam import find_sequences_with_barcode_and_pam
    fa = tmp_path / "topo.fasta"
    topo = [s + s[:100_000] for s in plasmids[2:]]
    fa.write_text("".join(f">r{i}\n{t}\n" for i, t in enumerate(topo)))
    got = find_sequences_with_barcode_and_pam(str(fa), 20, "NGG", "downstream", gpu_device)
    want = set()
    for t in topo:
        want |= oracle.sliding_sites_py(t, 20, "NGG", "downstream")
    assert got == want


def test_windows_and_expand_full_size(gpu_device):
    """BASELINE configs[3] size: 4,641,652-bp synthetic circular genome, NGG/20.
    Size-independent properties: site count equals bitmap popcount, positions
    ascending, every reported spacer is followed by xGG / preceded by CCx."""
    from barcoder_amd import pam as P, synth as S, encoding as E
    seq = S.genome(4_641_652, seed=4)
    g = P.pack_genome(seq, wrap=20 + 3 - 1)
    fwd, rev = P.pam_scan(g, "NGG", gpu_device)
    st = P.find_sites(g, "NGG", 20, "downstream", gpu_device)
    n_f, n_r = int((st.strand == 0).sum()), int((st.strand == 1).sum())
    assert 500_000 < n_f + n_r < 700_000
    pf = st.pam_pos[st.strand == 0]
    assert (np.diff(pf) > 0).all() and fwd[pf].all()
    # forward sites with the spacer inside [0, total): all PAM positions >= 20
    assert n_f == int(fwd[20:].sum())
    topo = seq + seq[:22]
    idx = np.random.default_rng(0).choice(st.start.size, 300, replace=False)
    sp = E.unpack_kmers(st.spacer[idx], 20)
    for j, s in zip(idx, sp):
        a = int(st.start[j])
        if st.strand[j] == 0:
            assert topo[a:a + 20] == s and topo[a + 21:a + 23] == "GG"
        else:
            assert topo[a:a + 20] == s[::-1].translate(str.maketrans("ACGT", "TGCA")) and topo[a - 3:a - 1] == "CC"


@pytest.mark.parametrize("pam,k", [("NGG", 20), ("NGNC", 32), ("TTTN", 23), ("G", 1), ("NNGRRT", 21), ("GGNNNNNNNCC", 9)])
@pytest.mark.parametrize("direction", ["downstream", "upstream"])
def test_fused_site_pipeline_equals_staged(gpu_device, plasmids, pam, k, direction):
    """find_sites (two launches: bc_pam_sites_count_dev / _emit_dev) against the separate stages (bitmaps ->
    position lists -> windows, dropped on the host) on lengths around the 1024-word blocks, with non-ACGT
    characters, wrapped genomes and inputs shorter than a window."""
    from barcoder_amd import pam as P, synth as S
    iupac = "R" in pam
    rng = np.random.default_rng(7)
    long_seq = S.genome(70_000, seed=11)
    noisy = bytearray(long_seq.encode())
    for i in rng.choice(len(noisy), 600, replace=False):
        noisy[i] = ord("N")
    cases = [(plasmids[0], 0), (plasmids[1], k + len(pam) - 1), (noisy.decode(), 0), (long_seq[:32768], 0),
             (long_seq[:32769], 5), (long_seq[:32767 + 1024 * 32], 0), (long_seq[:65], 0), ("ACGTNNGGCCGGAAGG", 0),
             ("GG", 0), ("", 0), ("A" * 31 + "GG" + "C" * 40, 3)]
    for seq, wrap in cases:
        if wrap and not seq:
            continue
        g = P.pack_genome(seq, wrap)
        a = P.find_sites(g, pam, k, direction, gpu_device, iupac=iupac)
        b = P.find_sites_staged(g, pam, k, direction, gpu_device, iupac=iupac)
        for f in ("start", "strand", "spacer", "pam_pos"):
            np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f"{pam} {k} {direction} len {len(seq)} wrap {wrap}: {f}")
