"""GPU: d = 1 best-hit search pinned by vectors the REFERENCE holds.  The reference has no read->library
mismatch search, but mismatch.py:81-111 (generate_mismatches) enumerates, for a spacer, its single-base variants
with the change string `{orig}{pos}{new}` and a y_pred score; tests/golden/mismatch.json keeps that output for six
spacers (made by oracle/gen_golden.py from the unmodified reference).  Pushed through the HIP matcher at
max_mismatch = 1, every variant must come back to its spacer at distance 1, and the device's mismatch mask must
reproduce the reference's change string and score."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _tables():
    m = json.load(open(os.path.join(GOLDEN, "mismatch.json")))
    out = []
    for t in m["tables"]:
        rows = [ln.split("\t") for ln in t["tsv"].splitlines() if ln]
        out.append((t["spacer"], rows))
    return m["params"], out


def test_reference_variants_return_to_their_spacer_at_d1(gpu_device):
    from barcoder_amd import DeviceLibrary, GpuMatcher, encoding as E, mismatch as M, synth as S, targets as T
    params, tables = _tables()
    spacers = [sp for sp, _ in tables]
    k = len(spacers[0])
    # library: the reference's spacers (upper case) in front of seeded decoys
    decoys = [s for s in E.unpack_kmers(S.library(5000, k), k) if s not in spacers]
    lib_s = [sp.upper() for sp in spacers] + decoys
    lib, fl = E.pack_kmers(lib_s, k)
    assert not fl.any()
    variants, origin, change, score = [], [], [], []
    for gi, (sp, rows) in enumerate(tables):
        assert len(rows) >= 10
        for o, v, ch, y in rows:
            assert o == sp and len(v) == k
            variants.append(v.upper()); origin.append(gi); change.append(ch); score.append(y)
    reads, fl = E.pack_kmers(variants, k)
    assert not fl.any()
    dl = DeviceLibrary(lib, k, gpu_device)
    for algo in ("seed", "brute", "seed_csr"):
        res = GpuMatcher(dl).match(reads, None, max_mismatch=1, algo=algo)
        np.testing.assert_array_equal(res.best_idx, np.array(origin, np.int32))
        assert (res.best_dist == 1).all() and (res.n_best == 1).all()
        np.testing.assert_array_equal(res.counts[:len(spacers)], np.bincount(origin, minlength=len(spacers)))
    assert dl.last_kernel == "k_match_seed"
    res = GpuMatcher(dl).match(reads, None, max_mismatch=1, algo="seed")
    assert dl.last_kernel in ("k_match_wq", "k_match_slab")  # 360 reads: the small-input kernel may take it
    # at d = 0 none of them is in the library
    assert (GpuMatcher(dl).match(reads, None, max_mismatch=0, algo="hash").best_idx == -1).all()
    # the mismatch mask (what a hit record carries) reproduces the reference's change column ...
    g = lib[res.best_idx]
    x = g ^ reads
    mask = ((x & np.uint64(0xFFFFFFFF)) | (x >> np.uint64(32))).astype(np.uint64)
    for i, (v, ch) in enumerate(zip(variants, change)):
        sp = lib_s[res.best_idx[i]]
        m = int(mask[i])
        assert bin(m).count("1") == 1
        pos = m.bit_length() - 1
        assert f"{sp[pos]}{pos + 1}{v[pos]}" == ch.upper()
        # targets.get_diff's form of the same record (targets.py:184-190): {target_nt}{i+1}{spacer_nt}
        assert T.diff_from_mask(sp, v, m) == f"{v[pos]}{pos + 1}{sp[pos]}" == T.get_diff(sp, v)
    # ... and the device score of (guide, read) pairs equals the reference's y_pred column
    y = M.y_pred_device([lib_s[j] for j in res.best_idx], variants, params, device=gpu_device)
    assert [f"{a:.4f}" for a in y] == score


def test_d1_counts_bound_d0_counts(gpu_device, tmp_path):
    """heuristicount scenarios of the reference with --mismatches 1: every barcode's documented count is at least
    the reference's exact-match count (a read within one substitution can only add), totals do not shrink."""
    import contextlib
    import io
    import types
    from barcoder_amd import heuristicount as H
    with gzip.open(os.path.join(GOLDEN, "heuristicount_main.json.gz"), "rt") as f:
        cases = json.load(f)
    grew = 0
    for s in cases:
        if s["name"] == "noflank":
            continue
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
        args = types.SimpleNamespace(fasta_file=str(fa), file1=str(f1), file2=str(f2) if f2 else None, mismatches=1)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            H.main(args, device=gpu_device)
        got = dict((ln.split("\t")[0], int(ln.split("\t")[1])) for ln in buf.getvalue().splitlines())
        ref = dict((ln.split("\t")[0], int(ln.split("\t")[1])) for ln in s["stdout"].splitlines())
        assert set(ref) <= set(got), s["name"]
        assert all(got[b] >= c for b, c in ref.items()), s["name"]
        grew += sum(got.values()) - sum(ref.values())
    assert grew > 0  # the scenarios contain 1-mismatch decoys: some must have been picked up


@pytest.mark.parametrize("budget", [2, 3])
def test_reference_multi_substitution_pairs_pin_d2_d3(gpu_device, oracle, budget):
    """d = 2 and d = 3 pinned by vectors the reference holds: the 42 (original, variant, y_pred) pairs that
    mismatch.py:15-35 (calculate_y_pred, `recalculate` mode) scored — 22 with one substitution, 9 with two, 10 with
    three, one identical (tests/golden/mismatch.json "pairs", from the unmodified reference).  With the variant as
    the read and the originals in front of 5 000 decoys as the library, the HIP matcher at max_mismatch = budget
    (wave kernel for 2, cascade incl. the pair index for 3) must return every variant within the budget to its
    original at distance = its number of substitutions with n_best = 1, leave the others unmatched, and
    bc_score_mismatch_dev(best guide, read) must equal the reference's y_pred bit for bit — which pins the mismatch
    mask of a d >= 2 hit record."""
    from barcoder_amd import DeviceLibrary, GpuMatcher, encoding as E, mismatch as M, synth as S
    m = json.load(open(os.path.join(GOLDEN, "mismatch.json")))
    pairs = [(o.upper(), v.upper(), y) for o, v, y in m["pairs"]]
    k = len(pairs[0][0])
    originals = list(dict.fromkeys(o for o, _, _ in pairs))
    decoys = [s for s in E.unpack_kmers(S.library(5000, k), k) if s not in originals]
    lib_s = originals + decoys
    lib, fl = E.pack_kmers(lib_s, k)
    assert not fl.any()
    variants = [v for _, v, _ in pairs]
    nsub = np.array([sum(a != b for a, b in zip(o, v)) for o, v, _ in pairs])
    assert sorted(set(nsub.tolist())) == [0, 1, 2, 3]
    want_idx = np.array([originals.index(o) if d <= budget else -1 for (o, _, _), d in zip(pairs, nsub)], np.int32)
    # no decoy comes within the budget of a variant (checked, not assumed): the oracle's brute force agrees
    ref = oracle.best_hit(lib_s, variants, budget)
    np.testing.assert_array_equal(ref["best_idx"], want_idx)
    reps = 64  # enough reads for the vectorised kernels (the same 42 variants over and over)
    reads, fl = E.pack_kmers(variants * reps, k)
    assert not fl.any()
    dl = DeviceLibrary(lib, k, gpu_device)
    for algo in ("seed", "brute", "seed_csr"):
        res = GpuMatcher(dl).match(reads, None, max_mismatch=budget, algo=algo)
        if algo == "seed":
            assert dl.last_kernel == ("k_match_wq" if budget == 2 else "cascade3:d2+k_match_pairs")
        np.testing.assert_array_equal(res.best_idx, np.tile(want_idx, reps), err_msg=algo)
        hit = np.tile(want_idx >= 0, reps)
        np.testing.assert_array_equal(res.best_dist[hit], np.tile(nsub, reps)[hit])
        assert (res.n_best[hit] == 1).all() and (res.n_best[~hit] == 0).all() and (res.best_dist[~hit] == 255).all()
        np.testing.assert_array_equal(res.counts[:len(originals)],
                                      reps * np.bincount(want_idx[want_idx >= 0], minlength=len(originals)))
    # the score of (best guide, read) = the reference's y_pred, exactly (float64)
    sel = [i for i in range(len(pairs)) if want_idx[i] >= 0]
    y = M.y_pred_device([lib_s[res.best_idx[i]] for i in sel], [variants[i] for i in sel], m["params"], device=gpu_device)
    for i, got in zip(sel, y):
        assert got == pairs[i][2] or (got is None and nsub[i] == 0), (pairs[i], got)
