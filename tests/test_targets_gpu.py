"""GPU: guide -> genome search (the Bowtie bypass) against the reference's own
shipped answer: the rows of Example_Libraries/CN-32-zmo.tsv that lie on the 4
shipped plasmids (tests/golden/cn32_plasmid_rows.tsv), and against the oracle
for mismatching searches."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
NAMES = ["CP023716.1", "CP023717.1", "CP023718.1", "CP023719.1"]


@pytest.fixture(scope="module")
def plasmids():
    from barcoder_amd.library import read_fasta_records
    from barcoder_amd.targets import Chromosome
    seqs = read_fasta_records(os.path.join(GOLDEN, "plasmids.fasta"))
    return [Chromosome(n, s, True) for n, s in zip(NAMES, seqs)]


def _golden_rows():
    with open(os.path.join(GOLDEN, "cn32_plasmid_rows.tsv")) as f:
        cols = f.readline().rstrip("\n").split("\t")
        return [dict(zip(cols, line.rstrip("\n").split("\t"))) for line in f]


def test_known_answer_table_reproduced(gpu_device, plasmids):
    """Library = the 706 plasmid spacers of the shipped TSV, genome = the 4 plasmids
    (circular), PAM NGNC downstream, 0 mismatches: the site list (spacer, chr, pam,
    target, coordinates, strand) equals the reference's rows."""
    from barcoder_amd.targets import find_hits, ranges_table
    gold = _golden_rows()
    want = {(r["spacer"], r["chr"], r["pam"], r["target"], int(r["tar_start"]), int(r["tar_end"]), r["sp_dir"],
             int(r["mismatches"])) for r in gold}
    spacers = list(dict.fromkeys(r["spacer"] for r in gold))
    assert len(spacers) == 706
    rows = find_hits(spacers, plasmids, pam="NGNC", num_mismatches=0, pam_direction="downstream", device=gpu_device)
    got = {(r["spacer"], r["chr"], r["pam"], r["target"], r["tar_start"], r["tar_end"], r["sp_dir"], r["mismatches"])
           for r in rows}
    assert got == want
    assert all(r["type"] == "perfect" and r["diff"] is None and r["len"] == 32 for r in rows)
    # sorted like targets.py:629-630, and the PySamParser-shaped table follows
    keys = [(r["chr"], min(r["tar_start"], r["tar_end"]), r["spacer"]) for r in rows]
    assert keys == sorted(keys)
    t = ranges_table(rows)
    assert set(t[0]) == {"Chromosome", "Start", "End", "Mapped", "Strand", "Barcode", "Mismatches"}
    assert any(r["tar_start"] < 0 for r in rows) == any(int(r["tar_start"]) < 0 for r in gold)


@pytest.mark.parametrize("direction", ["downstream", "upstream"])
@pytest.mark.parametrize("circular", [True, False])
@pytest.mark.parametrize("pam", ["NGG", "NG", ""])
def test_mismatch_search_vs_oracle(gpu_device, oracle, plasmids, direction, circular, pam):
    """<= 2 mismatches, NGG, 20-mers with planted substitutions: compare with a
    string-level restatement (oracle all_hits + oracle PAM extraction)."""
    from barcoder_amd.targets import Chromosome, find_hits
    rng = np.random.default_rng(5)
    seq = plasmids[0].seq[:6000]
    ch = Chromosome("c", seq, circular)
    k, d = 20, 2
    pl = len(pam)
    guides = []
    for _ in range(60):
        p = int(rng.integers(0, len(seq) - k))
        s = list(seq[p:p + k])
        for q in rng.choice(k, int(rng.integers(0, 3)), replace=False):
            s[q] = rng.choice([c for c in "ACGT" if c != s[q]])
        s = "".join(s)
        guides.append(s if rng.random() < 0.5 else oracle.rev_comp(s))
    guides += [seq[-7:] + seq[:13], oracle.rev_comp(seq[-3:] + seq[:17])]  # origin-spanning sites
    guides = list(dict.fromkeys(guides))
    rows = find_hits(guides, [ch], pam=pam, num_mismatches=d, pam_direction=direction, device=gpu_device)
    got = {(r["spacer"], r["tar_start"], r["tar_end"], r["sp_dir"], r["mismatches"], r["pam"], r["target"], r["diff"])
           for r in rows}
    # oracle: every window of the topological string, both strands
    n = len(seq)
    topo = seq + (seq[:k + pl - 1] if circular and k + pl > 1 else "")
    npos = n if circular else n - k + 1
    wins_f = [topo[p:p + k] for p in range(npos)]
    want = set()
    for strand, wins in (("F", wins_f), ("R", [oracle.rev_comp(w) for w in wins_f])):
        for q, g, dist in oracle.all_hits(guides, wins, d).tolist():
            a, b = q, q + k
            fwd_side_right = (strand == "F") == (direction == "downstream")
            ext = None
            if pl:
                if fwd_side_right:
                    if b + pl > len(topo):
                        continue
                    ext = topo[b:b + pl]
                else:
                    a2 = a - pl
                    if a2 < 0:
                        if not circular:
                            continue
                        a2 += n
                    ext = topo[a2:a2 + pl]
                if strand == "R":
                    ext = oracle.rev_comp(ext)
                if not oracle.pam_matches(pam, ext):
                    continue
            ts, te = oracle.fold_coords(a, b, n)
            target = wins[q]
            want.add((guides[g], ts, te, strand, dist, ext, target, oracle.get_diff(guides[g], target)))
    assert got == want
    assert len(got) >= {0: 40, 2: 5, 3: 0}[pl]


def test_mismatch_score_on_device_is_bit_exact(gpu_device):
    """bc_score_mismatch_dev vs the reference's own y_pred values (float64, no tolerance)
    and vs the pinned host mirror on 20k random pairs."""
    import json
    import random
    from barcoder_amd import mismatch as M
    m = json.load(open(os.path.join(GOLDEN, "mismatch.json")))
    params = m["params"]
    pairs = [(o, v, y) for o, v, y in m["pairs"]]
    got = M.y_pred_device([p[0] for p in pairs], [p[1] for p in pairs], params, gpu_device)
    for (o, v, want), g in zip(pairs, got):
        assert g == want, (o, v, g, want)  # None == None for the identical pair, exact float equality otherwise
    rng = random.Random(11)
    orig, var = [], []
    for _ in range(20000):
        o = "".join(rng.choice("ACGT") for _ in range(20))
        v = list(o)
        for p in rng.sample(range(20), rng.randrange(0, 6)):
            v[p] = rng.choice([c for c in "ACGT" if c != v[p]])
        orig.append(o)
        var.append("".join(v))
    got = M.y_pred_device(orig, var, params, gpu_device)
    want = [M.calculate_y_pred(o, v, params["GC_content"], params) for o, v in zip(orig, var)]
    assert got == want
    assert all(f"{a:.4f}" == f"{b:.4f}" for a, b in zip(got, want) if a is not None)


def test_annotated_table_matches_shipped_tsv(gpu_device, plasmids):
    """find_hits -> annotate -> final_table against every column of the reference's shipped rows on
    the plasmids (locus_tag, gene, offset, overlap, tar_dir included; `note` where all of a spacer's
    sites lie on the plasmids, since the chromosome's GenBank file is not shipped)."""
    from collections import Counter
    from barcoder_amd import targets as T
    gold = _golden_rows()
    genes = T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"), {c.name: len(c.seq) for c in plasmids})
    assert len(genes) == 160
    spacers = list(dict.fromkeys(r["spacer"] for r in gold))
    rows = T.find_hits(spacers, plasmids, pam="NGNC", num_mismatches=0, device=gpu_device)
    cols, table = T.final_table(T.annotate(rows, genes), {c.name: len(c.seq) for c in plasmids})
    assert cols == ["spacer", "locus_tag", "gene", "chr", "pam", "target", "tar_start", "tar_end", "offset", "overlap",
                    "sp_dir", "tar_dir", "note"]  # `mismatches` drops out: all zero in a plasmid-only run (:676-677)
    common = [c for c in cols if c != "note"]
    # the shipped library went through design_guides.py --orientation forward (design_guides.py:113-115):
    # rows whose spacer and gene strands differ were dropped there
    assert sum(r["sp_dir"] != r["tar_dir"] for r in table) == 3
    from barcoder_amd import design as D
    cols, table = D.filter_guides(cols, table, 32, orientation="forward")
    got = Counter(tuple(str(r[c]) for c in common) for r in table)
    want = Counter(tuple(r[c] for c in common) for r in gold)
    assert got == want and sum(got.values()) == 772
    # notes: comparable when the reference's site count equals what the plasmids alone show
    mine = {(r["spacer"], r["chr"], str(r["tar_start"]), r["locus_tag"]): r["note"] for r in table}
    n_cmp = 0
    for r in gold:
        note = mine[(r["spacer"], r["chr"], r["tar_start"], r["locus_tag"])]
        if note.split(",")[0] == r["note"].split(",")[0]:
            assert note == r["note"]
            n_cmp += 1
    assert n_cmp >= 600
    text = T.to_tsv(cols, table)
    assert text.count("\n") == 773 and text.startswith("spacer\tlocus_tag\tgene\tchr\tpam\ttarget")


def test_crispri_library_on_gpu_hits(gpu_device, plasmids):
    """testing_grounds.py:20-43 with the Bowtie / pysam / pyranges stages replaced: hits from the GPU, joined with
    the feature table, through `crispri.CRISPRiLibrary` + `pam.PAMFinder` — held against what the REFERENCE'S OWN
    `CRISPRiLibrary` (CRISPRiLibrary.py:4-120) and `PAMFinder` (PAMProcessor.py:60-97) computed from the same
    frame (tests/golden/crispri.json.gz, written by oracle/gen_golden.py with the reference imported unmodified;
    the frame there is built by plain string search, independently of this package)."""
    import gzip
    import json
    from barcoder_amd import targets as T
    from barcoder_amd.crispri import CRISPRiLibrary, joined_frame
    from barcoder_amd.pam import PAMFinder
    with gzip.open(os.path.join(GOLDEN, "crispri.json.gz"), "rt") as f:
        gold = json.load(f)
    linear = [T.Chromosome(c.name, c.seq, False) for c in plasmids]  # BowtieRunner.make_fasta writes the bare records
    lens = {c.name: len(c.seq) for c in linear}
    parts = T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"))  # per part, like GenBankParser.ranges
    spacers = list(dict.fromkeys(r["spacer"] for r in _golden_rows()))
    rows = T.find_hits(spacers, linear, pam="", num_mismatches=0, device=gpu_device)
    df = joined_frame(rows, parts, lens)
    as_records = lambda d: json.loads(d.to_json(orient="records"))
    assert as_records(df) == gold["frame"] and len(df) == 1540
    records = {c.name: c.seq for c in linear}
    for case in gold["cases"]:
        lib = CRISPRiLibrary(df.copy(), PAMFinder(records, case["pam"], case["direction"]))
        assert list(lib.targets_df["PAM"]) == case["PAM"]
        assert [bool(x) for x in lib.targets_df["Targeting"]] == case["Targeting"]
        for view in ("source_unique_targets", "mapped_targets", "unique_targets", "unambiguous_targets"):
            assert as_records(getattr(lib, view)) == case[view], (case["pam"], case["direction"], view)
        assert [int(i) for i in lib.unambiguous_targets.index] == case["unambiguous_index"]
    assert 0 < sum(gold["cases"][1]["Targeting"]) < sum(gold["cases"][0]["Targeting"])  # NGG turns most sites away


def test_design_guides_pipeline(gpu_device, plasmids, oracle):
    """design_guides.main on the GPU path (sliding PAM scan -> hits -> annotation -> filters) on the two smallest
    plasmids: the spacer universe is the oracle's sliding-scan set, every row obeys the filters asked for, and the
    tiling walk leaves guides at least `tile_size` apart along each gene."""
    from barcoder_amd import design as D, targets as T
    small = sorted(plasmids, key=lambda c: len(c.seq))[:2]
    genes = [g for g in T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"), {c.name: len(c.seq) for c in plasmids})
             if g.chrom in {c.name for c in small}]
    cols, full = D.design_guides(small, genes, "NGG", 20, device=gpu_device)
    want = set()
    for c in small:
        topo = c.seq + (c.seq[:100_000] if c.circular else "")
        want |= set(oracle.sliding_sites_py(topo, 20, "NGG", "downstream"))
    assert {r["spacer"] for r in full} == want and len(want) > 1000
    cols, rows = D.design_guides(small, genes, "NGG", 20, device=gpu_device, orientation="forward", omit_offtargets=True,
                                 omit_intergenic=True, full_overlap=True, tile_size=50)
    assert rows and all(r["sp_dir"] == r["tar_dir"] and "intergenic" not in r["note"] for r in rows)
    assert all(r["note"].startswith("1 site") for r in rows)
    # the tiling walk: a subset of the untiled result that keeps every gene's first guide; a guide chosen along one
    # gene stays in the table for every gene it overlaps, so spacing is only guaranteed along the gene that chose it
    _, untiled = D.design_guides(small, genes, "NGG", 20, device=gpu_device, orientation="forward", omit_offtargets=True,
                                 omit_intergenic=True, full_overlap=True)
    key = lambda r: (r["spacer"], r["locus_tag"], r["tar_start"])
    assert {key(r) for r in rows} < {key(r) for r in untiled}
    first = {}
    for r in untiled:  # with full_overlap the walk starts at a gene's first fully overlapping guide (:246-255)
        if r["overlap"] == 20:
            first[r["locus_tag"]] = min(first.get(r["locus_tag"], 1 << 30), r["offset"])
    got_first = {}
    for r in rows:
        if r["overlap"] == 20:
            got_first[r["locus_tag"]] = min(got_first.get(r["locus_tag"], 1 << 30), r["offset"])
    assert got_first == first
    assert len(rows) < len(untiled) // 2 and len(untiled) < len(full)
    assert T.to_tsv(cols, rows).startswith("spacer\tlocus_tag\tgene\tchr")


def _plasmid_genbank(tmp_path, plasmids):
    """The shipped plasmids (sequence + SnapGene gene table) in the CLI's input format."""
    from barcoder_amd import genbank as GB
    feats = {c.name: [] for c in plasmids}
    with open(os.path.join(GOLDEN, "plasmid_genes.tsv")) as fh:
        next(fh)
        for line in fh:
            c, parts, strand, lt, gn = line.rstrip("\n").split("\t")
            iv = [tuple(int(x) for x in part.split("-")) for part in parts.split(";")]
            if int(strand) == -1:
                iv = iv[::-1]  # complement(join(...)): parts last-first, as the reader hands them out
            q = {}
            if lt != "None":
                q["locus_tag"] = [lt]
            if gn != "None":
                q["gene"] = [gn]
            feats[c].append(GB.GbFeature("gene", iv, int(strand), q))
    recs = [GB.GbRecord(c.name, c.name.split(".")[0], c.seq, "circular", "Zymomonas mobilis", feats[c.name]) for c in plasmids]
    path = tmp_path / "plasmids.gb"
    GB.write_genbank(path, recs)
    return str(path)


def test_genbank_round_trip(tmp_path, plasmids, gpu_device):
    from barcoder_amd import genbank as GB, targets as T
    path = _plasmid_genbank(tmp_path, plasmids)
    recs = GB.read_genbank(path)
    assert [r.id for r in recs] == NAMES and all(r.topology == "circular" for r in recs)
    assert [r.seq for r in recs] == [c.seq.upper() for c in plasmids]
    chroms, genes, lens, org, topo, counts = T.chromosomes_and_genes(path)
    want = T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"), lens)
    key = lambda g: (g.chrom, g.start, g.end, g.strand, g.locus_tag, g.gene)
    assert sorted(map(key, genes)) == sorted(map(key, want)) and len(genes) == 160
    assert set(org.values()) == {"Zymomonas mobilis"} and sum(counts.values()) >= 150


def test_cli_twin_tsv_json_and_non_targeting(gpu_device, tmp_path, plasmids):
    """`python -m barcoder_amd.targets guides.fasta genome.gb NGNC 0 [--json]` (targets.py:864-885): the TSV carries the
    shipped rows; the JSON (to_json, records / indent 4 / nulls) parses to the same table; a guide that is absent
    from the genome comes out as ONE non-targeting row with empty coordinates (targets.py:354-366, :547-557), which
    also switches the `mismatches` column on (its missing value is not 0, :676-677)."""
    import json
    import subprocess
    import sys
    from barcoder_amd import design as D, targets as T
    gold = _golden_rows()
    gb = _plasmid_genbank(tmp_path, plasmids)
    spacers = list(dict.fromkeys(r["spacer"] for r in gold))
    absent = "ACGT" * 8
    assert all(absent not in (c.seq + c.seq[:40]).upper() for c in plasmids)
    fa = tmp_path / "guides.fasta"
    fa.write_text("".join(f">g{i} some description\n{s}\n" for i, s in enumerate(spacers)) + f">nowhere\n{absent}\n>nowhere_twin\n{absent}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda *extra: subprocess.run([sys.executable, "-m", "barcoder_amd.targets", str(fa), gb, "NGNC", "0", *extra],
                                        cwd=root, capture_output=True, text=True, timeout=600)
    tsv = run()
    assert tsv.returncode == 0, tsv.stderr[-1500:]
    lines = tsv.stdout.splitlines()
    cols = lines[0].split("\t")
    assert cols == ["spacer", "locus_tag", "gene", "chr", "count", "pam", "mismatches", "target", "tar_start", "tar_end",
                    "offset", "overlap", "sp_dir", "tar_dir", "note"]
    table = [dict(zip(cols, ln.split("\t"))) for ln in lines[1:]]
    bare = [r for r in table if r["spacer"] == absent]
    assert len(bare) == 1 and bare[0]["note"] == "non-targeting" and bare[0]["count"] == "2"
    assert all(bare[0][c] == "None" for c in cols if c not in ("spacer", "count", "note"))
    assert table[-1] is bare[0] or table[-1]["spacer"] == absent          # missing chr sorts last (pandas sort_values)
    # the targeting part equals the shipped rows once the library's own design filter is applied
    hit_rows = [dict(r, count=None) for r in table if r["spacer"] != absent]
    common = [c for c in cols if c not in ("note", "count")]
    _, kept = D.filter_guides(cols, hit_rows, 32, orientation="forward")
    from collections import Counter
    got = Counter(tuple(str(r[c]) for c in common) for r in kept)
    want = Counter(tuple(r[c] for c in common) for r in gold)
    assert got == want and sum(got.values()) == 772
    assert "Non-targeting Barcodes" in tsv.stderr and "Writing to TSV" in tsv.stderr
    # JSON: same table, typed
    js = run("--json")
    assert js.returncode == 0, js.stderr[-1500:]
    recs = json.loads(js.stdout)
    assert len(recs) == len(table) and list(recs[0]) == cols
    ints = {"count", "mismatches", "tar_start", "tar_end", "offset", "overlap"}
    for r_t, r_j in zip(table, recs):
        for c in cols:
            want_v = None if r_t[c] == "None" else (int(r_t[c]) if c in ints else r_t[c])
            assert r_j[c] == want_v, (c, r_t[c], r_j[c])
    assert js.stdout.startswith('[\n    {\n        "spacer":"') and '"locus_tag":null' in js.stdout
    # unknown sgRNA extension: message + exit status 1 (targets.py:579-581)
    bad = subprocess.run([sys.executable, "-m", "barcoder_amd.targets", str(tmp_path / "x.txt"), gb, "NGG", "0"], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "File extension not recognized" in bad.stderr


def test_to_json_equals_pandas(gpu_device, plasmids):
    """to_json against pandas' own DataFrame.to_json(orient="records", indent=4) (the call at targets.py:698) on a
    table with missing values, nullable integer columns and characters pandas escapes."""
    import pandas as pd
    from barcoder_amd import targets as T
    gold = _golden_rows()
    spacers = list(dict.fromkeys(r["spacer"] for r in gold))[:40] + ["ACGT" * 8]
    lens = {c.name: len(c.seq) for c in plasmids}
    genes = T.read_genes_tsv(os.path.join(GOLDEN, "plasmid_genes.tsv"), lens)
    rows = T.find_hits(spacers, plasmids, pam="NGNC", num_mismatches=1, device=gpu_device, report_unaligned=True)
    cols, table = T.final_table(T.annotate(rows, genes), lens)
    table[0] = dict(table[0], gene='a/b "q" é')
    df = pd.DataFrame(table).reindex(columns=cols)
    for c in ("mismatches", "offset", "overlap", "tar_start", "tar_end"):
        if c in df.columns:
            df[c] = df[c].astype("Int64")
    assert T.to_json(cols, table) == df.to_json(orient="records", indent=4)
    assert T.to_tsv(cols, table) == df.to_csv(sep="\t", index=False, na_rep="None")


def test_pam_dot_semantics_and_alignment_cap(gpu_device):
    """targets.py:224 turns N into '.', so a genome character that is not ACGT under an N of the PAM still matches
    (PAMProcessor.py:31 / design_guides.py:26 write [ATCG]: it does not, there).  max_alignments keeps the first
    hits per guide in (chromosome, position, strand) order — the documented stand-in for `bowtie -k 100`."""
    from barcoder_amd import pam as P, targets as T
    from barcoder_amd.encoding import pam_masks
    guide = "ACGTTGCAAGGCTTACGATC"
    seq = "TT" + guide + "RGG" + "CCCC" + guide + "AGG" + "TTTT" + guide + "AGA" + "GG"
    ch = [T.Chromosome("c", seq, False)]
    rows = T.find_hits([guide], ch, pam="NGG", num_mismatches=0, device=gpu_device)
    assert [(r["tar_start"], r["pam"]) for r in rows] == [(2, "RGG"), (29, "AGG")]
    fwd, _ = P.pam_scan(seq, "NGG", device=gpu_device)                 # GuideFinder flavour: [ATCG]GG
    assert not fwd[22] and fwd[49]
    assert pam_masks("NGG", n_is_dot=True).tolist() == [31, 4, 4] and pam_masks("NGG").tolist() == [15, 4, 4]
    many = T.Chromosome("m", ("AA" + guide + "TGGC") * 30, False)
    full = T.find_hits([guide], [many], pam="NGG", num_mismatches=0, device=gpu_device)
    capped = T.find_hits([guide], [many], pam="NGG", num_mismatches=0, device=gpu_device, max_alignments=7)
    assert len(full) == 30 and capped == full[:7]
    both = T.find_hits([("a", guide), ("b", guide), ("c", "T" * 20)], [many], pam="NGG", num_mismatches=0,
                       device=gpu_device, report_unaligned=True)
    assert len(both) == 61 and both[-1] == {"name": "c", "spacer": "T" * 20, "len": 20}
