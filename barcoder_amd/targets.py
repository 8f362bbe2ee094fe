"""Guide -> genome search with PAM filtering on the GPU: the Bowtie bypass.

Replaces, for the hit records themselves, the process boundary of
``targets.run_bowtie_and_parse`` (targets.py:467-539: bowtie-build + ``bowtie -v N``)
and ``BowtieRunner.align`` + ``PySamParser.ranges`` (BowtieRunner.py:104-141,
PySamParser.py:38-46).  Every k-mer window of the packed genome (both strands) is
matched against the guide library by ``bc_match_all`` (all hits <= N
substitutions); the PAM next to each hit comes from the ``bc_pam_scan`` bitmaps.

Mirrors (reference file:line):
  get_diff / get_coords / get_offset / get_overlap / pam_matches   targets.py:184-224
  extract_downstream_pam / extract_upstream_pam                    targets.py:227-307
  row building of parse_sam_output (no annotation join)            targets.py:354-410
  PySamParser.ranges column contract                               PySamParser.py:38-46

  create_locus_map as intervals + gene join (annotate)              targets.py:76-165, :412-462
  drop_duplicates / PAM filter / sort / note / columns / TSV        targets.py:605-701

Gene features come from any source as `Gene` records (tests use the SnapGene
feature tables of the shipped plasmids; a GenBank reader is not included).
"""
from __future__ import annotations

import os
import re
import sys
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .devmem import DevBuf
from .encoding import pam_masks, unpack_kmers
from .library import DeviceLibrary
from .matcher import GpuMatcher
from .pam import PackedGenome, _bits, pack_genome

_RC = str.maketrans("ATCGNatcgn", "TAGCNtagcn")


def rev_comp(s: str) -> str:
    return s[::-1].translate(_RC)


# ------------------------------------------------------------------ small helpers (same names as the reference)
def get_diff(spacer, target):
    """'{target_nt}{1-based pos}{spacer_nt}' joined by ',', None when identical (targets.py:184-190)."""
    diffs = [f"{t}{i + 1}{s}" for i, (t, s) in enumerate(zip(target, spacer)) if t != s]
    return ",".join(diffs) if diffs else None


def diff_from_mask(spacer: str, target: str, mask: int) -> str | None:
    """Same string from the device's mismatch mask (bit i = base i differs)."""
    out = []
    i = 0
    while mask:
        if mask & 1:
            out.append(f"{target[i]}{i + 1}{spacer[i]}")
        mask >>= 1
        i += 1
    return ",".join(out) if out else None


def get_coords(tar_start, tar_end, chrom_length):
    """targets.py:193-202."""
    a = tar_start % chrom_length
    b = tar_end % chrom_length if tar_end % chrom_length != 0 else chrom_length
    return f"({a}..{chrom_length}, 0..{b})" if a > b else f"{a}..{b}"


def get_offset(target_dir, tar_start, tar_end, feature_start, feature_end):
    if target_dir == "F":
        return tar_start - feature_start
    if target_dir == "R":
        return feature_end - tar_end
    return None


def get_overlap(tar_start, tar_end, feature_start, feature_end):
    a, b = max(tar_start, feature_start), min(tar_end, feature_end)
    return b - a if a < b else 0


def pam_matches(pam_pattern, extracted_pam):
    """targets.py:219-224."""
    if not extracted_pam:
        return False
    if pam_pattern == "N" * len(pam_pattern) or not pam_pattern:
        return True
    return bool(re.match(pam_pattern.replace("N", "."), extracted_pam))


@dataclass
class Chromosome:
    name: str
    seq: str
    circular: bool = False


# ------------------------------------------------------------------ the search
_COMP_LUT = np.full(256, ord("N"), np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP_LUT[_a] = _b


def _windows_as_strings(topo: np.ndarray, starts: np.ndarray, width: int, revcomp):
    """topo[starts[i] : starts[i] + width] for every i as upper-case str (reverse-complemented where asked)."""
    if starts.size == 0 or width == 0:
        return [""] * int(starts.size)
    m = topo[starts[:, None] + np.arange(width, dtype=np.int64)[None, :]]
    rc = _COMP_LUT[m[:, ::-1]]
    m = np.where(np.asarray(revcomp, bool)[:, None], rc, m)
    return m.view(f"S{width}").ravel().astype(str).tolist()


def find_hits(guides, chromosomes, pam: str = "", num_mismatches: int = 0, pam_direction: str = "downstream",
              device: int = 0, max_alignments: int | None = None, report_unaligned: bool = False):
    """All placements of every guide on every chromosome within `num_mismatches`
    substitutions whose adjacent PAM matches (`pam` empty or all-N: any).

    guides: sequences, {name: sequence}, or [(name, sequence), ...] (the reads of
    the reference's sgRNA FASTQ: several names may carry one sequence).

    Returns a list of dict rows with the keys parse_sam_output builds before the
    annotation join (targets.py:354-410): name, spacer, len, target, mismatches,
    chr, tar_start, tar_end, sp_dir, pam, coords, type, diff.  Circular
    chromosomes are searched across the origin (the reference's 100 kb overhang,
    targets.py:44-46, shrinks to k + |pam| - 1 bases here) and coordinates are
    folded exactly as targets.py:380-384; every site is reported once
    (the reference removes the overhang duplicates with drop_duplicates, :607).

    report_unaligned: a guide (name) with no placement that survives the PAM test
    gets the bare row {name, spacer, len} the reference emits for unmapped and
    PAM-rejected reads (targets.py:354-366) — what becomes a "non-targeting" row.

    max_alignments: the reference runs `bowtie -k 100` (targets.py:499): at most
    100 alignments per read, which ones being Bowtie's choice.  Here every
    placement is reported (None); an integer keeps the first `max_alignments`
    per guide in (chromosome order, position, strand) order — a deterministic
    stand-in, not Bowtie's selection (documented deviation).

    The whole search runs on the device per chromosome and strand: genome upload,
    PAM bitmaps (bc_pam_scan_dev), windows (bc_windows_dev), PAM-gated all-hits
    search (bc_match_all_pam_dev); only the surviving hit records come back.
    """
    if pam_direction not in ("downstream", "upstream"):
        raise ValueError("pam_direction must be 'upstream' or 'downstream'")
    if isinstance(guides, dict):
        pairs = list(guides.items())
    else:
        guides = list(guides)
        pairs = guides if (guides and isinstance(guides[0], tuple)) else [(g, g) for g in guides]
    pairs = [(nm, sq) for nm, sq in pairs]
    by_seq = {}
    for nm, sq in pairs:
        by_seq.setdefault(sq, []).append(nm)
    rows = []
    # one library per guide length (a FASTA of guides may mix lengths; the packed matcher wants one k)
    searchable = {}
    for sq in by_seq:
        if 1 <= len(sq) <= 32 and set(sq) <= set("ACGT"):
            searchable.setdefault(len(sq), []).append(sq)
    for k, labels in sorted(searchable.items()):
        rows += _find_hits_k(labels, k, chromosomes, pam, num_mismatches, pam_direction, device, max_alignments)
    hit_seqs = {r["spacer"] for r in rows}
    rows = [dict(r, name=nm) for r in rows for nm in by_seq[r["spacer"]]]  # one row per (name, hit) like the SAM query names
    rows.sort(key=lambda r: (r["chr"], min(r["tar_start"], r["tar_end"]), r["spacer"]))  # targets.py:629-630
    if report_unaligned:
        seen = set()
        for nm, sq in pairs:
            if sq not in hit_seqs and (nm, sq) not in seen:
                seen.add((nm, sq))
                rows.append({"name": nm, "spacer": sq, "len": len(sq)})
    return rows


def _find_hits_k(labels, k, chromosomes, pam, num_mismatches, pam_direction, device, max_alignments):
    lib = DeviceLibrary.from_strings(labels, device)
    if num_mismatches >= k:
        raise ValueError(f"{num_mismatches} mismatches on {k}-mers: every window would match")
    pl = len(pam)
    any_pam = (not pam) or pam == "N" * pl
    L = N.load()
    N.check(L.bc_dev_set(device))
    out = []
    per_guide = {}
    for ch in chromosomes:
        n = len(ch.seq)
        if n == 0:
            continue
        wrap = (k + pl - 1) if ch.circular else 0
        g = pack_genome(ch.seq, wrap)
        npos = n if ch.circular else max(0, n - k + 1)
        if npos == 0:
            continue
        nw = g.bases.size
        topo = np.frombuffer((ch.seq + ch.seq[:wrap]).upper().encode("ascii", errors="replace"), np.uint8)
        gate = not any_pam
        with DevBuf.from_numpy(g.bases, device) as d_b, DevBuf.from_numpy(g.invalid, device) as d_i, \
                DevBuf(nw * 4, device) as d_f, DevBuf(nw * 4, device) as d_r, \
                DevBuf(npos * 8, device) as d_w, DevBuf(npos, device) as d_fl, DevBuf(8, device) as d_n:
            if gate:
                # N = '.' here (targets.py:224): an odd genome character under an N of the PAM still matches
                masks = pam_masks(pam, n_is_dot=True)
                N.check(L.bc_pam_scan_dev(d_b.ptr, d_i.ptr, g.total, N.ptr(masks), pl, d_f.ptr, d_r.ptr, None))
            for strand in (0, 1):
                right = (strand == 0) == (pam_direction == "downstream")  # PAM footprint right of the window
                shift = k if right else -pl
                N.check(L.bc_windows_dev(d_b.ptr, d_i.ptr, g.total, None, npos, 0, k, strand, d_w.ptr, d_fl.ptr, None))
                cap = max(1024, 2 * len(labels))
                while True:
                    with DevBuf(cap * 12, device) as d_h:
                        N.check(L.bc_dev_memset(d_n.ptr, 0, 8, None))
                        if gate:
                            N.check(L.bc_match_all_pam_dev(lib.handle, d_w.ptr, d_fl.ptr, npos, int(num_mismatches),
                                                           (d_f if strand == 0 else d_r).ptr, shift, g.total - pl + 1,
                                                           n if ch.circular else 0, d_h.ptr, cap, d_n.ptr, None))
                        else:
                            N.check(L.bc_match_all_dev(lib.handle, d_w.ptr, d_fl.ptr, npos, int(num_mismatches),
                                                       d_h.ptr, cap, d_n.ptr, None))
                        m = int(d_n.to_numpy(np.uint64, 1)[0])
                        if m <= cap:
                            hits = d_h.to_numpy(N.HIT_DTYPE, m) if m else np.zeros(0, N.HIT_DTYPE)
                            break
                    cap = m
                if hits.size == 0:
                    continue
                hits = np.sort(hits, order=("query", "guide"))
                p = hits["query"].astype(np.int64)
                pp = p + shift
                if ch.circular:
                    pp = np.where(pp < 0, pp + n, pp)
                if not gate and pl:  # all-N PAM: only its footprint has to exist (targets.py:219-224)
                    keep = (pp >= 0) & (pp + pl <= g.total)
                    hits, p, pp = hits[keep], p[keep], pp[keep]
                gi = hits["guide"].astype(np.int64)
                dist = hits["dist"].astype(np.int64)
                if max_alignments is not None:
                    sel = np.ones(p.size, bool)
                    for j, gj in enumerate(gi.tolist()):
                        c = per_guide.get(gj, 0)
                        sel[j] = c < max_alignments
                        per_guide[gj] = c + 1
                    p, pp, gi, dist = p[sel], pp[sel], gi[sel], dist[sel]
                rc = np.full(p.size, strand == 1)
                targets = _windows_as_strings(topo, p, k, rc)
                pams = _windows_as_strings(topo, pp, pl, rc) if pl else [None] * p.size
                tar_start = p % n                               # targets.py:380-384
                tar_end = (p + k) % n
                tar_start = np.where(tar_end < tar_start, tar_start - n, tar_start)
                sp_dir = "F" if strand == 0 else "R"
                for j in range(p.size):
                    spacer = labels[int(gi[j])]
                    a, b, d = int(tar_start[j]), int(tar_end[j]), int(dist[j])
                    out.append({"name": spacer, "spacer": spacer, "len": k, "target": targets[j], "mismatches": d,
                                "chr": ch.name, "tar_start": a, "tar_end": b, "sp_dir": sp_dir, "pam": pams[j],
                                "coords": get_coords(a, b, n), "type": "mismatch" if d > 0 else "perfect",
                                "diff": get_diff(spacer, targets[j]) if d else None})
    lib.close()
    return out


def _slice(ch: Chromosome, a: int, b: int) -> str:
    n = len(ch.seq)
    if b <= n:
        return ch.seq[a:b].upper()
    return (ch.seq[a:] + ch.seq[:b - n]).upper()


def genome_windows(g: PackedGenome, npos: int, k: int, strand: int, device: int = 0):
    """k-mer at every position 0..npos-1 of a packed genome (bc_windows_dev)."""
    L = N.load()
    N.check(L.bc_dev_set(device))
    with DevBuf.from_numpy(g.bases, device) as d_b, DevBuf.from_numpy(g.invalid, device) as d_i, \
            DevBuf(npos * 8, device) as d_w, DevBuf(npos, device) as d_f:
        N.check(L.bc_windows_dev(d_b.ptr, d_i.ptr, g.total, None, npos, 0, k, strand, d_w.ptr, d_f.ptr, None))
        return d_w.to_numpy(np.uint64, npos), d_f.to_numpy(np.uint8, npos)


def ranges_table(rows):
    """The 7-column table PySamParser.ranges feeds to CRISPRiLibrary
    (PySamParser.py:38-46): Chromosome, Start, End, Mapped, Strand, Barcode,
    Mismatches — as a list of dicts (pandas.DataFrame(rows) gives the frame)."""
    out = []
    for r in rows:
        a = r["tar_start"]
        out.append({"Chromosome": r["chr"], "Start": a, "End": a + r["len"], "Mapped": True,
                    "Strand": "+" if r["sp_dir"] == "F" else "-", "Barcode": r["spacer"],
                    "Mismatches": r["mismatches"]})
    return out


# ------------------------------------------------------------------ annotation join + output table
@dataclass
class Gene:
    """One `gene` feature: what create_locus_map (targets.py:76-165) keeps per position."""
    chrom: str
    start: int          # 0-based, half-open
    end: int
    strand: int         # +1 / -1
    locus_tag: str | None
    gene: str | None = None


def read_genes_tsv(path, seq_lens=None):
    """chr, parts ("a-b;c-d", 0-based half-open), strand, locus_tag, gene -> list[Gene].
    Like create_locus_map: a feature in several parts contributes one Gene per part with that
    part's own bounds (targets.py:129-148), except a compound location that touches both ends of
    the record, which becomes ONE interval from the start of its last part to the end of its first
    part + len(record) (targets.py:105-128).  `seq_lens` ({chr: length}) enables that rule."""
    genes = []
    with open(path) as fh:
        next(fh)
        for line in fh:
            c, parts, strand, lt, gn = line.rstrip("\n").split("\t")
            iv = [tuple(int(x) for x in part.split("-")) for part in parts.split(";")]
            lt, gn = (None if lt == "None" else lt), (None if gn == "None" else gn)
            n = (seq_lens or {}).get(c)
            if len(iv) > 1 and n is not None and any(a == 0 or b == n for a, b in iv):
                tail = next((p for p in iv if p[1] == n), None)
                head = next((p for p in iv if p[0] == 0), None)
                if tail and head:
                    genes.append(Gene(c, tail[0], head[1] + n, int(strand), lt, gn))
                    continue
            genes += [Gene(c, a, b, int(strand), lt, gn) for a, b in iv]
    return genes


def annotate(rows, genes):
    """The gene join of parse_sam_output (targets.py:412-462): a hit row is repeated for every gene
    that covers at least one base of [tar_start, tar_end); rows without a gene keep None fields.
    The reference walks a per-base dict; interval overlap gives the same set.  Bare rows of
    unaligned guides ({name, spacer, len}, targets.py:354-366) pass through untouched."""
    by_chr = {}
    for g in genes:
        by_chr.setdefault(g.chrom, []).append(g)
    starts, longest = {}, {}
    for c, lst in by_chr.items():
        lst.sort(key=lambda g: g.start)
        starts[c] = np.array([g.start for g in lst])
        longest[c] = max((g.end - g.start for g in lst), default=0)  # a gene that reaches a starts after a - longest
    out = []
    for r in rows:
        if r.get("target") is None:
            out.append(dict(r))
            continue
        lst = by_chr.get(r["chr"], [])
        a, b = r["tar_start"], r["tar_end"]
        hit = []
        if lst:
            st = starts[r["chr"]]
            hi = int(np.searchsorted(st, b, side="left"))  # genes starting before tar_end
            lo = int(np.searchsorted(st, a - longest[r["chr"]], side="left"))
            hit = [g for g in lst[lo:hi] if g.end > a and g.start < b and max(a, g.start) < min(b, g.end)]
        if not hit:
            out.append(dict(r, locus_tag=None, gene=None, offset=None, overlap=None, tar_dir=None))
            continue
        seen = set()
        for g in hit:
            key = (g.locus_tag, g.gene, g.start, g.end, g.strand)
            if key in seen:
                continue
            seen.add(key)
            tdir = "F" if g.strand == 1 else "R" if g.strand == -1 else None
            out.append(dict(r, locus_tag=g.locus_tag, gene=g.gene if g.gene else g.locus_tag,
                            offset=get_offset(tdir, a, b, g.start, g.end), overlap=get_overlap(a, b, g.start, g.end),
                            tar_dir=tdir))
    return out


def create_note(sites, genes, intergenic):
    """targets.py:547-557."""
    if sites <= 0:
        return "non-targeting"
    parts = [f"{sites} {'site' if sites == 1 else 'sites'}"]
    if genes > 0:
        parts.append(f"{genes} {'gene' if genes == 1 else 'genes'}")
    if intergenic > 0:
        parts.append(f"{intergenic} intergenic")
    return ", ".join(parts)


ALL_KEYS = ["name", "spacer", "len", "target", "mismatches", "chr", "tar_start", "tar_end", "sp_dir", "pam",
            "coords", "type", "diff", "locus_tag", "gene", "offset", "overlap", "tar_dir"]


def final_table(rows, seq_lens):
    """The post-processing of targets.main (targets.py:605-688) on annotated rows:
    de-duplicate, drop the bare rows of spacers that do target, sort by (chr, min_tar, spacer)
    with missing values last (pandas sort_values), per-spacer site / gene / intergenic counts ->
    note, column selection.  Returns (columns, list of row dicts)."""
    def freeze(r, keys):
        return tuple(map(r.get, keys))  # (missing keys are None, as r.get(k) for every k)
    rows = list({freeze(r, ALL_KEYS): r for r in rows}.values())                      # drop_duplicates (:607)
    targeting = {r["spacer"] for r in rows if r.get("target") is not None}            # filter_offtargets_by_pam
    rows = [r for r in rows if not (r.get("target") is None and r["spacer"] in targeting)]
    if not any(r.get("target") is not None for r in rows):
        # the reference dies here: no row has a `target` column (KeyError in filter_offtargets_by_pam, :609-621)
        raise KeyError("target")

    def min_tar(r):                                                                    # :624-628
        if r.get("tar_start") is None:
            return float("inf")
        return r["tar_start"] - seq_lens[r["chr"]] if r["tar_start"] > r["tar_end"] else r["tar_start"]
    # sort_values(by=[chr, min_tar, spacer]): NaN last in each key
    rows.sort(key=lambda r: (r.get("chr") is None, r.get("chr") or "", min_tar(r), r["spacer"]))   # :629-630
    names = {}
    for r in rows:
        names.setdefault(r["spacer"], set()).add(r["name"])
    keys_wo_name = [k for k in ALL_KEYS if k != "name"]
    rows = list({freeze(r, keys_wo_name): r for r in rows}.values())                  # :635
    sites, genes, inter = {}, {}, {}
    for r in rows:
        if r.get("target") is not None:
            sites.setdefault(r["spacer"], set()).add(f"{r['chr']}_{r['coords']}")
            if r.get("locus_tag") is None:
                inter[r["spacer"]] = inter.get(r["spacer"], 0) + 1
        if r.get("locus_tag") is not None:
            genes[r["spacer"]] = genes.get(r["spacer"], 0) + 1
    out = []
    for r in rows:
        sp = r["spacer"]
        out.append(dict(r, count=len(names[sp]),
                        note=create_note(len(sites.get(sp, ())), genes.get(sp, 0), inter.get(sp, 0))))
    cols = ["spacer", "locus_tag", "gene", "chr"]
    if any(r["count"] != 1 for r in out):                                              # :672-673
        cols.append("count")
    pams = {r.get("pam") for r in out if r.get("pam") is not None}
    if len(pams) > 1:                                 # not (pam.isnull().all() or pam.nunique() == 1), :674-675
        cols.append("pam")
    if any(r.get("mismatches") != 0 for r in out):    # not (mismatches == 0).all(): a missing value is not 0, :676-677
        cols.append("mismatches")
    cols += ["target", "tar_start", "tar_end", "offset", "overlap", "sp_dir", "tar_dir", "note"]
    return cols, out


def to_tsv(cols, rows) -> str:
    """DataFrame.to_csv(sep='\\t', index=False, na_rep='None') of targets.py:701."""
    import csv
    import io
    buf = io.StringIO()
    w = csv.writer(buf, delimiter="\t", lineterminator="\n", quoting=csv.QUOTE_MINIMAL)  # pandas' writer settings
    w.writerow(cols)
    for r in rows:
        w.writerow(["None" if r.get(c) is None else str(r[c]) for c in cols])
    return buf.getvalue()


def _json_str(s: str) -> str:
    """a JSON string the way pandas' writer escapes it (ASCII only, '/' escaped)"""
    from json.encoder import encode_basestring_ascii  # (the C-accelerated escaper json.dumps itself uses)
    return encode_basestring_ascii(str(s)).replace("/", "\\/")


def to_json(cols, rows) -> str:
    """DataFrame.to_json(orient="records", indent=4) of targets.py:696-698: one object per row in
    column order, missing values (and the nullable Int64 columns' <NA>) as null, `"key":value`
    without a space, 4-space indentation."""
    if not rows:
        return "[\n\n]"
    from json.encoder import encode_basestring_ascii as esc
    keys = [f"        {_json_str(c)}:" for c in cols]
    memo = {}  # spacers, genes, locus tags and notes repeat from row to row

    def text(v):
        if v is None:
            return "null"
        if isinstance(v, str):
            t = memo.get(v)
            if t is None:
                t = memo[v] = esc(v).replace("/", "\\/")
            return t
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, (int, np.integer)):
            return str(int(v))
        if isinstance(v, (float, np.floating)):
            return "null" if v != v else repr(float(v))
        return _json_str(v)

    objs = []
    for r in rows:
        get = r.get
        objs.append("    {\n" + ",\n".join([k + text(get(c)) for k, c in zip(keys, cols)]) + "\n    }")
    return "[\n" + ",\n".join(objs) + "\n]"


# ------------------------------------------------------------------ GenBank input + the script's main()
def chromosomes_and_genes(genbank_path):
    """(chromosomes, genes, seq_lens, organisms, topologies, gene_counts) from a GenBank file: what
    create_topological_fasta (targets.py:35-56) and create_locus_map (:76-165) derive from it.  A `gene`
    feature in several parts contributes one interval per part, except a compound location that touches both
    ends of the record, which becomes ONE interval from the start of its last part to the end of its first
    part + len(record) (:105-128)."""
    from .genbank import read_genbank
    chroms, genes, seq_lens, organisms, topologies, counts = [], [], {}, {}, {}, {}
    for rec in read_genbank(genbank_path):
        n = len(rec.seq)
        chroms.append(Chromosome(rec.id, rec.seq, rec.topology == "circular"))
        seq_lens[rec.id], organisms[rec.id], topologies[rec.id] = n, rec.organism, rec.topology
        cnt = 0
        for ft in rec.features:
            if ft.key != "gene":
                continue
            cnt += 1
            lt = (ft.qualifiers.get("locus_tag") or [None])[0]
            gn = (ft.qualifiers.get("gene") or [None])[0]
            parts = ft.parts
            if len(parts) > 1 and any(a == 0 or b == n for a, b in parts):
                tail = next((p for p in parts if p[1] == n), None)
                head = next((p for p in parts if p[0] == 0), None)
                if tail is None or head is None:
                    raise ValueError(f"gene {lt}: a compound location that touches only one end of {rec.id}")  # the reference's next() raises there (:112-119)
                genes.append(Gene(rec.id, tail[0], head[1] + n, ft.strand, lt, gn))
                continue
            genes += [Gene(rec.id, a, b, ft.strand, lt, gn) for a, b in parts]
        counts[rec.id] = cnt
    return chroms, genes, seq_lens, organisms, topologies, counts


def read_guides(path):
    """[(name, sequence)] of the sgRNA file as the reference feeds it to Bowtie (targets.py:567-580):
    a FASTQ as it is, a FASTA (optionally .gz) record by record; the name is the first word of the header."""
    import gzip
    base = os.path.basename(path)
    op = gzip.open if path.endswith(".gz") else open
    out = []
    if ".fastq" in base:
        with op(path, "rt") as fh:
            lines = fh.read().splitlines()
        for i in range(0, len(lines) - 1, 4):
            if lines[i].startswith("@"):
                out.append((lines[i][1:].split()[0] if lines[i][1:].split() else "", lines[i + 1].strip()))
        return out
    if base.endswith(".fasta") or base.endswith(".fasta.gz"):
        name, seq = None, []
        with op(path, "rt") as fh:
            for line in fh:
                line = line.rstrip("\r\n")
                if line.startswith(">"):
                    if name is not None:
                        out.append((name, "".join(seq)))
                    w = line[1:].split()
                    name, seq = (w[0] if w else ""), []
                elif name is not None:
                    seq.append(line.strip())
        if name is not None:
            out.append((name, "".join(seq)))
        return out
    raise ValueError("File extension not recognized.")


def main(args, device: int = 0, out=None):
    """The script (targets.py:560-701, :864-885): sgRNA FASTA/FASTQ + GenBank genome + PAM + mismatches ->
    the annotated hit table on stdout as TSV, or JSON with --json; progress and a short summary on stderr.
    Bowtie is bypassed: the search is find_hits."""
    out = out or sys.stdout
    log = lambda msg: print(msg, file=sys.stderr)  # noqa: E731
    log("Initializing barcode target seeker")
    try:
        guides = read_guides(args.sgrna_file)
    except ValueError as e:
        log(str(e))
        sys.exit(1)
    log("Generating topological coordinate maps...")
    chroms, genes, seq_lens, organisms, topologies, gene_counts = chromosomes_and_genes(args.genome_file)
    log("Aligning annotations to genome...")
    rows = find_hits([(nm, sq.upper()) for nm, sq in guides], chroms, pam=args.pam, num_mismatches=args.mismatches,
                     pam_direction=args.pam_direction, device=device, report_unaligned=True)
    if not rows:
        raise RuntimeError("No results were returned from the search. Check your input files and parameters.")
    log("Finding matches...")
    try:
        cols, table = final_table(annotate(rows, genes), seq_lens)
    except KeyError as e:
        log(f"The following critical attribute is missing for every barcode: {e}")
        log(f"Are you sure your PAM should be '{args.pam}' '{args.pam_direction}' of the spacer?")
        log(f"The genome file '{args.genome_file}' has definitions for these chromosomes:")
        import json
        log(json.dumps(organisms, indent=4))
        sys.exit(1)
    if getattr(args, "json", False):
        log("Writing to JSON...")
        print(to_json(cols, table), file=out)
    else:
        log("Writing to TSV...")
        out.write(to_tsv(cols, table))
    lens = sorted({r["len"] for r in table})
    targeting = {r["spacer"] for r in table if r.get("target") is not None}
    coords_of = {}
    for r in table:
        if r.get("target"):
            coords_of.setdefault(r["spacer"], set()).add(r["coords"])
    summary = [("Barcodes", os.path.basename(args.sgrna_file)), ("Genbank Genome File", os.path.basename(args.genome_file)),
               ("PAM", args.pam), ("PAM Direction", args.pam_direction.capitalize()),
               ("Number of Mismatches", args.mismatches), ("Spacer Lengths", ",".join(str(x) for x in lens)),
               ("Organism", ", ".join(sorted({str(o) for o in organisms.values()})) or "Unknown"),
               ("Topology", ", ".join(sorted({str(t) for t in topologies.values()}))),
               ("Chromosomes", len(seq_lens)), ("Total Genes", f"{sum(gene_counts.values()):,}"),
               ("Chromosomes Targeted", len({r["chr"] for r in table if r.get("chr")})),
               ("Genes Targeted", len({r["locus_tag"] for r in table if r.get("locus_tag")})),
               ("Unique Barcodes", len({r["spacer"] for r in table})),
               ("Intergenic Barcodes", len({r["spacer"] for r in table if r.get("locus_tag") is None and r.get("chr")})),
               ("Off-targeting Barcodes", sum(1 for sp in targeting if len(coords_of.get(sp, ())) > 1)),
               ("Non-targeting Barcodes", len({r["spacer"] for r in table if r.get("target") is None}))]
    for k_, v in summary:
        log(f"{k_:>30}  {v}")


def _cli():
    import argparse
    p = argparse.ArgumentParser(description="Map barcodes to a circular genome")
    p.add_argument("sgrna_file", help="Path to sgrna_fasta_file", type=str)
    p.add_argument("genome_file", help="Path to genome_gb_file", type=str)
    p.add_argument("pam", help="PAM sequence", type=str)
    p.add_argument("mismatches", help="Number of allowed mismatches", type=int)
    p.add_argument("--pam_direction", choices=["upstream", "downstream"], default="downstream",
                   help="Direction of the PAM sequence")
    p.add_argument("--json", action="store_true", default=False, help="Output results in JSON format")
    p.add_argument("--device", type=int, default=0)
    a = p.parse_args()
    main(a, device=a.device)


if __name__ == "__main__":
    _cli()
