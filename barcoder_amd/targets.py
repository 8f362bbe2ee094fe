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

import re
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
def find_hits(guides, chromosomes, pam: str = "", num_mismatches: int = 0, pam_direction: str = "downstream",
              device: int = 0):
    """All placements of every guide on every chromosome within `num_mismatches`
    substitutions whose adjacent PAM matches (`pam` empty or all-N: any).

    Returns a list of dict rows with the keys parse_sam_output builds before the
    annotation join (targets.py:354-410): name, spacer, len, target, mismatches,
    chr, tar_start, tar_end, sp_dir, pam, coords, type, diff.  Circular
    chromosomes are searched across the origin (the reference's 100 kb overhang,
    targets.py:44-46, shrinks to k + |pam| - 1 bases here) and coordinates are
    folded exactly as targets.py:380-384; every site is reported once
    (the reference removes the overhang duplicates with drop_duplicates, :607).
    """
    if pam_direction not in ("downstream", "upstream"):
        raise ValueError("pam_direction must be 'upstream' or 'downstream'")
    names = None
    if isinstance(guides, dict):
        names, guides = list(guides.keys()), list(guides.values())
    guides = list(guides)
    labels = list(dict.fromkeys(guides))
    k = len(labels[0])
    lib = DeviceLibrary.from_strings(labels, device)
    matcher = GpuMatcher(lib)
    pl = len(pam)
    any_pam = (not pam) or pam == "N" * pl
    rows = []
    L = N.load()
    for ch in chromosomes:
        n = len(ch.seq)
        wrap = (k + pl - 1) if ch.circular else 0
        g = pack_genome(ch.seq, wrap)
        npos = n if ch.circular else max(0, n - k + 1)
        if npos == 0:
            continue
        # PAM bitmaps over the padded sequence
        if not any_pam:
            masks = pam_masks(pam)
            fwd_w = np.zeros(g.bases.size, np.uint32)
            rev_w = np.zeros(g.bases.size, np.uint32)
            N.check(L.bc_pam_scan_host(device, N.ptr(g.bases), N.ptr(g.invalid), g.total, N.ptr(masks), pl,
                                       N.ptr(fwd_w), N.ptr(rev_w)))
            pam_f, pam_r = _bits(fwd_w, g.total), _bits(rev_w, g.total)
        for strand in (0, 1):
            words, flags = genome_windows(g, npos, k, strand, device)
            hits = matcher.all_hits(words, flags, num_mismatches)
            if hits.size == 0:
                continue
            p = hits["query"].astype(np.int64)
            # PAM footprint on forward coordinates for this strand / direction
            if (strand == 0) == (pam_direction == "downstream"):
                pp = p + k           # right of the window
                pp_ok = pp + pl <= g.total
            else:
                pp = p - pl          # left of the window
                if ch.circular:
                    pp = np.where(pp < 0, pp + n, pp)
                pp_ok = pp >= 0
            if any_pam:
                keep = pp_ok if pl else np.ones(p.size, bool)
            else:
                bm = pam_f if strand == 0 else pam_r
                keep = pp_ok & bm[np.clip(pp, 0, g.total - 1)]
            for h, pam_at in zip(hits[keep], pp[keep]):
                pos, gi, dist = int(h["query"]), int(h["guide"]), int(h["dist"])
                topo_end = pos + k
                foot = _slice(ch, pos, topo_end)
                target = foot if strand == 0 else rev_comp(foot)
                ext = _slice(ch, int(pam_at), int(pam_at) + pl) if pl else None
                if ext is not None and strand == 1:
                    ext = rev_comp(ext)
                tar_start, tar_end = pos % n, topo_end % n  # targets.py:380-384
                if tar_end < tar_start:
                    tar_start -= n
                spacer = labels[gi]
                rows.append({
                    "name": spacer, "spacer": spacer, "len": k, "target": target, "mismatches": dist,
                    "chr": ch.name, "tar_start": tar_start, "tar_end": tar_end, "sp_dir": "F" if strand == 0 else "R",
                    "pam": ext, "coords": get_coords(tar_start, tar_end, n),
                    "type": "mismatch" if dist > 0 else "perfect", "diff": get_diff(spacer, target),
                })
    if names is not None:  # one row per (name, hit) like the SAM query names
        by_seq = {}
        for nm, sq in zip(names, guides):
            by_seq.setdefault(sq, []).append(nm)
        rows = [dict(r, name=nm) for r in rows for nm in by_seq[r["spacer"]]]
    rows.sort(key=lambda r: (r["chr"], min(r["tar_start"], r["tar_end"]), r["spacer"]))  # targets.py:629-630
    return rows


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
    The reference walks a per-base dict; interval overlap gives the same set."""
    by_chr = {}
    for g in genes:
        by_chr.setdefault(g.chrom, []).append(g)
    starts = {}
    for c, lst in by_chr.items():
        lst.sort(key=lambda g: g.start)
        starts[c] = np.array([g.start for g in lst])
    out = []
    for r in rows:
        lst = by_chr.get(r["chr"], [])
        a, b = r["tar_start"], r["tar_end"]
        hit = []
        if lst:
            hi = int(np.searchsorted(starts[r["chr"]], b, side="left"))  # genes starting before tar_end
            hit = [g for g in lst[:hi] if g.end > a and g.start < b and max(a, g.start) < min(b, g.end)]
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


def final_table(rows, seq_lens):
    """The post-processing of targets.main (targets.py:605-688) on annotated rows:
    de-duplicate, drop PAM-less rows of spacers that do target, sort by (chr, min_tar, spacer),
    per-spacer site / gene / intergenic counts -> note, column selection.
    Returns (columns, list of row dicts)."""
    def freeze(r, keys):
        return tuple(r.get(k) for k in keys)
    all_keys = ["name", "spacer", "len", "target", "mismatches", "chr", "tar_start", "tar_end", "sp_dir", "pam",
                "coords", "type", "diff", "locus_tag", "gene", "offset", "overlap", "tar_dir"]
    rows = list({freeze(r, all_keys): r for r in rows}.values())                      # drop_duplicates (:607)
    targeting = {r["spacer"] for r in rows if r.get("target") is not None}            # filter_offtargets_by_pam
    rows = [r for r in rows if not (r.get("target") is None and r["spacer"] in targeting)]

    def min_tar(r):                                                                    # :624-628
        if r.get("tar_start") is None:
            return float("inf")
        return r["tar_start"] - seq_lens[r["chr"]] if r["tar_start"] > r["tar_end"] else r["tar_start"]
    rows.sort(key=lambda r: (r.get("chr") or "", min_tar(r), r["spacer"]))            # :629-630
    names = {}
    for r in rows:
        names.setdefault(r["spacer"], set()).add(r["name"])
    keys_wo_name = [k for k in all_keys if k != "name"]
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
    if any(r["count"] != 1 for r in out):
        cols.append("count")
    pams = {r.get("pam") for r in out if r.get("pam") is not None}
    if not (not pams or (len(pams) == 1 and all(r.get("pam") is not None for r in out))):
        cols.append("pam")
    if any(r.get("mismatches") not in (0, None) for r in out):
        cols.append("mismatches")
    cols += ["target", "tar_start", "tar_end", "offset", "overlap", "sp_dir", "tar_dir", "note"]
    return cols, out


def to_tsv(cols, rows) -> str:
    """DataFrame.to_csv(sep='\\t', index=False, na_rep='None') of targets.py:701."""
    lines = ["\t".join(cols)]
    for r in rows:
        lines.append("\t".join("None" if r.get(c) is None else str(r[c]) for c in cols))
    return "\n".join(lines) + "\n"
