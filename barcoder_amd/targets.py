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

Gene annotation (``create_locus_map``, targets.py:76-165) needs GenBank features
and is not part of this path; rows carry the columns that do not depend on it.
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
