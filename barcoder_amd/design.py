"""Guide-library design on top of the GPU site scan and hit table: the host-side
filters of the reference's ``design_guides.py``.

``design_guides.main`` (design_guides.py:57-310) enumerates every PAM-adjacent
spacer of a genome, runs ``targets.py`` on them (Bowtie) and then thins the
annotated table with a chain of pandas filters.  Here the enumeration is
``pam.sliding_spacers`` (HIP), the hit table ``targets.find_hits`` / ``annotate``
/ ``final_table`` (HIP, Bowtie bypassed) and this module is the filter chain on
the (columns, rows) pair ``final_table`` returns — plain Python, same order,
same rules, including the pandas details that change results:

* the table makes a round trip through TSV in the reference (``to_csv(na_rep=
  "None")`` -> ``read_csv``), so ``None`` is NaN: comparisons with it are False
  (``==``) / True (``!=``), ``groupby`` drops it, ``sort_values`` puts it last;
* at the end every numeric column is ``to_numeric(...).fillna(0).astype(int)``
  (:296-302): a missing number prints as 0, a missing string as ``None``.
"""
from __future__ import annotations

import re

from . import targets as T

NUMERIC = ("count", "mismatches", "tar_start", "tar_end", "offset", "overlap")
_SITE, _GENE, _INTER = re.compile(r"(\d+) site"), re.compile(r"(\d+) gene"), re.compile(r"(\d+) intergenic")


def _num(note, rx):
    m = rx.search(note or "")
    return int(m.group(1)) if m else 0


def _by_gene(rows):
    """rows sorted by (locus_tag, offset) and grouped by locus_tag; rows without a
    locus_tag belong to no group (pandas groupby drops NaN keys)."""
    have = [r for r in rows if r.get("locus_tag") is not None]
    have.sort(key=lambda r: (r["locus_tag"], float("inf") if r.get("offset") is None else r["offset"]))
    groups, cur = [], None
    for r in have:
        if cur is None or r["locus_tag"] != cur[0]:
            cur = (r["locus_tag"], [])
            groups.append(cur)
        cur[1].append(r)
    return groups


def filter_guides(cols, rows, barcode_length: int, orientation: str | None = None, omit_offtargets: bool = False,
                  mismatches: int = 0, omit_ambiguous: bool = False, omit_intergenic: bool = False,
                  full_overlap: bool = False, tile_size: int = 0, keep_top: int = 0):
    """design_guides.py:108-310.  Returns (columns, rows) ready for ``targets.to_tsv``."""
    rows = [dict(r) for r in rows]
    for r in rows:                                                    # :110
        if r.get("target") is not None:
            r["target"] = r["target"].upper()
    if orientation == "forward":                                      # :112-114 (NaN == x is False)
        rows = [r for r in rows if r.get("tar_dir") is not None and r.get("sp_dir") == r.get("tar_dir")]
    elif orientation == "reverse":                                    # :116-118 (NaN != x is True)
        rows = [r for r in rows if r.get("sp_dir") is None or r.get("tar_dir") is None or r["sp_dir"] != r["tar_dir"]]
    if omit_offtargets:                                               # :120-137
        rows = [r for r in rows if _num(r.get("note"), _SITE) == 1]
    if mismatches > 0:                                                # :139-147 (the reference needs the column; an
        rows = [r for r in rows if not r.get("mismatches")]           # all-zero run has none: treated as zeros)
    if omit_ambiguous:                                                # :149-196
        rows = [r for r in rows if _num(r.get("note"), _SITE) == 1 and _num(r.get("note"), _GENE) == 1
                and _num(r.get("note"), _INTER) == 0]
    if omit_intergenic:                                               # :198-210
        rows = [r for r in rows if "intergenic" not in (r.get("note") or "")]
    if full_overlap:                                                  # :212-231
        keep = {r["spacer"] for r in rows if r.get("overlap") == barcode_length}
        rows = [r for r in rows if r["spacer"] in keep]
    if tile_size > 0:                                                 # :233-279
        selected = set()
        for _, group in _by_gene(rows):
            if full_overlap:
                full = [r["offset"] for r in group if r.get("overlap") == barcode_length]
                last = full[0] if full else None
            else:
                last = group[0].get("offset")
            if last is not None:
                selected.add(next(r["spacer"] for r in group if r.get("offset") == last))
            for r in group:
                if last is not None and r.get("offset") is not None and r["offset"] >= last + tile_size:
                    selected.add(r["spacer"])
                    last = r["offset"]
        rows = [r for r in rows if r["spacer"] in selected]
    if keep_top > 0:                                                  # :281-309
        if full_overlap:
            rows = [r for r in rows if r.get("overlap") is not None and r["overlap"] >= barcode_length]
        top = set()
        for _, group in _by_gene(rows):
            ranked = sorted((r for r in group if r.get("offset") is not None), key=lambda r: r["offset"])  # stable
            top.update(r["spacer"] for r in ranked[:keep_top])
        rows = [r for r in rows if r["spacer"] in top]
    # :296-302 numeric columns: NaN -> 0, integers
    for c in cols:
        vals = [r.get(c) for r in rows]
        if c in NUMERIC or (vals and all(v is None for v in vals)):
            for r in rows:
                r[c] = 0 if r.get(c) is None else int(r[c])
    # :304-306 sort, NaN last
    def key(r):
        return tuple((1, 0) if r.get(c) is None else (0, r[c]) for c in ("chr", "tar_start", "tar_end", "locus_tag",
                                                                         "offset", "overlap"))
    rows.sort(key=key)
    return cols, rows


def design_guides(chromosomes, genes, pam: str, barcode_length: int, pam_direction: str = "downstream",
                  mismatches: int = 0, device: int = 0, **filters):
    """The whole of design_guides.main on the GPU path: every PAM-adjacent spacer of the (topological) sequences
    (design_guides.py:22-49) -> hits within `mismatches` (targets.py via the matcher, Bowtie bypassed) -> gene
    annotation -> note / column logic -> the filter chain above.  Returns (columns, rows)."""
    from . import pam as P
    found = set()
    for ch in chromosomes:  # the topological form: a circular record carries its first 100 kb again (targets.py:35-49)
        topo = ch.seq + (ch.seq[:100_000] if ch.circular else "")
        found |= P.sliding_spacers(topo, barcode_length, pam, pam_direction, device)
    spacers = sorted(found)
    rows = T.find_hits(spacers, chromosomes, pam=pam, num_mismatches=mismatches, pam_direction=pam_direction,
                       device=device)
    cols, table = T.final_table(T.annotate(rows, genes), {c.name: len(c.seq) for c in chromosomes})
    return filter_guides(cols, table, barcode_length, mismatches=mismatches, **filters)
