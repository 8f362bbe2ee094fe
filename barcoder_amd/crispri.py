"""CRISPRiLibrary with the reference's interface, fed by the GPU hit table.

Mirrors (reference file:line):
  CRISPRiLibrary                       CRISPRiLibrary.py:4-120
  the joined frame it consumes         testing_grounds.py:32-43
      (PySamParser(...).ranges.join(GenBankParser(...).ranges).df: hit columns of
       PySamParser.py:38-46 + feature columns Type / Start_b / End_b / Strand_b)

`joined_frame` builds that frame from `targets.find_hits` rows and `Gene` records:
one row per (hit, overlapping feature) — the whole-record "source" feature first, then
every `gene` feature PART in record order (GenBankParser.py:69-103 lists the parts of a
compound location separately; read the gene table without the origin-merging rule:
`targets.read_genes_tsv(path)`), overlap tested regardless of strand like `PyRanges.join`.
The class itself is the same pandas logic (annotate PAM -> Targeting, source-unique,
mapped with Offset / Overlap, unique, unambiguous); `tests/golden/crispri.json.gz` holds what
the reference's own class computed from the same frame (oracle/gen_golden.py)."""
from __future__ import annotations

import pandas as pd

FRAME_COLUMNS = ["Chromosome", "Start", "End", "Mapped", "Strand", "Barcode", "Mismatches",
                 "Start_b", "End_b", "Strand_b", "Locus_Tag", "Gene", "Type"]


def joined_frame(rows, genes, seq_lens):
    """DataFrame with Chromosome, Start, End, Mapped, Strand, Barcode, Mismatches (PySamParser.py:38-46)
    and Start_b, End_b, Strand_b, Locus_Tag, Gene, Type of every feature the hit overlaps
    (GenBankParser.py:89-97).  Hits in (Chromosome, Start, End, Strand, Barcode) order."""
    by_chr = {}
    for g in genes:
        by_chr.setdefault(g.chrom, []).append(g)
    hits = []
    for r in rows:
        if "chr" not in r:  # the bare row of a guide without a placement: an unmapped read joins nothing
            continue
        a = r["tar_start"] % seq_lens[r["chr"]]  # SAM coordinates are unfolded positions on the reference
        hits.append((r["chr"], a, a + r["len"], "+" if r["sp_dir"] == "F" else "-", r["spacer"], r["mismatches"]))
    hits.sort()
    out = []
    for c, a, b, strand, spacer, mm in hits:
        base = {"Chromosome": c, "Start": a, "End": b, "Mapped": True, "Strand": strand, "Barcode": spacer, "Mismatches": mm}
        out.append(dict(base, Start_b=0, End_b=seq_lens[c], Strand_b="+", Locus_Tag=None, Gene=None, Type="source"))
        for g in by_chr.get(c, ()):
            if g.start < b and g.end > a:
                out.append(dict(base, Start_b=g.start, End_b=g.end, Strand_b="+" if g.strand == 1 else "-" if g.strand == -1 else ".",
                                Locus_Tag=g.locus_tag, Gene=g.gene, Type="gene"))
    return pd.DataFrame(out, columns=FRAME_COLUMNS)


class CRISPRiLibrary:
    def __init__(self, pyranges_df, pam_finder):
        self.targets_df = pyranges_df
        self.pam_finder = pam_finder
        self._annotate_targets()
        self.source_unique_targets = self._get_source_unique_targets()
        self.mapped_targets = self._get_mapped_targets()
        self.unique_targets = self._get_unique_targets()
        self.unambiguous_targets = self._get_unambiguous_targets()

    def _annotate_targets(self):
        df = self.targets_df
        df["PAM"] = [self.pam_finder.get_pam_seq(row) for row in df.itertuples(index=False)]
        df["Targeting"] = [self.pam_finder.pam_matches(p) for p in df["PAM"]]

    def _good(self):
        df = self.targets_df
        return (df["Targeting"] == True) & (df["Mapped"] == True)  # noqa: E712 - mirrors the reference's comparisons

    def _get_source_unique_targets(self):
        df = self.targets_df
        sel = df[(df["Type"] == "source") & self._good()]
        return sel.loc[~sel.duplicated(subset=["Barcode"])].reset_index(drop=True)

    def _get_mapped_targets(self):
        df = self.targets_df
        sel = df[(df["Type"] != "source") & self._good()].copy()
        off = [{"+": r.Start - r.Start_b, "-": r.End_b - r.End}.get(r.Strand_b, None) for r in sel.itertuples(index=False)]
        ov = [max(min(r.End, r.End_b) - max(r.Start, r.Start_b), 0) for r in sel.itertuples(index=False)]
        return sel.assign(Offset=off, Overlap=ov).reset_index(drop=True)

    def _get_unique_targets(self):
        m = self._get_mapped_targets()
        return m[m["Barcode"].isin(self.source_unique_targets.Barcode)] \
            .sort_values(["Chromosome", "Start", "End"]).reset_index(drop=True)

    def _get_unambiguous_targets(self):
        u = self.unique_targets
        return u[~u.duplicated(subset=["Barcode"]).reset_index(drop=True)]
