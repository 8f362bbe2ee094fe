"""Minimal GenBank flat-file reader: what targets.py needs from
``SeqIO.parse(handle, "genbank")`` (targets.py:35-56 create_topological_fasta,
:76-165 create_locus_map, :168-173 get_true_chrom_lengths) — record id, sequence,
topology, organism and the `gene` features with their locations and
/locus_tag, /gene qualifiers.  Biopython is not a dependency of this package
(and is absent from the build image: parity with its parser is unpinned; the
tests feed files written from the shipped plasmids' sequences and feature tables).

Location grammar handled: ``a..b``, ``a``, ``<a..>b``, ``a^b``, ``complement(...)``,
``join(...)`` / ``order(...)`` in any nesting.  Coordinates come out 0-based,
half-open, like Biopython's FeatureLocation.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass, field


@dataclass
class GbFeature:
    key: str
    parts: list            # [(start, end), ...] 0-based half-open, in the order Biopython keeps them
    strand: int | None     # +1 / -1 / None (mixed)
    qualifiers: dict = field(default_factory=dict)   # name -> list of values


@dataclass
class GbRecord:
    id: str
    name: str
    seq: str
    topology: str | None
    organism: str | None
    features: list


_RANGE = re.compile(r"^[<>]?(\d+)(?:(\.\.|\^|\.)[<>]?(\d+))?$")


def _split_top(s: str):
    """split on commas that are not inside parentheses"""
    out, depth, cur = [], 0, []
    for c in s:
        if c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
    out.append("".join(cur))
    return out


def parse_location(loc: str):
    """-> (parts, strand).  complement(join(a,b)) lists its parts last-first, as Biopython does."""
    loc = loc.replace(" ", "")
    if loc.startswith("complement(") and loc.endswith(")"):
        parts, strand = parse_location(loc[len("complement("):-1])
        strands = {-s if s else s for s in ([strand] if not isinstance(strand, list) else strand)}
        return parts[::-1], (strands.pop() if len(strands) == 1 else None)
    for kw in ("join(", "order(", "bond("):
        if loc.startswith(kw) and loc.endswith(")"):
            parts, strands = [], set()
            for piece in _split_top(loc[len(kw):-1]):
                p, s = parse_location(piece)
                parts += p
                strands.add(s)
            return parts, (strands.pop() if len(strands) == 1 else None)
    if ":" in loc:  # reference to another entry: not on this sequence
        raise ValueError(f"remote location {loc!r}")
    m = _RANGE.match(loc)
    if not m:
        raise ValueError(f"cannot parse location {loc!r}")
    a = int(m.group(1))
    b = int(m.group(3)) if m.group(3) else a
    if m.group(2) == "^":  # between two bases: empty interval at b - 1
        return [(a, a)], 1
    return [(a - 1, b)], 1


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")


def read_genbank(path):
    """Every record of a GenBank flat file as GbRecord."""
    records = []
    with _open(path) as fh:
        lines = fh.read().splitlines()
    i, n = 0, len(lines)
    while i < n:
        if not lines[i].startswith("LOCUS"):
            i += 1
            continue
        toks = lines[i].split()
        name = toks[1] if len(toks) > 1 else ""
        topology = "circular" if "circular" in toks else "linear" if "linear" in toks else None
        rid, accession, organism = None, None, None
        features, seq = [], []
        i += 1
        while i < n and not lines[i].startswith("//"):
            ln = lines[i]
            if ln.startswith("VERSION"):
                t = ln.split()
                rid = t[1] if len(t) > 1 else None
            elif ln.startswith("ACCESSION"):
                t = ln.split()
                accession = t[1] if len(t) > 1 else None
            elif ln.startswith("  ORGANISM"):
                organism = ln[12:].strip()
            elif ln.startswith("FEATURES"):
                i += 1
                cur = None        # [key, location text, qualifier lines]
                while i < n and lines[i][:1] == " ":
                    fl = lines[i]
                    if len(fl) > 5 and fl[5] != " ":          # new feature: key at column 6
                        if cur:
                            features.append(cur)
                        cur = [fl[5:21].strip(), fl[21:].strip(), []]
                    elif cur is not None:
                        body = fl[21:] if len(fl) > 21 else fl.strip()
                        if body.startswith("/"):
                            cur[2].append(body)
                        elif cur[2]:                           # a qualifier value continues
                            glue = "" if cur[2][-1].startswith("/translation") else " "
                            cur[2][-1] += glue + body.strip()
                        else:
                            cur[1] += body.strip()             # location continues
                    i += 1
                if cur:
                    features.append(cur)
                continue
            elif ln.startswith("ORIGIN"):
                i += 1
                while i < n and not lines[i].startswith("//"):
                    seq.append("".join(c for c in lines[i] if c.isalpha()))
                    i += 1
                continue
            i += 1
        i += 1  # past "//"
        feats = []
        for key, loc, quals in features:
            try:
                parts, strand = parse_location(loc)
            except ValueError:
                continue
            q = {}
            for text in quals:
                k, _, v = text[1:].partition("=")
                v = v.strip()
                if len(v) >= 2 and v[0] == '"' and v[-1] == '"':
                    v = v[1:-1].replace('""', '"')
                q.setdefault(k, []).append(v)
            feats.append(GbFeature(key, parts, strand, q))
        records.append(GbRecord(rid or accession or name, name, "".join(seq).upper(), topology, organism, feats))
    return records


def write_genbank(path, records):
    """Write GbRecord-like objects (id, seq, topology, organism, features with key/parts/strand/qualifiers)
    as a GenBank flat file — enough for this module and for Biopython-style readers; used by the tests
    to put the shipped plasmids' sequences and SnapGene feature tables into the CLI's input format."""
    def loc_text(parts, strand):
        body = [f"{a + 1}..{b}" for a, b in (parts[::-1] if strand == -1 else parts)]
        s = body[0] if len(body) == 1 else "join(" + ",".join(body) + ")"
        return f"complement({s})" if strand == -1 else s
    with open(path, "w") as f:
        for r in records:
            acc = r.id.split(".")[0]
            f.write(f"LOCUS       {acc:<16} {len(r.seq):>11} bp    DNA     {r.topology or 'linear':<8} BCT 01-JAN-2000\n")
            f.write(f"DEFINITION  {r.id}.\nACCESSION   {acc}\nVERSION     {r.id}\n")
            f.write(f"SOURCE      {r.organism or '.'}\n  ORGANISM  {r.organism or '.'}\n            .\n")
            f.write("FEATURES             Location/Qualifiers\n")
            f.write(f"     source          1..{len(r.seq)}\n")
            for ft in r.features:
                f.write(f"     {ft.key:<16}{loc_text(ft.parts, ft.strand)}\n")
                for k, vals in ft.qualifiers.items():
                    for v in vals:
                        f.write(f'                     /{k}="{v}"\n')
            f.write("ORIGIN\n")
            s = r.seq.lower()
            for a in range(0, len(s), 60):
                chunk = s[a:a + 60]
                f.write(f"{a + 1:>9} " + " ".join(chunk[b:b + 10] for b in range(0, len(chunk), 10)) + "\n")
            f.write("//\n")
