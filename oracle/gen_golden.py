#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE in this container.

Run from the repo root:  python oracle/gen_golden.py
Needs /root/reference (read-only); never runs on the GPU box.  The reference's
modules are imported unmodified.  Two of their imports are absent offline and
are not on the measured path: `zstandard` (only used to open .zst files,
heuristicount.py:46,64) and `babel` (log-text number formatting, Logger.py:1);
empty stand-in modules are registered for those two names so the import
succeeds.  Nothing from the reference is copied: only inputs and the outputs it
computed are written (JSON / TSV / FASTA data).

Outputs
  heuristicount_chunks.json.gz process_chunk I/O pairs (inputs + Counter, insertion order)
  heuristicount_main.json.gz  main()/sample_data/find_flanks on small FASTQ scenarios
  mismatch.json               mismatch.py y_pred values and generated variant tables
  plasmids.fasta              the 4 shipped SnapGene plasmids as FASTA (reference DATA)
  cn32_plasmid_rows.tsv       rows of Example_Libraries/CN-32-zmo.tsv on those plasmids
  cn32_spacers.txt.gz         the `spacer` column of CN-32-zmo.tsv (library of config 1)
  pam_guidefinder.json.gz     GuideFinder.find_guides_from_pam outputs on the plasmids
  plasmid_genes.tsv           gene features of the 4 plasmids (SnapGene feature table: reference DATA)
  crispri.json.gz             CRISPRiLibrary (CRISPRiLibrary.py:4-120) + PAMFinder (PAMProcessor.py:60-97) run on
                              the hit x feature frame of the shipped CN-32 spacers on the plasmids: the frame
                              (input) and every view the class derives from it (outputs), for three PAM settings.
                              `Bio.Seq` (an import the class never uses, CRISPRiLibrary.py:1) gets a stand-in module.
"""
import contextlib
import gzip
import io
import json
import os
import random
import struct
import sys
import tempfile
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def import_reference():
    for name in ("zstandard",):
        sys.modules.setdefault(name, types.ModuleType(name))
    b, bn = types.ModuleType("babel"), types.ModuleType("babel.numbers")
    bn.format_decimal = lambda x, **k: str(x)
    b.numbers = bn
    sys.modules.setdefault("babel", b)
    sys.modules.setdefault("babel.numbers", bn)
    # CRISPRiLibrary.py:1 imports Bio.Seq.Seq and never touches it
    bio, bioseq = types.ModuleType("Bio"), types.ModuleType("Bio.Seq")
    bioseq.Seq = type("Seq", (), {})
    bio.Seq = bioseq
    sys.modules.setdefault("Bio", bio)
    sys.modules.setdefault("Bio.Seq", bioseq)
    sys.path.insert(0, REF)
    import heuristicount
    import mismatch
    import PAMProcessor
    return heuristicount, mismatch, PAMProcessor


def gz_write(name, text):
    """gzip with a zeroed header timestamp: regenerating the fixtures reproduces them byte for byte"""
    with open(os.path.join(OUT, name), "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
        f.write(text.encode())


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, nsub):
    s = list(s)
    for p in rng.sample(range(len(s)), nsub):
        s[p] = rng.choice([c for c in "ACGT" if c != s[p]])
    return "".join(s)


def rc(s):
    return s[::-1].translate(str.maketrans("ATCGN", "TAGCN"))


# ----------------------------------------------------------------------------- chunks
def chunk_scenarios(h):
    rng = random.Random(20240601)
    out = []

    def run(name, chunk, barcodes, L_fwd_start, L_rev_start, bc_len, L_fwd, R_fwd, L_rev, R_rev, need_swap):
        fwd = {(L_fwd or "") + b + (R_fwd or "") for b in barcodes}
        rev = {(L_rev or "") + rc(b) + (R_rev or "") for b in barcodes}
        try:
            counts, n = h.process_chunk(chunk, fwd, rev, L_fwd_start, L_rev_start, bc_len, L_fwd, R_fwd, L_rev,
                                        R_rev, need_swap)
            result = {"counts": list(counts.items()), "n": n}
        except ValueError as e:
            result = {"error": str(e)}
        out.append({"name": name, "chunk": [chunk[0], chunk[1]], "barcodes": barcodes,
                    "L_fwd_start": L_fwd_start, "L_rev_start": L_rev_start, "bc_len": bc_len, "L_fwd": L_fwd,
                    "R_fwd": R_fwd, "L_rev": L_rev, "R_rev": R_rev, "need_swap": need_swap, **result})

    for k in (20, 32, 12):
        barcodes = list(dict.fromkeys(rand_seq(rng, k) for _ in range(120)))
        Lf, Rf = "CCGGATTCAG", "GTTTTAGAGC"
        pre = 5

        def make_read(kind):
            bc = rng.choice(barcodes)
            if kind == "mm1":
                bc = mutate(rng, bc, 1)
            elif kind == "mm2":
                bc = mutate(rng, bc, 2)
            elif kind == "rand":
                bc = rand_seq(rng, k)
            r = rand_seq(rng, pre) + Lf + bc + Rf + rand_seq(rng, 6)
            if kind == "N":
                p = rng.randrange(len(r))
                r = r[:p] + "N" + r[p + 1:]
            elif kind == "badL":
                r = rand_seq(rng, pre) + mutate(rng, Lf, 1) + bc + Rf + rand_seq(rng, 6)
            elif kind == "badR":
                r = rand_seq(rng, pre) + Lf + bc + mutate(rng, Rf, 2) + rand_seq(rng, 6)
            elif kind == "short":
                r = r[:pre + len(Lf) + rng.randrange(0, k + len(Rf))]
            elif kind == "lower":
                p = pre + len(Lf) + rng.randrange(k)
                r = r[:p] + r[p].lower() + r[p + 1:]
            elif kind == "odd":
                p = pre + len(Lf) + rng.randrange(k)
                r = r[:p] + "R" + r[p + 1:]
            return r

        kinds = ["ok"] * 60 + ["mm1"] * 10 + ["mm2"] * 5 + ["rand"] * 5 + ["N"] * 4 + ["badL"] * 4 + \
                ["badR"] * 4 + ["short"] * 4 + ["lower"] * 2 + ["odd"] * 2
        reads = [make_read(rng.choice(kinds)) for _ in range(500)] + [""]
        run(f"fwd_flanks_k{k}", [reads, None], barcodes, pre, None, k, Lf, Rf, None, None, False)
        # no flanks at all: every window of the right length counts (documented or '*')
        run(f"fwd_noflanks_k{k}", [reads, None], barcodes, pre + len(Lf), None, k, None, None, None, None, False)
        # only a left / only a right flank
        run(f"fwd_Lonly_k{k}", [reads, None], barcodes, pre + 4, None, k, Lf[4:], None, None, None, False)
        run(f"fwd_Ronly_k{k}", [reads, None], barcodes, pre + len(Lf), None, k, None, Rf[:5], None, None, False)
        # reverse-orientation single-end: the chunk's reads arrive as reads2 via need_swap
        rreads = [rc(r) if "N" not in r else r for r in reads]
        Lr, Rr = rc(Rf), rc(Lf)
        run(f"rev_single_k{k}", [rreads, None], barcodes, None, 6, k, None, None, Lr, Rr, True)
        # paired-end, mates agree except for a few corrupted ones
        mates = []
        for r in reads:
            m = rc(r) if "N" not in r else r
            if rng.random() < 0.05 and len(m) > 30:
                m = mutate(rng, m, 1)
            mates.append(m)
        run(f"paired_k{k}", [reads, mates], barcodes, pre, 6, k, Lf, Rf, Lr, Rr, False)
        run(f"paired_swapped_k{k}", [mates, reads], barcodes, pre, 6, k, Lf, Rf, Lr, Rr, True)
    # mate-count mismatch -> ValueError (heuristicount.py:483-486)
    run("paired_len_mismatch", [["ACGT" * 10, "ACGT" * 10], ["ACGT" * 10]], ["ACGTACGTAC"] * 1, 0, 0, 10, None, None,
        None, None, False)
    return out


# ----------------------------------------------------------------------------- main()
def main_scenarios(h):
    rng = random.Random(77)
    out = []
    tmp = tempfile.mkdtemp(prefix="bc_golden_")

    def scenario(name, k, n_bc, n_reads, paired, reverse_single=False, ext="fastq", with_flanks=True, pre=5):
        barcodes = list(dict.fromkeys(rand_seq(rng, k) for _ in range(n_bc)))
        Lf, Rf = ("CCGGATTCAG", "GTTTTAGAGCTAGAAATAGC") if with_flanks else ("", "")
        reads1, reads2 = [], []
        for _ in range(n_reads):
            u = rng.random()
            bc = rng.choice(barcodes)
            if u > 0.95:
                bc = mutate(rng, bc, 2)
            elif u > 0.85:
                bc = mutate(rng, bc, 1)
            if with_flanks:
                r = rand_seq(rng, pre) + Lf + bc + Rf
            else:
                r = rand_seq(rng, pre) + bc + rand_seq(rng, 12)
            if rng.random() < 0.01:
                p = rng.randrange(len(r))
                r = r[:p] + "N" + r[p + 1:]
            reads1.append(r)
            reads2.append(rc(r))
        fa = os.path.join(tmp, f"{name}.fasta")
        with open(fa, "w") as f:
            for i, b in enumerate(barcodes):
                f.write(f">bc{i}\n{b}\n")

        def write_reads(path, reads):
            with open(path, "w") as f:
                for i, r in enumerate(reads):
                    if ext == "fastq":
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                    else:
                        f.write(r + "\n")

        f1 = os.path.join(tmp, f"{name}_1.{ext}")
        f2 = os.path.join(tmp, f"{name}_2.{ext}") if paired else None
        write_reads(f1, reads2 if reverse_single else reads1)
        if paired:
            write_reads(f2, reads2)
        lib = h.read_fasta(fa)
        sd = h.sample_data(f1, f2, lib, paired)
        (n_novel, off1, off2, s1, s2, observed, need_swap, n_chunks) = sd
        fl1 = h.find_flanks(s1, off1, k) if s1 is not None and off1 is not None else None
        fl2 = h.find_flanks(s2, off2, k) if s2 is not None and off2 is not None and s2 else None
        args = types.SimpleNamespace(fasta_file=fa, file1=f1, file2=f2)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            h.main(args)
        out.append({
            "name": name, "k": k, "barcodes": barcodes, "ext": ext, "paired": paired,
            "reads1": reads2 if reverse_single else reads1, "reads2": reads2 if paired else None,
            "sample_data": {"n_novel": n_novel, "offset1": off1, "offset2": off2, "n_valid1": len(s1 or ()),
                            "n_valid2": len(s2 or ()), "n_observed": len(observed), "need_swap": need_swap,
                            "num_chunks": n_chunks},
            "flanks1": list(fl1) if fl1 else None, "flanks2": list(fl2) if fl2 else None,
            "stdout": buf.getvalue(),
        })

    scenario("fwd_single", 20, 200, 3000, paired=False)
    scenario("rev_single", 20, 200, 3000, paired=False, reverse_single=True)
    scenario("paired", 20, 150, 2500, paired=True)
    scenario("reads_format_k32", 32, 120, 2000, paired=False, ext="reads")
    scenario("noflank", 20, 100, 2000, paired=False, with_flanks=False, pre=7)
    return out


# ----------------------------------------------------------------------------- mismatch.py
def mismatch_cases(mm):
    rng = random.Random(5)
    # the weights are plain CSV data (feature,weight,...); mismatch.read_parameters (mismatch.py:38-46)
    # builds the same {feature: weight} dict through pandas
    import csv
    with open(os.path.join(REF, "mismatch_parameters.csv")) as f:
        params = {row["feature"]: float(row["weight"]) for row in csv.DictReader(f)}
    pairs = []
    for _ in range(40):
        o = rand_seq(rng, 20)
        v = mutate(rng, o, rng.choice([1, 1, 2, 3]))
        pairs.append((o, v, mm.calculate_y_pred(o, v, params["GC_content"], params)))
    pairs.append(("ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGT", None))
    pairs.append(("ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGA",
                  mm.calculate_y_pred("ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGA", params["GC_content"], params)))
    tables = []
    for _ in range(6):
        sp = rand_seq(rng, 20)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mm.generate_mismatches([sp], 0.0, 1.0, 0.1, params, sp)
        tables.append({"spacer": sp, "tsv": buf.getvalue()})
    return {"params": params, "pairs": pairs, "tables": tables}


# ----------------------------------------------------------------------------- plasmids + PAM
def read_snapgene(path):
    """SnapGene .dna: packets `type(1B) len(4B BE) data`; packet 0 = flags(1B) + sequence."""
    with open(path, "rb") as f:
        data = f.read()
    at = 0
    while at < len(data):
        t = data[at]
        (ln,) = struct.unpack(">I", data[at + 1:at + 5])
        body = data[at + 5:at + 5 + ln]
        if t == 0:
            return body[1:].decode("ascii").upper(), bool(body[0] & 1)
        at += 5 + ln
    raise ValueError("no sequence packet")


def read_snapgene_genes(path):
    """Gene features of a SnapGene file (packet 10 = XML feature table): 0-based half-open
    interval, strand (+1 / -1 from directionality 1 / 2), locus_tag and gene qualifiers — the
    fields targets.create_locus_map (targets.py:76-165) takes from the GenBank record."""
    import xml.etree.ElementTree as ET
    with open(path, "rb") as f:
        data = f.read()
    at, xml = 0, None
    while at < len(data):
        t = data[at]
        (ln,) = struct.unpack(">I", data[at + 1:at + 5])
        if t == 10:
            xml = data[at + 5:at + 5 + ln].decode("utf8")
            break
        at += 5 + ln
    genes = []
    for f in ET.fromstring(xml):
        if f.get("type") != "gene":
            continue
        q = {qq.get("name"): (qq.find("V").get("text") or qq.find("V").get("predef")) for qq in f.findall("Q")}
        parts = []
        for seg in f.findall("Segment"):
            a, b = seg.get("range").split("-")
            parts.append((int(a) - 1, int(b)))  # a > b: the segment runs across the origin of a circular record
        strand = {"1": 1, "2": -1}.get(f.get("directionality"), 0)
        genes.append((parts, strand, q.get("locus_tag"), q.get("gene")))
    return genes


class Rec:
    """Duck-typed record for GuideFinder: .seq supports str() and .reverse_complement()."""

    class _S(str):
        def reverse_complement(self):
            return Rec._S(rc(self))

    def __init__(self, s):
        self.seq = Rec._S(s)


def pam_cases(P):
    names = ["CP023716", "CP023717", "CP023718", "CP023719"]
    seqs = {}
    for nme in names:
        s, circ = read_snapgene(os.path.join(REF, "GCA_003054575.1", nme + ".dna"))
        seqs[nme + ".1"] = (s, circ)
    with open(os.path.join(OUT, "plasmids.fasta"), "w") as f:
        for k, (s, circ) in seqs.items():
            f.write(f">{k} circular={int(circ)}\n")
            for i in range(0, len(s), 80):
                f.write(s[i:i + 80] + "\n")
    with open(os.path.join(OUT, "plasmid_genes.tsv"), "w") as f:
        f.write("chr\tparts\tstrand\tlocus_tag\tgene\n")
        for nme in names:
            for parts, strand, lt, gn in read_snapgene_genes(os.path.join(REF, "GCA_003054575.1", nme + ".dna")):
                L = len(seqs[nme + ".1"][0])
                # an origin-spanning segment becomes the two parts GenBank's join(a..L,1..b) has
                flat = [q for a, b in parts for q in ([(a, b)] if a < b else [(a, L), (0, b)])]
                f.write("\t".join([nme + ".1", ";".join(f"{a}-{b}" for a, b in flat), str(strand), lt or "None",
                                   gn or "None"]) + "\n")
    recs = {k: Rec(s) for k, (s, _) in seqs.items()}
    out = {}
    for pam, L in (("NGG", 20), ("NGNC", 32), ("TTTN", 23), ("GGGGGGG", 20)):
        for direction in ("downstream", "upstream"):
            out[f"{pam}_{L}_{direction}"] = P.GuideFinder(recs, pam, direction, L).find_guides_from_pam()
    gz_write("pam_guidefinder.json.gz", json.dumps(out))
    # rows of the shipped library that lie on these plasmids (known-answer table)
    rows, spacers = [], []
    with open(os.path.join(REF, "Example_Libraries", "CN-32-zmo.tsv")) as f:
        header = f.readline()
        cols = header.rstrip("\n").split("\t")
        ci, si = cols.index("chr"), cols.index("spacer")
        for line in f:
            parts = line.rstrip("\n").split("\t")
            spacers.append(parts[si])
            if parts[ci] in seqs:
                rows.append(line)
    with open(os.path.join(OUT, "cn32_plasmid_rows.tsv"), "w") as f:
        f.write(header)
        f.writelines(rows)
    gz_write("cn32_spacers.txt.gz", "\n".join(spacers) + "\n")
    return len(rows), len(spacers), len(set(spacers))


# ----------------------------------------------------------------------------- CRISPRiLibrary views
class SliceRec:
    """Duck-typed record for PAMFinder: .seq slices to an object with .reverse_complement() and str()."""

    class _S(str):
        def __getitem__(self, key):
            return SliceRec._S(str.__getitem__(self, key))

        def reverse_complement(self):
            return SliceRec._S(rc(self))

    def __init__(self, s):
        self.seq = SliceRec._S(s)


def crispri_cases(P):
    """testing_grounds.py:20-43 with the stages that cannot run here written out as plain data: the hit table
    (PySamParser.py:38-46 columns) = every exact placement of the shipped CN-32 spacers on either strand of the four
    linear plasmid records (what `bowtie -a -v 0` reports for them), joined with the `source` and per-part `gene`
    features (GenBankParser.py:69-103 columns, a strand-agnostic overlap join like `PyRanges.join`).  The class and
    PAMFinder are the reference's own, imported unmodified."""
    import pandas as pd
    import CRISPRiLibrary as C
    seqs = {}
    with open(os.path.join(OUT, "plasmids.fasta")) as f:
        for line in f:
            if line.startswith(">"):
                name = line[1:].split()[0]
                seqs[name] = []
            else:
                seqs[name].append(line.strip())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    feats = {k: [("source", 0, len(s), "+", None, None)] for k, s in seqs.items()}  # record.features: source first
    with open(os.path.join(OUT, "plasmid_genes.tsv")) as f:
        next(f)
        for line in f:
            c, parts, strand, lt, gn = line.rstrip("\n").split("\t")
            for part in parts.split(";"):
                a, b = (int(x) for x in part.split("-"))
                feats[c].append(("gene", a, b, "+" if strand == "1" else "-" if strand == "-1" else ".",
                                 None if lt == "None" else lt, None if gn == "None" else gn))
    spacers = []
    with open(os.path.join(OUT, "cn32_plasmid_rows.tsv")) as f:
        cols = f.readline().rstrip("\n").split("\t")
        for line in f:
            spacers.append(line.split("\t")[cols.index("spacer")])
    spacers = list(dict.fromkeys(spacers))
    hits = []
    for sp in spacers:
        for c, s in seqs.items():
            for strand, pat in (("+", sp), ("-", rc(sp))):
                at = s.find(pat)
                while at >= 0:
                    hits.append((c, at, at + len(sp), strand, sp))
                    at = s.find(pat, at + 1)
    hits.sort()
    frame = []
    for c, a, b, strand, sp in hits:
        for typ, fa, fb, fs, lt, gn in feats[c]:
            if fa < b and fb > a:
                frame.append({"Chromosome": c, "Start": a, "End": b, "Mapped": True, "Strand": strand, "Barcode": sp,
                              "Mismatches": 0, "Start_b": fa, "End_b": fb, "Strand_b": fs, "Locus_Tag": lt, "Gene": gn,
                              "Type": typ})
    recs = {k: SliceRec(s) for k, s in seqs.items()}

    def records(df):
        return json.loads(df.to_json(orient="records"))

    out = {"frame": frame, "cases": []}
    for pam, direction in (("NGNC", "downstream"), ("NGG", "downstream"), ("NGNC", "upstream")):
        lib = C.CRISPRiLibrary(pd.DataFrame(frame), P.PAMFinder(recs, pam, direction))
        out["cases"].append({
            "pam": pam, "direction": direction,
            "PAM": list(lib.targets_df["PAM"]), "Targeting": [bool(x) for x in lib.targets_df["Targeting"]],
            "source_unique_targets": records(lib.source_unique_targets),
            "mapped_targets": records(lib.mapped_targets),
            "unique_targets": records(lib.unique_targets),
            "unambiguous_targets": records(lib.unambiguous_targets),
            "unambiguous_index": [int(i) for i in lib.unambiguous_targets.index],
        })
    gz_write("crispri.json.gz", json.dumps(out))
    return len(hits), len(frame), [(len(c["mapped_targets"]), len(c["unique_targets"]), len(c["unambiguous_targets"]),
                                   sum(c["Targeting"])) for c in out["cases"]]


def main():
    os.makedirs(OUT, exist_ok=True)
    h, mm, P = import_reference()
    gz_write("heuristicount_chunks.json.gz", json.dumps(chunk_scenarios(h)))
    gz_write("heuristicount_main.json.gz", json.dumps(main_scenarios(h)))
    with open(os.path.join(OUT, "mismatch.json"), "w") as f:
        json.dump(mismatch_cases(mm), f)
    print("plasmid rows / spacers / unique:", pam_cases(P))
    print("crispri hits / frame rows / (mapped, unique, unambiguous, targeting) per case:", crispri_cases(P))


if __name__ == "__main__":
    main()
