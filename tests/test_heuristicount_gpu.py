"""GPU parity for the process_chunk drop-in and the main() driver against golden
vectors produced by the reference itself."""
import contextlib
import gzip
import io
import json
import os
import types
from collections import Counter

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _load(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        return json.load(f)


def _rc(s):
    return s[::-1].translate(str.maketrans("ATCGN", "TAGCN"))


def test_process_chunk_drop_in(gpu_device):
    """Same 11 positional arguments as heuristicount.py:428-440, same (Counter, int)
    out — key order (first-seen) included."""
    from barcoder_amd.heuristicount import process_chunk
    n = 0
    for c in _load("heuristicount_chunks.json.gz"):
        fwd = {(c["L_fwd"] or "") + b + (c["R_fwd"] or "") for b in c["barcodes"]}
        rev = {(c["L_rev"] or "") + _rc(b) + (c["R_rev"] or "") for b in c["barcodes"]}
        args = (tuple(c["chunk"]), fwd, rev, c["L_fwd_start"], c["L_rev_start"], c["bc_len"], c["L_fwd"], c["R_fwd"],
                c["L_rev"], c["R_rev"], c["need_swap"])
        if "error" in c:
            with pytest.raises(ValueError, match="must be the same"):
                process_chunk(*args)
            continue
        counts, total = process_chunk(*args)
        assert isinstance(counts, Counter) and total == c["n"], c["name"]
        assert list(counts.items()) == [tuple(x) for x in c["counts"]], c["name"]
        n += 1
    assert n >= 21


def test_main_stdout_matches_reference(gpu_device, tmp_path, capsys):
    """End to end: FASTA + FASTQ/.reads files in, `barcode<TAB>count` lines out,
    byte-identical to the reference's stdout (first-seen order)."""
    from barcoder_amd import heuristicount as H
    for s in _load("heuristicount_main.json.gz"):
        if s["name"] == "noflank":
            continue  # the reference's own answer depends on set iteration order here (spurious 1-nt flanks)
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
        args = types.SimpleNamespace(fasta_file=str(fa), file1=str(f1), file2=str(f2) if f2 else None)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            H.main(args, device=gpu_device)
        assert buf.getvalue() == s["stdout"], s["name"]
        assert len(s["stdout"].splitlines()) > 50


def test_main_swallows_errors_like_reference(gpu_device, tmp_path, caplog):
    from barcoder_amd import heuristicount as H
    fa = tmp_path / "few.fasta"
    fa.write_text(">a\nACGT\n")
    rd = tmp_path / "r.reads"
    rd.write_text("ACGT\n")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        H.main(types.SimpleNamespace(fasta_file=str(fa), file1=str(rd), file2=None))  # logs, does not raise
    assert buf.getvalue() == ""
    assert "fewer than 10" in caplog.text


def test_config1_cn32_library(gpu_device, oracle):
    """BASELINE configs[0]: the shipped CN-32-zmo library (9,503 unique 32-mers)
    vs 10k synthetic reads with the SURVEY 8d cfg1 layout, against the oracle."""
    import random
    from barcoder_amd.heuristicount import ChunkCounter
    with gzip.open(os.path.join(GOLDEN, "cn32_spacers.txt.gz"), "rt") as f:
        lib = list(dict.fromkeys(f.read().split()))
    assert len(lib) == 9503
    rng = random.Random(1234)
    Lf, Rf = "CCGGATTCAG", "GTTTTAGAGCTAGAAATAGC"
    reads = []
    for _ in range(10_000):
        bc = list(rng.choice(lib))
        u = rng.random()
        for p in rng.sample(range(32), 2 if u > 0.95 else 1 if u > 0.85 else 0):
            bc[p] = rng.choice([c for c in "ACGT" if c != bc[p]])
        r = "".join(rng.choice("ACGT") for _ in range(5)) + Lf + "".join(bc) + Rf
        if rng.random() < 0.01:
            p = rng.randrange(len(r))
            r = r[:p] + "N" + r[p + 1:]
        reads.append(r)
    cc = ChunkCounter(lib, 32, 5, None, Lf, Rf, None, None, False, device=gpu_device)
    got, n = cc.process((reads, None))
    want, n2 = oracle.process_chunk_py((reads, None), lib, 5, None, 32, Lf, Rf, None, None, False)
    assert n == n2 == 10_000
    assert list(got.items()) == list(want.items())
    assert sum(v for k, v in got.items() if not k.endswith("*")) > 8000


def test_cli_twin(gpu_device, tmp_path):
    """`python -m barcoder_amd.heuristicount lib.fasta reads.fastq.gz` prints the reference's stdout."""
    import subprocess
    import sys
    s = next(x for x in _load("heuristicount_main.json.gz") if x["name"] == "fwd_single")
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">bc{i}\n{b}\n" for i, b in enumerate(s["barcodes"])))
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wt") as f:
        for i, r in enumerate(s["reads1"]):
            f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "barcoder_amd.heuristicount", str(fa), str(fq)], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout == s["stdout"]
    assert "Total Reads" in out.stderr and "Documented Barcode Reads" in out.stderr
