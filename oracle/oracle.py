"""CPU oracle for the barcoder hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``barcoder_amd`` never does.

Two layers:

* ``liboracle.so`` (``oracle.c``): string-based C restatement — the brute-force
  sweep that defines the results (sizes up to ~1e5 reads x 1e5 guides) and a
  pigeonhole-indexed search with the same outputs (tests hold the two against
  each other) for read-by-read checks at 1e8 reads and as the timed CPU baseline.
* pure-Python restatements of the reference's per-read / per-position
  semantics for small cases, each citing the reference lines it follows.

Parity status is stated in ``oracle.c``'s header and in DESIGN.md: d = 0 and
the PAM scans are pinned by golden vectors produced by the reference itself
(``oracle/gen_golden.py``); best hit for d > 0 is PARITY UNPINNED (Bowtie 1.3.1
is absent and the reference holds no expected output for it).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from collections import Counter

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile liboracle.so (gcc) if missing or stale; returns its path."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_best_hit.restype = None
        L.orc_best_hit_indexed.restype = None
        L.orc_all_hits.restype = C.c_uint64
        L.orc_process_chunk_single.restype = None
        L.orc_pam_scan.restype = None
        L.orc_rev_comp.restype = None
        L.orc_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def threads() -> int:
    return int(lib().orc_threads())


# --------------------------------------------------------------------------- strings
def rev_comp(s: str) -> str:
    """heuristicount.py:29-30 — reverse, then A<->T, C<->G, N->N, rest unchanged."""
    comp = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}
    return "".join(comp.get(c, c) for c in reversed(s))


def kmers_to_array(seqs, k: int) -> np.ndarray:
    """n strings of length k -> (n, k) uint8 array of their ASCII bytes."""
    b = "".join(seqs).encode("ascii")
    a = np.frombuffer(b, dtype=np.uint8)
    assert a.size == len(seqs) * k, "all sequences must have length k"
    return a.reshape(len(seqs), k).copy()


# --------------------------------------------------------------------------- best hit
def best_hit(lib_seqs, read_seqs, max_mm: int, skip=None, base: int = 0):
    """Brute-force best hit on strings (oracle.c orc_best_hit).

    Returns dict(best_idx int32[n], best_dist uint8[n], n_best uint8[n],
    counts uint64[G], first_seen uint64[G]).
    """
    G, n = len(lib_seqs), len(read_seqs)
    k = len(lib_seqs[0])
    la = kmers_to_array(lib_seqs, k)
    ra = kmers_to_array(read_seqs, k) if n else np.zeros((0, k), np.uint8)
    bi = np.empty(n, np.int32)
    bd = np.empty(n, np.uint8)
    nb = np.empty(n, np.uint8)
    counts = np.zeros(G, np.uint64)
    fs = np.full(G, np.iinfo(np.uint64).max, np.uint64)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    lib().orc_best_hit(_p(la, C.c_char), C.c_uint32(G), C.c_int(k), _p(ra, C.c_char), _p(sk, C.c_uint8),
                       C.c_uint64(n), C.c_int(max_mm), _p(bi, C.c_int32), _p(bd, C.c_uint8),
                       _p(nb, C.c_uint8), _p(counts, C.c_uint64), _p(fs, C.c_uint64), C.c_uint64(base))
    return dict(best_idx=bi, best_dist=bd, n_best=nb, counts=counts, first_seen=fs)


def best_hit_arrays(la, ra, max_mm: int, skip=None, base: int = 0, indexed: bool = True):
    """Best hit on ASCII byte arrays ((G, k) and (n, k) uint8): the brute-force sweep (indexed=False, oracle.c
    orc_best_hit) or the pigeonhole-indexed search with the same definition (orc_best_hit_indexed, ~1000x faster;
    tests/test_oracle_golden.py holds the two against each other)."""
    la = np.ascontiguousarray(la, np.uint8)
    ra = np.ascontiguousarray(ra, np.uint8)
    G, k = la.shape
    n = ra.shape[0]
    bi = np.empty(n, np.int32)
    bd = np.empty(n, np.uint8)
    nb = np.empty(n, np.uint8)
    counts = np.zeros(G, np.uint64)
    fs = np.full(G, np.iinfo(np.uint64).max, np.uint64)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    fn = lib().orc_best_hit_indexed if indexed else lib().orc_best_hit
    fn(_p(la, C.c_char), C.c_uint32(G), C.c_int(k), _p(ra, C.c_char), _p(sk, C.c_uint8), C.c_uint64(n),
       C.c_int(max_mm), _p(bi, C.c_int32), _p(bd, C.c_uint8), _p(nb, C.c_uint8), _p(counts, C.c_uint64),
       _p(fs, C.c_uint64), C.c_uint64(base))
    return dict(best_idx=bi, best_dist=bd, n_best=nb, counts=counts, first_seen=fs)


def best_hit_py(lib_seqs, read_seqs, max_mm: int):
    """Same definition, pure Python (tiny cases; checks oracle.c itself)."""
    out = []
    for r in read_seqs:
        best = (None, 255, 0)
        for g, s in enumerate(lib_seqs):
            d = sum(a != b for a, b in zip(r, s))
            if d > max_mm:
                continue
            if d < best[1]:
                best = (g, d, 1)
            elif d == best[1]:
                best = (best[0], d, best[2] + 1)
        out.append((-1 if best[0] is None else best[0], best[1], min(best[2], 255)))
    return out


def all_hits(lib_seqs, read_seqs, max_mm: int, skip=None):
    """Every (query, guide, dist) within max_mm, sorted (oracle.c orc_all_hits)."""
    G, n = len(lib_seqs), len(read_seqs)
    k = len(lib_seqs[0])
    la = kmers_to_array(lib_seqs, k)
    ra = kmers_to_array(read_seqs, k)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    cap = 1 << 16
    while True:
        tr = np.empty((cap, 3), np.uint32)
        m = lib().orc_all_hits(_p(la, C.c_char), C.c_uint32(G), C.c_int(k), _p(ra, C.c_char),
                               _p(sk, C.c_uint8), C.c_uint64(n), C.c_int(max_mm), _p(tr, C.c_uint32),
                               C.c_uint64(cap))
        if m <= cap:
            return tr[:m]
        cap = int(m)


# --------------------------------------------------------------------------- process_chunk
def process_chunk_py(chunk, barcodes, L_fwd_start, L_rev_start, bc_len, L_fwd, R_fwd, L_rev, R_rev,
                     need_swap):
    """Restatement of heuristicount.py:428-562 (process_chunk) in one pass.

    ``barcodes`` is the UNflanked library (any iterable of str); the flanked
    sets of :690-697 are rebuilt here.  Returns (Counter, n_reads) exactly like
    the reference: documented barcodes keyed by sequence, undocumented ones by
    sequence + "*".
    """
    L_fwd, R_fwd, L_rev, R_rev = (x or "" for x in (L_fwd, R_fwd, L_rev, R_rev))
    fwd_set = {L_fwd + b + R_fwd for b in barcodes}            # :696
    rev_set = {L_rev + rev_comp(b) + R_rev for b in barcodes}  # :583, :697
    a, b = chunk
    r1, r2 = (b, a) if need_swap else (a, b)                   # :444-447
    counts = Counter()

    def window(read, start, L, R, lib_set):
        w = read[start:start + len(L) + bc_len + len(R)]       # :527 / :492-497
        in_lib = w in lib_set                                  # :450-453
        seq = w[len(L):len(w) - len(R)]                        # :455-457
        has = w.startswith(L) and w.endswith(R)                # :458-460
        return in_lib, has, seq

    if r1 and r2:                                              # :540 paired
        if len(r1) != len(r2):
            raise ValueError("Length of reads1 and reads2 must be the same for paired-end data.")  # :483-486
        for x, y in zip(r1, r2):
            if "N" in x or "N" in y:                           # :489
                continue
            in1, has1, s1 = window(x, L_fwd_start, L_fwd, R_fwd, fwd_set)
            in2, has2, s2 = window(y, L_rev_start, L_rev, R_rev, rev_set)
            if s1 != rev_comp(s2):                             # :506
                continue
            if in1 and in2 and has1 and has2:                  # :509-515
                counts[s1] += 1
            elif has1 and has2:                                # :517-518
                counts[s1 + "*"] += 1
    elif r1:                                                   # :555 forward single-end
        for x in r1:
            if "N" in x:                                       # :524
                continue
            in1, has1, s1 = window(x, L_fwd_start, L_fwd, R_fwd, fwd_set)
            if in1 and has1:
                counts[s1] += 1
            elif has1:
                counts[s1 + "*"] += 1
    elif r2:                                                   # :557 reverse single-end
        for y in r2:
            if "N" in y:
                continue
            in2, has2, s2 = window(y, L_rev_start, L_rev, R_rev, rev_set)
            s2 = rev_comp(s2)                                  # :532-533
            if in2 and has2:
                counts[s2] += 1
            elif has2:
                counts[s2 + "*"] += 1
    return counts, (len(r1) if r1 else len(r2))                # :562


def process_chunk_single_c(reads, barcodes, start, L, R, reverse=False):
    """oracle.c orc_process_chunk_single on a list of reads; returns Counter."""
    barcodes = list(barcodes)
    k = len(barcodes[0])
    L, R = L or "", R or ""
    text = "".join(reads).encode("ascii")
    lens = np.fromiter((len(r) for r in reads), np.uint64, len(reads))
    off = np.zeros(len(reads) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    la = kmers_to_array(barcodes, k)
    out = np.empty(len(reads), np.int32)
    so = np.empty(len(reads), np.int64)
    sl = np.empty(len(reads), np.int64)
    lib().orc_process_chunk_single(C.c_char_p(text), _p(off, C.c_uint64), C.c_uint64(len(reads)),
                                   _p(la, C.c_char), C.c_uint32(len(barcodes)), C.c_int(k),
                                   C.c_int64(start), C.c_char_p(L.encode()), C.c_int(len(L)),
                                   C.c_char_p(R.encode()), C.c_int(len(R)), C.c_int(1 if reverse else 0),
                                   _p(out, C.c_int32), _p(so, C.c_int64), _p(sl, C.c_int64))
    counts = Counter()
    for i in np.nonzero(out >= -1)[0]:
        seq = text[so[i]:so[i] + sl[i]].decode()
        if reverse:
            seq = rev_comp(seq)
        counts[seq if out[i] >= 0 else seq + "*"] += 1
    return counts, out


# --------------------------------------------------------------------------- PAM scans
def pam_bitmaps(seq: str, pam: str):
    """Per-position PAM / revcomp(PAM) match flags on the forward strand (oracle.c)."""
    s = seq.encode("ascii")
    fwd = np.zeros(len(s), np.uint8)
    rev = np.zeros(len(s), np.uint8)
    lib().orc_pam_scan(C.c_char_p(s), C.c_uint64(len(s)), C.c_char_p(pam.encode()), C.c_int(len(pam)),
                       _p(fwd, C.c_uint8), _p(rev, C.c_uint8))
    return fwd, rev


def guidefinder_py(seqs, pam: str, direction: str, length: int):
    """Restatement of PAMProcessor.py:34-57 (GuideFinder.find_guides_from_pam):
    non-overlapping left-to-right PAM matches on each strand, guide clipped at
    the sequence ends, duplicates kept, strand order forward then reverse."""
    rx = re.compile(pam.replace("N", "[ATCG]"))
    out = []
    for seq in seqs:
        for s in (seq, rev_comp(seq)):
            pos = 0
            while True:
                m = rx.search(s, pos)
                if not m:
                    break
                if direction == "downstream":
                    out.append(s[max(0, m.start() - length):m.start()])
                elif direction == "upstream":
                    out.append(s[m.end():min(m.end() + length, len(s))])
                else:
                    raise ValueError("Direction must be 'upstream' or 'downstream'")
                pos = m.end() if m.end() > m.start() else m.end() + 1
    return out


def sliding_sites_py(seq: str, barcode_length: int, pam: str, direction: str):
    """Restatement of design_guides.py:22-49 for ONE (already topological)
    sequence: every start i on both strands, PAM tested next to the spacer,
    spacer must be pure ACGT; returns the set of spacers."""
    rx = re.compile(pam.replace("N", "[ATGC]"))
    found = set()
    pl = len(pam)
    for s in (seq, rev_comp(seq)):
        for i in range(len(s) - barcode_length - pl + 1):
            if direction == "downstream":
                hit = rx.match(s[i + barcode_length:i + barcode_length + pl])
            elif direction == "upstream":
                hit = rx.match(s[i - pl:i])  # negative start for i < pl: empty slice (:42-44 quirk)
            else:
                hit = None
            if hit:
                sp = s[i:i + barcode_length]
                if all(c in "GATC" for c in sp):
                    found.add(sp)
    return found


# --------------------------------------------------------------------------- hit records
def get_diff(spacer: str, target: str):
    """targets.py:184-190."""
    d = [f"{t}{i + 1}{s}" for i, (t, s) in enumerate(zip(target, spacer)) if t != s]
    return ",".join(d) if d else None


def get_coords(tar_start: int, tar_end: int, chrom_length: int) -> str:
    """targets.py:193-202."""
    a = tar_start % chrom_length
    b = tar_end % chrom_length if tar_end % chrom_length != 0 else chrom_length
    return f"({a}..{chrom_length}, 0..{b})" if a > b else f"{a}..{b}"


def fold_coords(ref_start: int, ref_end: int, true_len: int):
    """targets.py:380-384."""
    s, e = ref_start % true_len, ref_end % true_len
    if e < s:
        s -= true_len
    return s, e


def downstream_pam(topo: str, start: int, end: int, pam_len: int, direction: str):
    """targets.py:227-266 extract_downstream_pam on a topological string."""
    if direction == "F":
        if end + pam_len > len(topo):
            return None
        return topo[end:end + pam_len].upper()
    if start - pam_len < 0:
        return None
    return rev_comp(topo[start - pam_len:start].upper())


def upstream_pam(topo: str, start: int, end: int, pam_len: int, direction: str):
    """targets.py:269-307 extract_upstream_pam."""
    if direction == "F":
        if start - pam_len < 0:
            return None
        return topo[start - pam_len:start].upper()
    if end + pam_len > len(topo):
        return None
    return rev_comp(topo[end:end + pam_len].upper())


def pam_matches(pam_pattern: str, extracted: str) -> bool:
    """targets.py:219-224."""
    if not extracted:
        return False
    if not pam_pattern or pam_pattern == "N" * len(pam_pattern):
        return True
    return bool(re.match(pam_pattern.replace("N", "."), extracted))


# --------------------------------------------------------------------------- mismatch model
def y_pred(original: str, variant: str, params: dict):
    """mismatch.py:15-35 calculate_y_pred (float64)."""
    if original is None or variant is None or original == variant or len(original) != len(variant):
        return None
    y = params["intercept"]
    for pos, (o, v) in enumerate(zip(original, variant)):
        if o != v:
            y += params[f"{pos}"]
            y += params[f"{o}{v}"]
    gc = (original.count("G") + original.count("C")) / len(original)
    return y + params["GC_content"] * gc
