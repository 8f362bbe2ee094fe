"""Sequence <-> packed "plane form" words (host side of K1; include/barcoder.h)."""
from __future__ import annotations

import numpy as np

from . import _native as N


def kmer_mask(k: int) -> np.uint64:
    m = 0xFFFFFFFF if k >= 32 else (1 << k) - 1
    return np.uint64(m | (m << 32))


def pack_kmers(seqs, k: int | None = None):
    """Equal-length strings -> (uint64[n] packed, uint8[n] flags); flag 1 = not pure ACGT."""
    seqs = list(seqs)
    n = len(seqs)
    if n == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    if k is None:
        k = len(seqs[0])
    if any(len(s) != k for s in seqs):
        raise ValueError("All barcodes must be the same length")  # heuristicount.py:594
    buf = "".join(seqs).encode("ascii", errors="replace")
    out = np.empty(n, np.uint64)
    flags = np.empty(n, np.uint8)
    N.check(N.load().bc_pack_kmers(buf, n, k, k, N.ptr(out), N.ptr(flags)))
    return out, flags


def unpack_kmers(packed, k: int):
    packed = np.ascontiguousarray(packed, np.uint64)
    n = packed.size
    if n == 0:
        return []
    out = np.empty(n * k, np.uint8)
    N.check(N.load().bc_unpack_kmers(N.ptr(packed), n, k, N.ptr(out)))
    s = out.tobytes().decode("ascii")
    return [s[i * k:(i + 1) * k] for i in range(n)]


def revcomp_kmers(packed, k: int):
    packed = np.ascontiguousarray(packed, np.uint64)
    out = np.empty_like(packed)
    N.check(N.load().bc_revcomp_kmers(N.ptr(packed), packed.size, k, N.ptr(out)))
    return out


def reads_to_text(reads):
    """list[str] -> (bytes text, uint64[n+1] offsets) for bc_pack_windows."""
    n = len(reads)
    text = "".join(reads).encode("ascii", errors="replace")
    off = np.zeros(n + 1, np.uint64)
    if n:
        np.cumsum(np.fromiter((len(r) for r in reads), np.uint64, n), out=off[1:])
    return text, off


def pack_windows(text: bytes, off, start: int, L: str, R: str, k: int, reverse: bool = False,
                 n_threads: int = 0):
    """Front half of process_chunk (heuristicount.py:449-462, :520-533) -> (packed, status)."""
    n = off.size - 1
    out = np.empty(n, np.uint64)
    status = np.empty(n, np.uint8)
    Lb, Rb = (L or "").encode("ascii"), (R or "").encode("ascii")
    if isinstance(text, np.ndarray):
        text = N.ptr(np.ascontiguousarray(text)) if text.size else b""
    off = np.ascontiguousarray(off, np.uint64)
    N.check(N.load().bc_pack_windows(text, N.ptr(off), n, int(start), Lb, len(Lb), Rb, len(Rb), k,
                                     1 if reverse else 0, n_threads, N.ptr(out), N.ptr(status)))
    return out, status


def hash_kmers(seqs, k: int | None = None) -> np.ndarray:
    """Equal-length strings of any length -> uint64[n]: the 64-bit text hash that stands for a barcode longer than
    32 nt in the GPU's exact table (bc_hash_kmers)."""
    seqs = list(seqs)
    n = len(seqs)
    if n == 0:
        return np.zeros(0, np.uint64)
    if k is None:
        k = len(seqs[0])
    if any(len(s) != k for s in seqs):
        raise ValueError("All barcodes must be the same length")  # heuristicount.py:594
    buf = "".join(seqs).encode("ascii", errors="replace")
    out = np.empty(n, np.uint64)
    N.check(N.load().bc_hash_kmers(buf, n, k, N.ptr(out)))
    return out


def hash_windows(text: bytes, off, start: int, L: str, R: str, k: int, reverse: bool = False, n_threads: int = 0):
    """pack_windows for barcodes of 33 ... 1000 nt -> (hash uint64[n], status uint8[n], windows uint8[n, k]): the
    hash and the text of the barcode as the library spells it (bc_hash_windows); both only where status is WIN_OK."""
    n = off.size - 1
    out = np.zeros(n, np.uint64)
    status = np.empty(n, np.uint8)
    wins = np.zeros((n, k), np.uint8)
    Lb, Rb = (L or "").encode("ascii"), (R or "").encode("ascii")
    if isinstance(text, np.ndarray):
        text = N.ptr(np.ascontiguousarray(text)) if text.size else b""
    off = np.ascontiguousarray(off, np.uint64)
    N.check(N.load().bc_hash_windows(text, N.ptr(off), n, int(start), Lb, len(Lb), Rb, len(Rb), k,
                                     1 if reverse else 0, n_threads, N.ptr(out), N.ptr(status), N.ptr(wins)))
    return out, status, wins


IUPAC = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
         "B": 14, "D": 13, "H": 11, "V": 7}


PAM_ANY_CHAR = 16  # BC_PAM_ANY_CHAR


def pam_masks(pam: str, iupac: bool = False, n_is_dot: bool = False) -> np.ndarray:
    """PAM string -> 4-bit base sets.  Like the reference only N is a wildcard
    (PAMProcessor.py:31, design_guides.py:26, targets.py:224); pass iupac=True
    to read the other IUPAC codes as sets (an extension).  n_is_dot=True gives N
    the meaning targets.py:224 gives it ('.': any character, also a genome base
    that is not ACGT) instead of PAMProcessor's / design_guides' [ATCG]."""
    out = np.zeros(len(pam), np.uint8)
    for i, c in enumerate(pam):
        if c in "ACGTN" or (iupac and c in IUPAC):
            out[i] = IUPAC[c] | (PAM_ANY_CHAR if (n_is_dot and c == "N") else 0)
        else:
            raise ValueError(f"PAM character {c!r} can never match a genome base (only ACGT and N are supported)")
    return out
