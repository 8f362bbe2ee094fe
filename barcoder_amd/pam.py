"""PAM site scan on the GPU with the reference's Python call shapes.

Mirrors (reference file:line):
  PAMProcessor                          PAMProcessor.py:4-24
  GuideFinder.find_guides_from_pam      PAMProcessor.py:27-57   (flavour A: regex, non-overlapping)
  PAMFinder.get_pam_seq / pam_matches   PAMProcessor.py:60-97
  find_sequences_with_barcode_and_pam   design_guides.py:22-49  (flavour B: every position)

The per-position pattern test, the k-mer extraction and the validity test run in
bc_pam.hip; what remains here is choosing which of the device-reported sites a
given reference flavour keeps (string slicing for output labels included).
"""
from __future__ import annotations

import re
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .devmem import DevBuf, PinnedBlock
from .encoding import pam_masks, unpack_kmers
from .library import read_fasta_records

_LOWER_TO_X = bytes.maketrans(b"acgtn", b"xxxxx")


@dataclass
class PackedGenome:
    bases: np.ndarray    # uint64[ceil(total/32)] plane-form words
    invalid: np.ndarray  # uint32[...] 1 bit per position
    length: int          # true sequence length
    wrap: int            # bases repeated after the end (circular sequences)

    @property
    def total(self) -> int:
        return self.length + self.wrap


def pack_genome(seq, wrap: int = 0, case_sensitive: bool = False) -> PackedGenome:
    """ASCII sequence -> device layout.  case_sensitive=True makes lower-case
    letters non-matching, like the reference's regexes on a raw string."""
    b = seq if isinstance(seq, (bytes, bytearray)) else str(seq).encode("ascii", errors="replace")
    if case_sensitive:
        b = bytes(b).translate(_LOWER_TO_X)
    n = len(b)
    if wrap and n == 0:
        raise ValueError("cannot wrap an empty sequence")
    nw = (n + wrap + 31) // 32
    bases = np.zeros(max(nw, 1), np.uint64)
    inv = np.zeros(max(nw, 1), np.uint32)
    N.check(N.load().bc_genome_pack(bytes(b), n, wrap, N.ptr(bases), N.ptr(inv)))
    return PackedGenome(bases[:nw], inv[:nw], n, wrap)


def _bits(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


def pam_scan(seq, pam: str, device: int = 0, wrap: int = 0, iupac: bool = False):
    """(fwd, rev) boolean arrays over positions [0, len+wrap): fwd[p] = PAM
    matches seq[p:p+|pam|]; rev[p] = revcomp(PAM) matches there (a PAM on the
    reverse strand).  Runs bc_pam_scan on `device`."""
    g = seq if isinstance(seq, PackedGenome) else pack_genome(seq, wrap)
    masks = pam_masks(pam, iupac)
    if g.total == 0:
        return np.zeros(0, bool), np.zeros(0, bool)
    fwd = np.zeros(g.bases.size, np.uint32)
    rev = np.zeros(g.bases.size, np.uint32)
    N.check(N.load().bc_pam_scan_host(device, N.ptr(g.bases), N.ptr(g.invalid), g.total, N.ptr(masks), masks.size,
                                      N.ptr(fwd), N.ptr(rev)))
    return _bits(fwd, g.total), _bits(rev, g.total)


@dataclass
class Sites:
    """All PAM-adjacent k-mer sites of one sequence, both strands.
    `start` is the forward-coordinate start of the spacer's footprint,
    `strand` 0 = forward, 1 = reverse (spacer = revcomp of the footprint)."""
    start: np.ndarray    # int64
    strand: np.ndarray   # uint8
    spacer: np.ndarray   # uint64 packed, oriented 5'->3' on its own strand
    pam_pos: np.ndarray  # int64: forward-coordinate start of the PAM footprint


def _at(p, off: int):
    return N.vp((p.value or 0) + off)


def find_sites(g: PackedGenome, pam: str, k: int, direction: str = "downstream", device: int = 0,
               iupac: bool = False) -> Sites:
    """Device pipeline in two launches (bc_pam_sites_count_dev / bc_pam_sites_emit_dev): PAM test and spacer-window
    test (inside [0, total), pure ACGT) per position, then one {PAM position, oriented spacer} record per site —
    forward-strand sites in ascending order, then reverse-strand sites."""
    if direction not in ("downstream", "upstream"):
        raise ValueError("Direction must be 'upstream' or 'downstream'")  # PAMProcessor.py:53
    L = N.load()
    masks = pam_masks(pam, iupac)
    pl = masks.size
    nw = g.bases.size
    if g.total == 0:
        z = np.zeros(0, np.int64)
        return Sites(z, np.zeros(0, np.uint8), np.zeros(0, np.uint64), z)
    dirn = 0 if direction == "downstream" else 1
    nb = (nw + 1023) // 1024
    N.check(L.bc_dev_set(device))
    # one allocation in: [bases | invalid | block counts]; one out: [spacer words | positions | totals]
    o_inv, o_cnt = nw * 8, nw * 8 + ((nw * 4 + 15) // 16) * 16
    with DevBuf(o_cnt + 2 * (nb + 1) * 4, device) as d_in:
        N.check(L.bc_dev_h2d(d_in.ptr, N.ptr(np.ascontiguousarray(g.bases)), nw * 8, None))
        N.check(L.bc_dev_h2d(_at(d_in.ptr, o_inv), N.ptr(np.ascontiguousarray(g.invalid)), nw * 4, None))
        N.check(L.bc_pam_sites_count_dev(d_in.ptr, _at(d_in.ptr, o_inv), g.total, N.ptr(masks), pl, k, dirn,
                                         _at(d_in.ptr, o_cnt), None))
        cnt = np.empty(2 * nb, np.uint32)
        N.check(L.bc_dev_d2h(N.ptr(cnt), _at(d_in.ptr, o_cnt), cnt.nbytes, None))
        n_f, n_r = int(cnt[0::2].sum(dtype=np.uint64)), int(cnt[1::2].sum(dtype=np.uint64))
        n = n_f + n_r
        o_pos = n * 8
        with DevBuf(n * 12 + 32, device) as d_out:
            N.check(L.bc_pam_sites_emit_dev(d_in.ptr, _at(d_in.ptr, o_inv), g.total, N.ptr(masks), pl, k, dirn,
                                            _at(d_in.ptr, o_cnt), _at(d_out.ptr, o_pos), d_out.ptr, n,
                                            _at(d_out.ptr, ((n * 12 + 15) // 16) * 16), None))
            host = PinnedBlock(n * 12 + 16)  # page-locked: the copy runs at the link's rate; the views below keep it alive
            if n:
                N.check(L.bc_dev_d2h(host.ptr, d_out.ptr, n * 12, None))
    words = host.array(np.uint64, n)
    pos = host.array(np.uint32, n, n * 8).astype(np.int64)
    strand = np.zeros(n, np.uint8)
    strand[n_f:] = 1
    start = pos.copy()
    start[:n_f] += -k if dirn == 0 else pl
    start[n_f:] += pl if dirn == 0 else -k
    return Sites(start, strand, words, pos)


def find_sites_staged(g: PackedGenome, pam: str, k: int, direction: str = "downstream", device: int = 0,
                      iupac: bool = False) -> Sites:
    """The same result from the separate stages (pattern bitmaps -> position lists -> k-mer windows, six
    launches): the building blocks targets.find_hits uses, kept as the cross-check of the fused pipeline."""
    if direction not in ("downstream", "upstream"):
        raise ValueError("Direction must be 'upstream' or 'downstream'")  # PAMProcessor.py:53
    L = N.load()
    masks = pam_masks(pam, iupac)
    pl = masks.size
    nw = g.bases.size
    if g.total == 0:
        z = np.zeros(0, np.int64)
        return Sites(z, np.zeros(0, np.uint8), np.zeros(0, np.uint64), z)
    N.check(L.bc_dev_set(device))
    with DevBuf.from_numpy(g.bases, device) as d_b, DevBuf.from_numpy(g.invalid, device) as d_i, \
            DevBuf(nw * 4, device) as d_f, DevBuf(nw * 4, device) as d_r, \
            DevBuf(8, device) as d_n, DevBuf((nw // 1024 + 2) * 4, device) as d_tmp:
        N.check(L.bc_pam_scan_dev(d_b.ptr, d_i.ptr, g.total, N.ptr(masks), pl, d_f.ptr, d_r.ptr, None))
        out = []
        # (bitmap, strand, spacer offset relative to the PAM position)
        plan = [(d_f, 0, -k if direction == "downstream" else pl),
                (d_r, 1, pl if direction == "downstream" else -k)]
        for d_bm, strand, off in plan:
            cap = g.total
            with DevBuf(cap * 4, device) as d_pos:
                N.check(L.bc_sites_expand_dev(d_bm.ptr, nw, d_pos.ptr, cap, d_n.ptr, d_tmp.ptr, None))
                n = int(d_n.to_numpy(np.uint64, 1)[0])
                with DevBuf(max(n, 1) * 8, device) as d_w, DevBuf(max(n, 1), device) as d_fl:
                    N.check(L.bc_windows_dev(d_b.ptr, d_i.ptr, g.total, d_pos.ptr, n, off, k, strand,
                                             d_w.ptr, d_fl.ptr, None))
                    pos = d_pos.to_numpy(np.uint32, n).astype(np.int64)
                    words = d_w.to_numpy(np.uint64, n)
                    bad = d_fl.to_numpy(np.uint8, n).astype(bool)
            keep = ~bad
            out.append((pos[keep] + off, np.full(int(keep.sum()), strand, np.uint8), words[keep], pos[keep]))
    return Sites(np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]),
                 np.concatenate([o[2] for o in out]), np.concatenate([o[3] for o in out]))


# ------------------------------------------------------------------ reference flavours
def _rc(s: str) -> str:
    return s[::-1].translate(str.maketrans("ATCGNatcgn", "TAGCNtagcn"))


def _greedy(pos: np.ndarray, pl: int) -> np.ndarray:
    """Left-to-right non-overlapping selection = re.finditer's behaviour for a
    fixed-length pattern (PAMProcessor.py:40)."""
    keep, last = [], -1
    for p in pos.tolist():
        if p >= last:
            keep.append(p)
            last = p + pl
    return np.asarray(keep, np.int64)


class PAMProcessor:
    def __init__(self, records, pam, direction):
        self.records = records
        self.pam = pam.replace("N", "[ATCG]")
        self.pam_raw = pam
        self.direction = direction

    def get_sequence(self, row):
        sequence = _seq_of(self.records[row.Chromosome])[row.Start:row.End]
        return _rc(sequence) if row.Strand == "-" else sequence

    def get_strand(self, strand_symbol):
        s = str(strand_symbol).lower().strip()
        if s in ("+", "1", "+1", "fwd", "forward"):
            return 1
        if s in ("-", "-1", "rev", "reverse"):
            return -1
        raise ValueError(f"Unrecognized strand symbol: {strand_symbol}")


def _seq_of(record) -> str:
    return str(record.seq) if hasattr(record, "seq") else str(record)


class GuideFinder:
    """Drop-in for PAMProcessor.py:27-57.  `records` maps id -> record; a record
    is anything with a str()-able `.seq`, or a plain string."""

    def __init__(self, records, pam, direction, length, device: int = 0):
        self.records = records
        self.pam = pam.replace("N", "[ATCG]")
        self.pam_raw = pam
        self.direction = direction
        self.length = length
        self.device = device

    def find_guides_from_pam(self):
        if self.direction not in ("downstream", "upstream"):
            raise ValueError("Direction must be 'upstream' or 'downstream'")
        guides = []
        pl, Lg = len(self.pam_raw), self.length
        for _rid, record in self.records.items():
            s = _seq_of(record)
            n = len(s)
            fwd, rev = pam_scan(pack_genome(s, 0, case_sensitive=True), self.pam_raw, self.device)
            # forward strand: matches in ascending position
            for p in _greedy(np.flatnonzero(fwd), pl).tolist():
                if self.direction == "downstream":
                    guides.append(s[max(0, p - Lg):p])
                else:
                    guides.append(s[p + pl:min(p + pl + Lg, n)])
            # reverse strand: position q on revcomp(s) <-> forward footprint [n-q-pl, n-q)
            q = (n - pl - np.flatnonzero(rev))[::-1]
            t = None
            for qq in _greedy(q, pl).tolist():
                if t is None:
                    t = _rc(s)
                if self.direction == "downstream":
                    guides.append(t[max(0, qq - Lg):qq])
                else:
                    guides.append(t[qq + pl:min(qq + pl + Lg, n)])
        return guides


class PAMFinder(PAMProcessor):
    """PAMProcessor.py:60-97 — PAM next to a hit interval (pure slicing)."""

    def __init__(self, records, pam, direction):
        super().__init__(records, pam, direction)
        self.pam_length = len(pam)

    def get_pam_seq(self, row):
        strand = self.get_strand(row.Strand)
        if self.direction not in ("upstream", "downstream"):
            raise ValueError("direction must be 'upstream' or 'downstream'")
        seq = _seq_of(self.records[row.Chromosome])
        # both branches of the reference take the same slices (PAMProcessor.py:69-89)
        if strand == 1:
            pam_sequence = seq[row.End:row.End + self.pam_length]
        else:
            pam_sequence = _rc(seq[row.Start - self.pam_length:row.Start])  # python slice semantics, like Bio.Seq
        return str(pam_sequence)

    def pam_matches(self, sequence):
        return bool(re.search(self.pam, sequence))


def find_sequences_with_barcode_and_pam(topological_fasta_file_name, barcode_length, pam,
                                        pam_direction: str = "downstream", device: int = 0):
    """design_guides.py:22-49 on the GPU: the set of pure-ACGT spacers of every
    PAM-adjacent site on both strands of every record of a (topological) FASTA.
    The reference reads the direction from a module global (`args.pam_direction`,
    :32/:40); here it is a parameter."""
    found = set()
    pl, k = len(pam), barcode_length
    for s in read_fasta_records(topological_fasta_file_name):
        found |= sliding_spacers(s, k, pam, pam_direction, device)
    return found


def sliding_spacers(seq: str, k: int, pam: str, direction: str = "downstream", device: int = 0) -> set:
    """Flavour B for one (already topological) sequence string."""
    n, pl = len(seq), len(pam)
    g = pack_genome(seq, 0, case_sensitive=True)
    st = find_sites(g, pam, k, direction, device)
    # the reference's loop bound i in range(len - k - pl + 1) (design_guides.py:31)
    # applies to the spacer start on the strand being read, for both directions
    i_fwd = st.start
    i_rev = n - k - st.start
    i = np.where(st.strand == 0, i_fwd, i_rev)
    lo = pl if direction == "upstream" else 0  # negative-slice quirk (:42-44): no match for i < |pam|
    keep = (i >= lo) & (i <= n - k - pl)
    return set(unpack_kmers(np.unique(st.spacer[keep]), k))
