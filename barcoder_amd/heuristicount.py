"""GPU barcode counter with the call shapes and outputs of the reference's
heuristicount.py.  The per-read hot loop (`process_chunk`) runs on the MI355X;
file reading, sampling and flank discovery are host logic.

Mirrors (reference file:line):
  rev_comp             heuristicount.py:29-30
  read_fasta           heuristicount.py:41-57
  open_reads_file      heuristicount.py:60-72
  validate_barcodes    heuristicount.py:75-97
  read_in_chunks       heuristicount.py:100-153
  sample_data          heuristicount.py:156-379
  find_flanks          heuristicount.py:382-425
  process_chunk        heuristicount.py:428-562   <- bc_pack_windows + bc_match (HIP)
  main                 heuristicount.py:565-888

Usage:  python -m barcoder_amd.heuristicount library.fasta reads1.fastq [reads2.fastq]
"""
from __future__ import annotations

import argparse
import gzip
import logging
import operator
import os
import sys
from collections import Counter
from contextlib import nullcontext

import numpy as np

from . import _native as N
from .encoding import hash_windows, pack_kmers, pack_windows, reads_to_text, unpack_kmers
from .library import DeviceLibrary
from .matcher import GpuMatcher

log = logging.getLogger("barcoder_amd.heuristicount")

_COMP = str.maketrans("ATCGN", "TAGCN")
FEEDBACK = ("If this assumption disrupts your pipeline, we welcome your feedback on our GitHub page:\n"
            "https://github.com/ryandward/barcoder/issues")


def rev_comp(sequence: str) -> str:
    return sequence[::-1].translate(_COMP)


def safe_len(s):
    return 0 if s is None else len(s)


def _zst_open(path, mode):
    try:
        import zstandard as zstd
    except ImportError as e:  # the reference needs the same optional package for .zst
        raise ValueError(f'"{path}": reading .zst needs the zstandard package') from e
    return zstd.open(path, mode)


def read_fasta(fasta_file):
    """Every non-header line, stripped, as a set (no upper-casing; a blank line
    yields "" exactly as in the reference)."""
    if fasta_file.endswith(".gz"):
        opener = gzip.open
    elif fasta_file.endswith(".zst"):
        opener = _zst_open
    elif fasta_file.endswith((".fasta", ".fa")):
        opener = open
    else:
        raise ValueError(f'"{fasta_file}" does not appear to be a supported fasta file: .fasta or .fa.')
    with opener(fasta_file, "rt") as fh:
        return {line.strip() for line in fh if not line.startswith(">")}


_ASCII_SPACE_NOT_NL = " \t\r\x0b\x0c\x1c\x1d\x1e\x1f"  # what str.strip() removes from an ASCII string, "\n" apart
from itertools import repeat as _repeat  # noqa: E402


def read_fasta_ordered(fasta_file, check_extension: bool = False):
    """Same lines as read_fasta, de-duplicated in file order (= guide index order)."""
    if check_extension and not fasta_file.endswith((".gz", ".zst", ".fasta", ".fa")):
        raise ValueError(f'"{fasta_file}" does not appear to be a supported fasta file: .fasta or .fa.')
    if fasta_file.endswith(".gz"):
        opener = gzip.open
    elif fasta_file.endswith(".zst"):
        opener = _zst_open
    else:
        opener = open
    with opener(fasta_file, "rt") as fh:
        text = fh.read()  # (text mode: every line end already "\n", as file iteration sees them)
    lines = text.split("\n")
    if lines and lines[-1] == "":  # (behind the last newline, or an empty file: iteration yields no such line)
        lines.pop()
    # the usual file — ASCII, no blank or tab anywhere, header and sequence lines alternating — needs no per-line work
    # (str.strip() has nothing to strip once "\n" is the only white space; 1e5 entries: 35 -> 15 ms of a job's fixed cost)
    if text.isascii() and not any(c in text for c in _ASCII_SPACE_NOT_NL):
        heads, seqs = lines[0::2], lines[1::2]
        if (len(heads) == len(seqs) and all(map(str.startswith, heads, _repeat(">")))
                and not any(map(str.startswith, seqs, _repeat(">")))):
            return list(dict.fromkeys(seqs))
        return list(dict.fromkeys([ln for ln in lines if not ln.startswith(">")]))
    return list(dict.fromkeys([ln.strip() for ln in lines if not ln.startswith(">")]))


def open_reads_file(file_path, mode):
    if file_path.endswith(".gz"):
        return gzip.open(file_path, mode)
    if file_path.endswith(".zst"):
        return _zst_open(file_path, mode)
    if file_path.endswith((".fastq", ".fq", ".reads")):
        return open(file_path, mode)
    raise ValueError(f'"{file_path}" does not appear to be a supported reads file: .fastq or .reads.')


def validate_barcodes(barcodes):
    if isinstance(barcodes, list):
        sequences = set(barcodes)
    elif isinstance(barcodes, set):
        sequences = barcodes
    else:
        raise ValueError("Pass a list or set of barcodes to validate.")
    if len(sequences) < 10:
        raise ValueError("The input contains fewer than 10 sequences. Please provide at least 10 short barcodes.")
    if sequences and max(map(len, sequences)) > 1000:  # (one C-level pass; the loop only to name the offender)
        for seq in sequences:
            if len(seq) > 1000:
                raise ValueError(f'The sequence "{seq[:30]}..." is longer than 1,000 bases. '
                                 "Provide a list or fasta file of short barcodes.")


def read_in_chunks(file1, file2=None, chunk_size=2 ** 16):
    """Yield (reads1, reads2|None) lists of `chunk_size` sequence lines."""
    base = os.path.splitext(file1)[0] if file1.endswith((".gz", ".zst")) else file1
    if base.endswith(".fastq"):
        per_record, seq_line = 4, 1
    elif base.endswith(".reads"):
        per_record, seq_line = 1, 0
    else:
        raise ValueError("Unsupported file type. Must be '.fastq' or '.reads'.")

    def sequences(handle):
        # the sequence line counts as soon as it is read, even if the record's
        # remaining lines are missing (the reference appends before it skips them)
        while True:
            for j in range(per_record):
                line = handle.readline()
                if not line:
                    return
                if j == seq_line:
                    yield line.strip()

    with open_reads_file(file1, "rt") as f1, (open_reads_file(file2, "rt") if file2 else nullcontext()) as f2:
        it1 = sequences(f1)
        it2 = sequences(f2) if file2 else None
        reads1, reads2 = [], []
        for r1 in it1:
            if it2 is not None:
                r2 = next(it2, None)
                if r2 is None:  # mate file ended first: the reference stops here, keeping r1
                    reads1.append(r1)
                    break
                reads2.append(r2)
            reads1.append(r1)
            if len(reads1) >= chunk_size:
                yield reads1[:chunk_size], (reads2[:chunk_size] if reads2 else None)
                reads1, reads2 = reads1[chunk_size:], (reads2[chunk_size:] if reads2 else [])
        if reads1:
            yield reads1, (reads2 if reads2 else None)


# ------------------------------------------------------------------------------ sampling
def sample_data(file1, file2, barcodes, is_paired):
    """Orientation / offset discovery on the first chunks of reads (host logic,
    heuristicount.py:156-379).  Returns the same 8-tuple as the reference."""
    n_lib = len(barcodes)
    rev_barcodes = {rev_comp(b) for b in barcodes}
    k = len(next(iter(barcodes)))
    div = [0, 0]
    orients = (Counter(), Counter())
    offsets = (Counter(), Counter())
    valid = (set(), set())
    seen_reads, observed, all_novel = set(), set(), []
    satisfied = False
    n_chunks = 0
    read1 = read2 = None
    for chunk1, chunk2 in read_in_chunks(file1, file2 if is_paired else None, chunk_size=n_lib):
        n_chunks += 1
        c_orients = ([], [])
        c_offsets = ([], [])
        c_barcodes, c_reads = set(), set()
        for read1, read2 in zip(chunk1, chunk2 if chunk2 else [None] * len(chunk1)):
            if read1 in c_reads or (read2 and read2 in c_reads):
                continue
            seen_reads.add(read1)
            if is_paired:
                seen_reads.add(read2)
            for i in range(len(read1) - k + 1):
                kmers = (read1[i:i + k], read2[i:i + k] if is_paired else None)
                for side in (0, 1):
                    kmer = kmers[side]
                    if side == 1 and not is_paired:
                        break
                    if kmer in c_barcodes:
                        break  # already seen this chunk: the reference `continue`s to the next i
                    read = read1 if side == 0 else read2
                    for lib, label in ((barcodes, "forward"), (rev_barcodes, "reverse")):
                        if kmer in lib:
                            div[side] += 1
                            c_barcodes.add(kmer)
                            observed.add(kmers[0])  # sic: the reference records read1's k-mer for both mates
                            c_orients[side].append(label)
                            c_offsets[side].append(i)
                            valid[side].add(read)
                            c_reads.add(read)
        all_novel.extend(c_barcodes)
        for side in (0, 1):
            orients[side].update(c_orients[side])
            offsets[side].update(c_offsets[side])
        top1, top2 = offsets[0].most_common(2), offsets[1].most_common(2)
        need = 5 * n_lib
        if is_paired:
            if div[0] >= need and div[1] >= need:
                log.info("Many barcodes seen enough in reads...")
                satisfied = True
        elif div[0] >= need:
            log.info("Many barcodes seen enough in reads...")
            satisfied = True
        if len(seen_reads) >= need and all_novel:
            log.info("Read depth diversity satisfied...")
            satisfied = True
        if len(all_novel) >= need:
            log.info("Barcode frequency diversity satisfied...")
            satisfied = True
        if satisfied:
            def dominant(top):
                return len(top) > 1 and top[0][1] >= 2 * top[1][1]
            if is_paired:
                done = (len(top1) == 1 and len(top2) == 1) or (dominant(top1) and dominant(top2))
            else:
                done = len(top1) == 1 or dominant(top1)
            if done:
                log.info("Dominant offsets found...")
                break
    if not satisfied:
        log.warning("Sequencing depth is probably insufficient! Continuing anyway...")

    def top(counter, present):
        return counter.most_common(1)[0][0] if present else None

    o1, off1 = top(orients[0], read1), top(offsets[0], read1)
    o2, off2 = top(orients[1], read2), top(offsets[1], read2)
    if o1 == "forward" or o2 == "reverse":
        return len(seen_reads), off1, off2, valid[0], valid[1], observed, False, n_chunks
    if o1 == "reverse" or o2 == "forward":
        return len(seen_reads), off2, off1, valid[1], valid[0], observed, True, n_chunks
    raise ValueError("Unable to determine orientation of reads. Please check the input files.")


def find_flanks(reads, start, bc_len, max_flank=10):
    """Majority <=max_flank-nt flanks left and right of the barcode window, a
    shorter flank displacing a longer one only when it is >3x as common
    (heuristicount.py:382-425)."""
    tallies = {"L": Counter(), "R": Counter()}
    for read in reads:
        left = read[start - max_flank:start] if start - max_flank >= 0 else read[0:start]
        right = read[start + bc_len:start + bc_len + max_flank]
        for n in range(len(left), 0, -1):
            tallies["L"][left[-n:]] += 1
        for n in range(len(right), 0, -1):
            tallies["R"][right[:n]] += 1

    def best(tally):
        chosen = None
        for n in range(max_flank, 0, -1):
            cands = [s for s in tally if len(s) == n]
            if not cands:
                continue
            top = max(cands, key=tally.__getitem__)
            if chosen is None or tally[top] > 3 * tally[chosen]:
                chosen = top
        return chosen

    return best(tallies["L"]), best(tallies["R"])


class NativeSample:
    """sample_data + find_flanks through the C ABI (bc_sample_*): same results as the
    Python functions above, without their per-k-mer interpreter cost."""

    def __init__(self, file1, file2, barcodes, is_paired):
        import ctypes as C
        ordered = list(barcodes)
        k = len(ordered[0])
        if any(len(b) != k for b in ordered):
            raise ValueError("All barcodes must be the same length")
        blob = "".join(ordered).encode("ascii", errors="replace")
        self._h = N.vp()
        f2 = file2.encode() if (is_paired and file2) else None
        N.check(N.load().bc_sample_data(file1.encode(), f2, blob, len(ordered), k, C.byref(self._h)))
        nn, o1, o2 = C.c_int64(), C.c_int64(), C.c_int64()
        v1, v2, ob, ch = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        sw, sat = C.c_int(), C.c_int()
        N.check(N.load().bc_sample_info(self._h, C.byref(nn), C.byref(o1), C.byref(o2), C.byref(v1), C.byref(v2),
                                        C.byref(ob), C.byref(sw), C.byref(ch), C.byref(sat)))
        self.n_novel = nn.value
        self.offset1 = None if o1.value < 0 else o1.value
        self.offset2 = None if o2.value < 0 else o2.value
        self.n_valid1, self.n_valid2, self.n_observed = v1.value, v2.value, ob.value
        self.need_swap, self.num_chunks, self.satisfied = bool(sw.value), ch.value, bool(sat.value)
        if not self.satisfied:
            log.warning("Sequencing depth is probably insufficient! Continuing anyway...")

    def flanks(self, which: int, start, bc_len: int, max_flank: int = 10):
        """find_flanks(sample1 if which == 0 else sample2, start, bc_len)."""
        import ctypes as C
        if start is None:  # the reference would slice with None; an empty sample gives (None, None) there
            return None, None
        L, R = C.create_string_buffer(max_flank + 1), C.create_string_buffer(max_flank + 1)
        ll, rl = C.c_int(), C.c_int()
        N.check(N.load().bc_sample_flanks(self._h, which, int(start), bc_len, max_flank, L, C.byref(ll), R, C.byref(rl)))
        return (None if ll.value < 0 else L.value.decode()), (None if rl.value < 0 else R.value.decode())

    def close(self):
        if self._h:
            N.load().bc_sample_free(self._h)
            self._h = N.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------ hot path
def window_labels(words, k: int):
    """Undocumented windows as strings: packed words (k <= 32) or rows of text (a 1-D array of k-byte void items:
    barcodes of more than 32 nt)."""
    words = np.asarray(words)
    if words.size == 0:
        return []
    if words.dtype.kind == "V":
        raw = words.tobytes().decode("ascii", errors="replace")
        return [raw[i * k:(i + 1) * k] for i in range(words.size)]
    return unpack_kmers(words, k)


class ChunkCounter:
    """Device-side state for process_chunk: the library in HBM plus the window
    geometry.  One instance serves every chunk of a run."""

    def __init__(self, barcodes, bc_len, L_fwd_start, L_rev_start, L_fwd, R_fwd, L_rev, R_rev, need_swap,
                 device: int = 0, max_mismatch: int = 0, algo: str = "auto", fwd_set=None, rev_set=None, lib=None):
        self.k = bc_len
        # more than 32 nt: exact match through a text hash, hits verified against the text (LongDeviceLibrary)
        self.long = bc_len > 32
        if self.long and max_mismatch:
            raise ValueError(f"mismatch-tolerant counting needs barcodes of at most 32 nt (got {bc_len})")
        self.geom = (L_fwd_start, L_rev_start, L_fwd or "", R_fwd or "", L_rev or "", R_rev or "")
        self.need_swap = need_swap
        self.max_mismatch, self.algo = max_mismatch, algo
        ordered = list(dict.fromkeys(barcodes))
        self._ordered = ordered
        # the flanked string sets are needed for odd windows only: built when the first one turns up
        self._fwd_set, self._rev_set = fwd_set, rev_set
        # the device library depends on the barcodes alone, not on the geometry: a caller may have started building it
        # before the geometry was known (count_barcodes does, under the sampling pass)
        self.lib = lib if lib is not None else self.device_library(ordered, bc_len, device)
        self.labels = self.lib.labels
        self.matcher = GpuMatcher(self.lib)

    @staticmethod
    def device_library(ordered, bc_len, device: int = 0):
        """The library in HBM for a de-duplicated barcode list: entries that cannot be 2-bit packed (wrong length, a
        character outside ACGT) are left out — they can only ever equal an ODD window, which the host compares as a
        string; the packer's flags say which (one native pass instead of a Python test per entry)."""
        same = [b for b in ordered if len(b) == bc_len]
        packed = None
        if same and bc_len <= 32:
            packed, bad = pack_kmers(same, bc_len)
            if bad.any():
                labels, packed = [b for b, f in zip(same, bad.tolist()) if not f], packed[bad == 0]
            else:
                labels = same
        else:
            labels = [b for b in same if set(b) <= set("ACGT")]
        if not labels:
            raise ValueError("no pure-ACGT barcode of the stated length in the library")
        # (the labels are distinct and already packed: no second pass over the strings)
        return DeviceLibrary(packed, bc_len, device, labels=labels) if packed is not None else \
            DeviceLibrary.from_strings(labels, device)

    @property
    def fwd_set(self):
        if self._fwd_set is None:
            self._fwd_set = {self.geom[2] + b + self.geom[3] for b in self._ordered}
        return self._fwd_set

    @property
    def rev_set(self):
        if self._rev_set is None:
            self._rev_set = {self.geom[4] + rev_comp(b) + self.geom[5] for b in self._ordered}
        return self._rev_set

    # windows of one mate: (packed, status, odd -> (in_lib, seq)); `buf` = (text, off)
    def _windows(self, buf, start, L, R, reverse, which):
        text, off = buf
        if self.long:
            packed, status, wins = hash_windows(text, off, start, L, R, self.k, reverse)
            self._wins[which] = wins
        else:
            packed, status = pack_windows(text, off, start, L, R, self.k, reverse)
        odd = {}
        for i in np.flatnonzero(status == N.WIN_ODD).tolist():
            w = _read_at(buf, i)[start:start + len(L) + self.k + len(R)]
            odd[i] = (w in (self.rev_set if which == "rev" else self.fwd_set), w[len(L):len(w) - len(R)])
        return packed, status, odd

    def process(self, chunk):
        """(Counter, n_reads) for one chunk of read lists, bit-for-bit the
        reference's result when max_mismatch == 0."""
        a, b = chunk
        return self.process_buffers(reads_to_text(a) if a else None, reads_to_text(b) if b else None)

    def process_buffers(self, buf_a, buf_b):
        """Same on (text, offsets) buffers (NativeReader output); either may be None."""
        return self.match_buffers(buf_a, buf_b).counter(self)

    def match_buffers(self, buf_a, buf_b, first_index: int = 0) -> "ChunkResult":
        """The device pass of one chunk, results left in array form (no per-key
        Python work): what the bulk driver accumulates."""
        buf1, buf2 = (buf_b, buf_a) if self.need_swap else (buf_a, buf_b)
        self._wins = {}
        n1 = 0 if buf1 is None else buf1[1].size - 1
        n2 = 0 if buf2 is None else buf2[1].size - 1
        s1, s2, Lf, Rf, Lr, Rr = self.geom
        odd_events = []  # (read index, key)
        if n1 and n2:
            if n1 != n2:
                raise ValueError("Length of reads1 and reads2 must be the same for paired-end data.")
            n = n1
            p1, st1, odd1 = self._windows(buf1, s1, Lf, Rf, False, "fwd")
            p2, st2, odd2 = self._windows(buf2, s2, Lr, Rr, True, "rev")
            ok = (st1 == N.WIN_OK) & (st2 == N.WIN_OK) & (p1 == p2)
            if self.long:  # equal hashes are a hint, equal texts the fact (seq1 == rev_comp(seq2), :506)
                ok &= (self._wins["fwd"] == self._wins["rev"]).all(axis=1)
            flags = (~ok).astype(np.uint8)
            res = self._match(p1, flags, first_index, "fwd")
            # pairs with a window that cannot be packed: exact string semantics on the host
            for i in sorted(set(odd1) | set(odd2)):
                x, y = _read_at(buf1, i), _read_at(buf2, i)
                if "N" in x or "N" in y:
                    continue
                w1 = x[s1:s1 + len(Lf) + self.k + len(Rf)]
                w2 = y[s2:s2 + len(Lr) + self.k + len(Rr)]
                q1, q2 = w1[len(Lf):len(w1) - len(Rf)], w2[len(Lr):len(w2) - len(Rr)]
                has = w1.startswith(Lf) and w1.endswith(Rf) and w2.startswith(Lr) and w2.endswith(Rr)
                if q1 != rev_comp(q2) or not has:
                    continue
                odd_events.append((first_index + i, q1 if (w1 in self.fwd_set and w2 in self.rev_set) else q1 + "*"))
            miss_words = p1
        else:
            reverse = not n1
            buf = buf2 if reverse else buf1
            n = n2 if reverse else n1
            if not n:
                return ChunkResult(0, None, None, None, None, None, [])
            start, L, R = (s2, Lr, Rr) if reverse else (s1, Lf, Rf)
            packed, status, odd = self._windows(buf, start, L, R, reverse, "rev" if reverse else "fwd")
            res = self._match(packed, (status != N.WIN_OK).astype(np.uint8), first_index, "rev" if reverse else "fwd")
            for i, (in_lib, seq) in odd.items():
                if reverse:
                    seq = rev_comp(seq)
                odd_events.append((first_index + i, seq if in_lib else seq + "*"))
            miss_words = packed
        # undocumented: group the missed (but flank-valid) windows
        miss = np.flatnonzero(res.best_idx == N.IDX_MISS)
        if self.long:  # the windows' texts (k bytes each) stand where the packed words do
            wins = self._wins["fwd" if n1 else "rev"]  # (pairs are matched on the forward mate's window)
            miss_words = np.ascontiguousarray(wins).view(np.dtype((np.void, self.k))).ravel()
        if miss.size:
            words, first, cnt = np.unique(miss_words[miss], return_index=True, return_counts=True)
            first = miss[first].astype(np.uint64) + np.uint64(first_index)
        else:
            words, first, cnt = miss_words[:0], np.zeros(0, np.uint64), np.zeros(0, np.int64)
        return ChunkResult(n, res.counts, res.first_seen, words, cnt.astype(np.uint64), first, odd_events)

    def _match(self, words, flags, first_index, which):
        """The GPU lookup of one chunk.  Long barcodes: the exact table is keyed on text hashes, so every hit is held
        against the barcode's text, and the tallies are taken from the verified hits."""
        if not self.long:
            return self.matcher.match(words, flags, self.max_mismatch, self.algo, first_index=first_index)
        res = self.matcher.match(words, flags, 0, "hash", first_index=first_index)
        hit = np.flatnonzero(res.best_idx >= 0)
        if hit.size:
            same = (self.lib.label_bytes[res.best_idx[hit]] == self._wins[which][hit]).all(axis=1)
            res.best_idx[hit[~same]] = N.IDX_MISS  # a hash collision with a library entry: not that barcode
            hit = hit[same]
        G = self.lib.size
        res.counts = np.bincount(res.best_idx[hit], minlength=G).astype(np.uint64)
        res.first_seen = np.full(G, np.iinfo(np.uint64).max, np.uint64)
        if hit.size:
            np.minimum.at(res.first_seen, res.best_idx[hit][::-1], (hit[::-1] + first_index).astype(np.uint64))
        return res


class ChunkResult:
    """One chunk's outcome in array form: documented counts / first-seen per guide,
    undocumented packed windows with counts and first indices, odd (string) events."""

    def __init__(self, n, counts, first_seen, u_words, u_counts, u_first, odd_events):
        self.n, self.counts, self.first_seen = n, counts, first_seen
        self.u_words, self.u_counts, self.u_first, self.odd_events = u_words, u_counts, u_first, odd_events

    def counter(self, cc):
        """The reference's (Counter, n): keys inserted in first-seen order."""
        if self.n == 0:
            return Counter(), 0
        events = [(f, key, 1) for f, key in self.odd_events]
        seen = np.flatnonzero(self.counts)
        events += [(f, cc.labels[g], c) for g, c, f in
                   zip(seen.tolist(), self.counts[seen].tolist(), self.first_seen[seen].tolist())]
        if self.u_words.size:
            events += [(f, s + "*", c) for s, f, c in
                       zip(window_labels(self.u_words, cc.k), self.u_first.tolist(), self.u_counts.tolist())]
        counts = Counter()
        for _f, key, c in sorted(events, key=lambda e: e[0]):
            counts[key] += c
        return counts, self.n


class PackedCounter:
    """Counter-like view over undocumented windows kept packed (millions of
    distinct keys would cost minutes as Python strings): supports len(),
    values(), most_common(k), items() and `key in`.  items() runs in first-seen
    order; the ordering is only computed when something asks for it."""

    def __init__(self, k, words, counts, first, extra: Counter):
        self.k, self._words, self._counts, self._first, self.extra = k, words, counts, first, extra
        self._ordered = False

    def _order(self):
        if not self._ordered:
            order = np.argsort(self._first, kind="stable")
            self._words, self._counts, self._first = self._words[order], self._counts[order], self._first[order]
            self._ordered = True

    @property
    def words(self):
        self._order()
        return self._words

    @property
    def counts(self):
        self._order()
        return self._counts

    @property
    def first(self):
        self._order()
        return self._first

    def __len__(self):
        return int(self._words.size) + len(self.extra)

    def values(self):
        return list(self._counts.tolist()) + list(self.extra.values())

    def total(self):
        return int(self._counts.sum()) + sum(self.extra.values())

    def most_common(self, k=None):
        n = int(self._counts.size)
        if k is None:
            k = len(self)
        c = self._counts.astype(np.int64)
        if 0 < k < n:  # top k without sorting millions of entries: partition, then order the few candidates
            kth = np.partition(c, n - k)[n - k]
            gt = np.flatnonzero(c > kth)
            eq = np.flatnonzero(c == kth)
            need = k - gt.size
            if eq.size > need:  # many ties at the cut (say, millions of singletons): the earliest-seen ones win
                eq = eq[np.argpartition(self._first[eq], need - 1)[:need]]
            cand = np.concatenate([gt, eq])
        else:
            cand = np.arange(n)
        # ties resolve towards the earlier first-seen entry, like Counter.most_common on insertion order
        top = cand[np.lexsort((self._first[cand], -c[cand]))][:k]
        out = [(s + "*", int(v)) for s, v in zip(window_labels(self._words[top], self.k), self._counts[top].tolist())]
        out += list(self.extra.items())
        return sorted(out, key=lambda kv: -kv[1])[:k]

    def items(self):
        yield from ((s + "*", int(c)) for s, c in zip(window_labels(self.words, self.k), self.counts.tolist()))
        yield from self.extra.items()


def _read_at(buf, i: int) -> str:
    text, off = buf
    a, b = int(off[i]), int(off[i + 1])
    raw = text[a:b]
    return (raw.tobytes() if isinstance(raw, np.ndarray) else bytes(raw)).decode("ascii", errors="replace")


_COUNTERS = {}


def process_chunk(chunk, bcs_with_flanks_fwd, bcs_with_flanks_rev, L_fwd_start, L_rev_start, bc_len, L_fwd, R_fwd,
                  L_rev, R_rev, need_swap, device: int = 0):
    """Drop-in for heuristicount.py:428-562 with the same 11 positional
    arguments and the same (Counter, int) result; the set lookups and Counter
    updates run on the GPU.  The device library is built from the flanked sets
    on first use and cached for later chunks."""
    key = (id(bcs_with_flanks_fwd), id(bcs_with_flanks_rev), len(bcs_with_flanks_fwd or ()),
           len(bcs_with_flanks_rev or ()), L_fwd_start, L_rev_start, bc_len, L_fwd, R_fwd, L_rev, R_rev, need_swap,
           device)
    cc = _COUNTERS.get(key)
    if cc is None:
        Lf, Rf, Lr, Rr = (x or "" for x in (L_fwd, R_fwd, L_rev, R_rev))
        if bcs_with_flanks_fwd:
            barcodes = [s[len(Lf):len(s) - len(Rf)] for s in bcs_with_flanks_fwd]
        else:
            barcodes = [rev_comp(s[len(Lr):len(s) - len(Rr)]) for s in bcs_with_flanks_rev]
        cc = ChunkCounter(barcodes, bc_len, L_fwd_start, L_rev_start, L_fwd, R_fwd, L_rev, R_rev, need_swap,
                          device=device, fwd_set=bcs_with_flanks_fwd or set(), rev_set=bcs_with_flanks_rev or set())
        if len(_COUNTERS) > 8:
            _COUNTERS.clear()
        _COUNTERS[key] = cc
    return cc.process(chunk)


# ------------------------------------------------------------------------------ driver
def merge_undocumented(tables, bc_len: int = 0):
    """[(words, counts, first), ...] -> one table: unique packed windows with summed counts and
    the smallest first index (any order of the input tables: ranks, chunks).  bc_len > 32 names the
    width of the text rows the empty table has then (a rank that owned no chunk, an empty input):
    packed-word code must never see it as k-mers."""
    tables = [t for t in tables if t[0] is not None and t[0].size]
    if not tables:
        z = np.zeros(0, np.uint64)
        return (np.zeros(0, np.dtype((np.void, bc_len))) if bc_len > 32 else z), z, z
    if np.asarray(tables[0][0]).dtype.kind == "V":  # windows of more than 32 nt: rows of text instead of packed words
        w = np.concatenate([t[0] for t in tables])
        c = np.concatenate([t[1] for t in tables]).astype(np.uint64)
        f = np.concatenate([t[2] for t in tables]).astype(np.uint64)
        uw, inv = np.unique(w, return_inverse=True)
        uc = np.zeros(uw.size, np.uint64)
        uf = np.full(uw.size, np.iinfo(np.uint64).max, np.uint64)
        np.add.at(uc, inv, c)
        np.minimum.at(uf, inv, f)
        return uw, uc, uf
    if len(tables) == 1 and (tables[0][0].size < 2 or bool((tables[0][0][1:] > tables[0][0][:-1]).all())):
        return tuple(np.asarray(x, np.uint64) for x in tables[0])  # already one sorted table of distinct words
    w = np.concatenate([t[0] for t in tables])
    c = np.concatenate([t[1] for t in tables]).astype(np.uint64)
    f = np.concatenate([t[2] for t in tables]).astype(np.uint64)
    order = np.lexsort((f, w))
    w, c, f = w[order], c[order], f[order]
    starts = np.flatnonzero(np.concatenate(([True], w[1:] != w[:-1])))
    return w[starts], np.add.reduceat(c, starts), f[starts]


def shard_plan(file1, file2, world: int) -> str:
    """How the bulk pass is split over `world` ranks: "bytes" = byte ranges of one uncompressed
    single-end file (every rank reads only its part); "stride" = chunk c belongs to rank
    c % world (compressed or paired input: every rank decodes the stream, the per-read work
    — windows, match, tally — is split)."""
    plain = not file1.endswith((".gz", ".zst"))
    return "bytes" if (world > 1 and not file2 and plain) else "stride"


def count_barcodes(fasta_file, file1, file2=None, device: int = 0, max_mismatch: int = 0, rank: int = 0,
                   world: int = 1, chunk_reads: int = 1 << 20):
    """Everything main() computes, returned instead of printed: dict with
    doc (first-seen ordered {barcode: count}), undoc Counter, total_reads, and
    the discovered geometry.

    rank / world: this process is one of `world` ranks of a torch.distributed group (one per
    GPU, barcoder_amd.dist): the bulk pass covers this rank's shard and the per-guide counts /
    first-seen keys are all-reduced (dist.count_sharded), which replaces the Pool fan-out and
    Counter merge of heuristicount.py:703-737.  Every rank returns the job-wide result."""
    from . import dist as D
    import time as _time
    t_start = _time.perf_counter()
    ordered_lines = read_fasta_ordered(fasta_file, check_extension=True)  # one pass over the file: set and file order
    barcodes = set(ordered_lines)
    validate_barcodes(barcodes)
    lengths = set(map(len, ordered_lines))
    if len(lengths) != 1:
        log.warning("One of the assumptions of this analysis is that all barcodes are the same length\n%s", FEEDBACK)
        raise ValueError("All barcodes must be the same length")
    bc_len = lengths.pop()
    is_paired = bool(file2)
    log.info("Sampling reads to identify diversity characteristics...")
    ordered = ordered_lines
    # the device library needs the barcodes only: it is built (GPU start-up included) on a second thread while the
    # sampling pass — pure host work inside one native call, the GIL released — discovers the geometry
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    t_lib0 = _time.perf_counter()
    lib_future = pool.submit(lambda: (ChunkCounter.device_library(ordered, bc_len, device), _time.perf_counter()))
    try:
        smp = NativeSample(file1, file2, ordered, is_paired)
        bc_start1, bc_start2, need_swap = smp.offset1, smp.offset2, smp.need_swap
        log.info("Sampled %s diverse contexts in %s chunks and found %s barcodes in %s forward and %s reverse reads...",
                 f"{smp.n_novel:,}", smp.num_chunks, f"{smp.n_observed:,}", f"{smp.n_valid1:,}", f"{smp.n_valid2:,}")
        fwd_name = os.path.basename(file1)
        rev_name = os.path.basename(file2) if file2 else None
        if need_swap:
            log.info("Swapping orientation...")
            fwd_name, rev_name = rev_name, fwd_name
        log.info("Identifying flanking sequences...")
        # sample1 / sample2 are never None in the reference (empty sets at worst): heuristicount.py:626-638
        L_fwd, R_fwd = smp.flanks(0, bc_start1, bc_len)
        L_fwd_start = bc_start1 - len(L_fwd) if L_fwd else 0
        L_rev, R_rev = smp.flanks(1, bc_start2, bc_len)
        L_rev_start = bc_start2 - len(L_rev) if L_rev else 0
        smp.close()
        # the two mates must show reverse-complementary flanks (heuristicount.py:641-688)
        L_rev_rc = rev_comp(L_rev) if L_rev else None
        R_rev_rc = rev_comp(R_rev) if R_rev else None
        violated = False
        if L_fwd and R_rev_rc:
            m = min(len(L_fwd), len(R_rev_rc))
            violated |= L_fwd[-m:] != R_rev_rc[:m]
        if R_fwd and L_rev_rc:
            m = min(len(R_fwd), len(L_rev_rc))
            violated |= R_fwd[:m] != L_rev_rc[:m]
        if violated:
            log.warning("Expected complemtarity not found for flanking sequences...")
            if L_fwd and R_rev_rc:
                log.info("Left flank:  '%s' (./%s) != '%s' (./%s revcomp)", L_fwd, fwd_name, R_rev_rc, rev_name)
            if R_fwd and L_rev_rc:
                log.info("Right flank: '%s' (./%s) != '%s' (./%s revcomp)", R_fwd, fwd_name, L_rev_rc, rev_name)
            raise ValueError("A critical error occurred: Flank complementarity violation")

        log.info("Executing high-throughput read analysis...")
        t_sampled = _time.perf_counter()
        dev_lib, t_lib_done = lib_future.result()
    except BaseException:
        lib_future.cancel()  # (a build already running finishes on its thread and is dropped)
        raise
    finally:
        pool.shutdown(wait=True)
    cc = ChunkCounter(ordered, bc_len, L_fwd_start, L_rev_start, L_fwd, R_fwd, L_rev, R_rev, need_swap,
                      device=device, max_mismatch=max_mismatch, lib=dev_lib)
    t_lib = _time.perf_counter()
    G = len(cc.labels)
    plan = shard_plan(file1, file2, world)
    # the device path takes a single file — plain, or .gz / .zst decoded into the upload buffers — whose start offset is
    # known (not None after sampling)
    # (barcodes of more than 32 nt: text hashes + verification on the device too; paired input of such barcodes takes the
    # host front half, see ChunkCounter._match)
    compressed1 = file1.endswith((".gz", ".zst"))
    device_ingest = (not is_paired and (plan == "bytes" or world == 1 or compressed1)
                     and (bc_start2 if need_swap else bc_start1) is not None and os.environ.get("BC_DEVICE_INGEST", "1") != "0"
                     and len(L_fwd or "") <= 15 and len(R_fwd or "") <= 15 and len(L_rev or "") <= 15 and len(R_rev or "") <= 15)
    # paired-end: two uncompressed files; with several ranks every rank takes byte ranges of both files that hold the same
    # reads (ingest.paired_read_ranges: newline counts per slice, made common knowledge by one small gather)
    compressed_pair = is_paired and (file1.endswith((".gz", ".zst")) or file2.endswith((".gz", ".zst")))
    # (barcodes of more than 32 nt pair on the device too: the stores keep bc_len bytes of text per read and file — up to
    # 256 nt, beyond that the host front half)
    device_pairs = (bc_len <= 256 and is_paired and (not compressed_pair or world == 1)
                    and bc_start1 is not None and bc_start2 is not None and os.environ.get("BC_DEVICE_INGEST", "1") != "0"
                    and len(L_fwd or "") <= 15 and len(R_fwd or "") <= 15 and len(L_rev or "") <= 15 and len(R_rev or "") <= 15)
    undoc_tables, odd_events = [], []
    n_mine = [0]
    t_bulk = [0.0]
    from .ingest import NativeReader

    def bulk(a, b):
        """dist.count_sharded's match_shard.  Plan "bytes": [a, b) is a byte range of file1; plan
        "stride": [a, b) are the chunk residues (c % world) this rank owns (everything when world = 1).
        Native reader (bc_reader_*) -> bc_pack_windows -> GPU match, 1M reads at a time; per-chunk results
        stay in array form and are merged once at the end.  The first-seen key of a read is its global
        index (stride) or the file offset of its chunk + its position in the chunk (bytes): both grow
        with the read's position in the file, which is all the output order needs."""
        counts = np.zeros(G + 1, np.uint64)  # last element: reads seen by this rank
        first_seen = np.full(G + 1, np.iinfo(np.uint64).max, np.uint64)
        t0 = _time.perf_counter()
        if device_pairs:
            # both mate files through the device (bc_stream_count_pair); the file whose reads carry the barcode in
            # library orientation is reads1 after the reference's swap (heuristicount.py:444-447)
            from .ingest import paired_read_ranges, stream_count_pair
            s1, s2, Lf, Rf, Lr, Rr = cc.geom
            p_fwd, p_rev = (file2, file1) if need_swap else (file1, file2)
            c64, f64 = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
            key0, rng_f, rng_r = (0, None, None) if world == 1 else paired_read_ranges(p_fwd, p_rev, rank, world, D.gather_objects)
            sr = stream_count_pair(cc.lib.handle, p_fwd, p_rev, s1, Lf, Rf, s2, Lr, Rr, max_mismatch, cc.algo, key0, c64, f64,
                                   range_fwd=rng_f, range_rev=rng_r)
            counts[:G], first_seen[:G] = c64, f64
            undoc_tables.append((sr.u_words, sr.u_counts, sr.u_first))
            for key, w1p, st1, w2p, st2, t1, t2 in sr.odd:  # string semantics (ChunkCounter.match_buffers)
                if st1 == N.WIN_SKIP or st2 == N.WIN_SKIP:
                    continue
                # a packed mate's barcode: from its word, or (more than 32 nt: the word is a hash) the text that came along
                b1 = t1 if (bc_len > 32 and st1 != N.WIN_ODD) else None
                b2 = t2 if (bc_len > 32 and st2 != N.WIN_ODD) else None
                w1 = t1[s1:s1 + len(Lf) + bc_len + len(Rf)] if st1 == N.WIN_ODD else \
                    Lf + (b1 if b1 is not None else unpack_kmers(np.array([w1p], np.uint64), bc_len)[0]) + Rf
                w2 = t2[s2:s2 + len(Lr) + bc_len + len(Rr)] if st2 == N.WIN_ODD else \
                    Lr + rev_comp(b2 if b2 is not None else unpack_kmers(np.array([w2p], np.uint64), bc_len)[0]) + Rr
                q1, q2 = w1[len(Lf):len(w1) - len(Rf)], w2[len(Lr):len(w2) - len(Rr)]
                has = w1.startswith(Lf) and w1.endswith(Rf) and w2.startswith(Lr) and w2.endswith(Rr)
                if q1 != rev_comp(q2) or not has:
                    continue
                odd_events.append((key, q1 if (w1 in cc.fwd_set and w2 in cc.rev_set) else q1 + "*"))
            n_mine[0] = sr.n_reads
            counts[G] = sr.n_reads
            t_bulk[0] = _time.perf_counter() - t0
            return counts, first_seen
        if device_ingest:
            # single uncompressed file: raw bytes -> HBM, everything per read on the device (bc_stream_count_file).
            # Nominal byte cuts become record-aligned ranges; the key of a read = range start + its ordinal.
            from .ingest import record_boundary, stream_count_file
            part = None
            if plan == "bytes":
                lo, hi = record_boundary(file1, a), record_boundary(file1, b)
            else:
                lo, hi = 0, os.path.getsize(file1)
                if world > 1:  # a compressed file over several ranks: chunk c of the decoded stream is rank c % world's
                    part = (rank, world)
            s1, s2, Lf, Rf, Lr, Rr = cc.geom
            reverse = bool(need_swap)
            start, L, R = (s2, Lr, Rr) if reverse else (s1, Lf, Rf)
            c64, f64 = np.zeros(G, np.uint64), np.full(G, np.iinfo(np.uint64).max, np.uint64)
            sr = stream_count_file(cc.lib.handle, file1, lo, hi, start, L, R, reverse, max_mismatch, cc.algo, lo, c64, f64, part=part)
            counts[:G], first_seen[:G] = c64, f64
            undoc_tables.append((sr.u_words, sr.u_counts, sr.u_first))
            for key, text in sr.odd:  # string semantics for windows that cannot be packed (ChunkCounter._windows)
                w = text[start:start + len(L) + bc_len + len(R)]
                seq = w[len(L):len(w) - len(R)]
                if reverse:
                    seq = rev_comp(seq)
                odd_events.append((key, seq if w in (cc.rev_set if reverse else cc.fwd_set) else seq + "*"))
            n_mine[0] = sr.n_reads
            counts[G] = sr.n_reads
            t_bulk[0] = _time.perf_counter() - t0
            return counts, first_seen
        rng = (a, b) if plan == "bytes" else None
        seen = 0
        with NativeReader(file1, max_reads=chunk_reads, byte_range=rng) as r1, \
                (NativeReader(file2, max_reads=chunk_reads) if is_paired else nullcontext()) as r2:
            c = 0
            while True:
                key = r1.tell() if plan == "bytes" else seen
                b1 = r1.next()
                n1 = b1[1].size - 1
                b2 = None
                if r2 is not None:
                    b2 = r2.next(max(n1, 1))  # same number of mates; a shorter file ends the pairing with a ValueError
                n2 = 0 if b2 is None else b2[1].size - 1
                if n1 == 0 and n2 == 0:
                    break
                if n1 and n2 and n1 != n2:  # every rank sees this, whoever owns the chunk
                    raise ValueError("Length of reads1 and reads2 must be the same for paired-end data.")
                mine = plan == "bytes" or (a <= c % world < b)
                c += 1
                seen += max(n1, n2)
                if not mine:
                    continue
                res = cc.match_buffers(b1 if n1 else None, b2 if n2 else None, first_index=key)
                n_mine[0] += res.n
                if res.n:
                    counts[:G] += res.counts
                    np.minimum(first_seen[:G], res.first_seen, out=first_seen[:G])
                    undoc_tables.append((res.u_words, res.u_counts, res.u_first))
                    odd_events.extend(res.odd_events)
        counts[G] = n_mine[0]
        t_bulk[0] = _time.perf_counter() - t0
        return counts, first_seen

    n_units = os.path.getsize(file1) if plan == "bytes" else world
    from . import devmem
    devmem.drain()  # the bulk pass allocates on the C side (stream slots, window stores): idle pooled buffers go first
    counts, first_seen = D.count_sharded(n_units, bulk, G + 1, rank, world, device=device)
    total_reads = int(counts[G])
    counts, first_seen = counts[:G], first_seen[:G]
    mine_u = merge_undocumented(undoc_tables, bc_len)
    if world > 1:  # variable-length leftovers: undocumented windows (padded tensors) and the few odd string events
        if bc_len > 32:  # rows of text: few enough to pickle
            mine_u = merge_undocumented(D.gather_objects(mine_u), bc_len)
        else:
            mine_u = merge_undocumented(D.gather_tables(mine_u, device=device))
        odd_events = [e for part in D.gather_objects(odd_events) for e in part]
    uw, uc, uf = mine_u
    # documented: first-seen order (heuristicount.py:876-877 prints Counter insertion order)
    extra_undoc = Counter()
    odd_doc = []
    for f, key in odd_events:
        if key.endswith("*"):
            extra_undoc[key] += 1
        else:
            odd_doc.append((f, key, 1))
    seen = np.flatnonzero(counts)
    if not odd_doc:
        # the usual case, without a Python object per event: a guide's first read is its own (one read, one guide), so
        # the first-seen keys are distinct and their order is the Counter's insertion order
        order = seen[np.argsort(first_seen[seen], kind="stable")]
        idx = order.tolist()
        names = operator.itemgetter(*idx)(cc.labels) if len(idx) > 1 else tuple(cc.labels[i] for i in idx)
        doc = Counter(dict(zip(names, counts[order].tolist())))
    else:  # documented barcodes seen in odd windows (string semantics): events merged by first-seen key
        doc_events = [(int(f), cc.labels[g], int(c)) for g, c, f in
                      zip(seen.tolist(), counts[seen].tolist(), first_seen[seen].tolist())] + odd_doc
        doc = Counter()
        for _f, key, c in sorted(doc_events, key=lambda e: e[0]):
            doc[key] += c
    undoc = PackedCounter(bc_len, uw, uc, uf, extra_undoc)
    return dict(doc=doc, undoc=undoc, total_reads=total_reads, barcodes=barcodes, bc_len=bc_len,
                bc_start1=bc_start1, bc_start2=bc_start2, flanks_fwd=(L_fwd, R_fwd), flanks_rev=(L_rev, R_rev),
                need_swap=need_swap, forward_name=fwd_name, reverse_name=rev_name, shard_plan=plan,
                bulk_pass=("device ingest (bc_stream_count_pair)" if device_pairs else
                           "device ingest (bc_stream_count_file)" if device_ingest else "host front half"),
                bulk_mreads_per_s=(n_mine[0] / t_bulk[0] / 1e6 if t_bulk[0] > 0 else None),
                seconds=dict(sampling=t_sampled - t_start, library=t_lib - t_sampled,
                             library_build_overlapped=t_lib_done - t_lib0, bulk=t_bulk[0],
                             merge=_time.perf_counter() - t_lib - t_bulk[0]))


def summary_lines(r, fasta_file):
    """The fields of the reference's rich summary table (heuristicount.py:757-874) as plain text."""
    doc, undoc, total = r["doc"], r["undoc"], r["total_reads"]
    nd, nu = sum(doc.values()), (undoc.total() if hasattr(undoc, "total") else sum(undoc.values()))
    frac = (lambda x: f"{(x / total if total else 0):.3f}")
    rows = [("Barcodes", os.path.basename(fasta_file)), ("Forward Reads", r["forward_name"]),
            ("Reverse Reads", r["reverse_name"]), ("Barcode Length", r["bc_len"]),
            ("Forward Offset", r["bc_start1"]), ("Reverse Offset", r["bc_start2"]),
            ("Forward Flanks", "{}...{}".format(*r["flanks_fwd"]) if any(r["flanks_fwd"]) else None),
            ("Reverse Flanks", "{}...{}".format(*r["flanks_rev"]) if any(r["flanks_rev"]) else None),
            ("Total Reads", f"{total:,}"), ("Documented Barcode Reads", f"{nd:,}"),
            ("Undocumented Barcode Reads", f"{nu:,}"), ("Documented Barcodes", f"{len(r['barcodes']):,}"),
            ("Seen Documented Barcodes", f"{len(doc):,}"),
            ("Unseen Documented Barcodes", f"{len(r['barcodes']) - len(doc):,}"),
            ("Uncocumented Barcodes", f"{len(undoc):,}"), ("Barcoded Reads Fraction", frac(nd + nu)),
            ("Documented Fraction", frac(nd)), ("Undocumented Fraction", frac(nu))]
    out = [f"{k:>30}  {v}" for k, v in rows if v]
    for title, c in (("Documented", doc), ("Undocumented", undoc)):
        top = c.most_common(min(5, len(c)))
        out.append(f"{'Top %d %s Barcodes' % (len(top), title):>30}")
        out += [f"{b:>30}  {n:,}" for b, n in top]
    return out


def main(args, device: int = 0, rank: int = 0, world: int = 1):
    """Same contract as heuristicount.main: log to stderr, `barcode<TAB>count`
    for documented barcodes in first-seen order on stdout, errors logged and
    swallowed (heuristicount.py:882-888).  With world > 1 every rank computes,
    rank 0 prints."""
    log.info("Initializing heuristic barcode counting...")
    try:
        r = count_barcodes(args.fasta_file, args.file1, args.file2, device=device,
                           max_mismatch=getattr(args, "mismatches", 0), rank=rank, world=world,
                           chunk_reads=getattr(args, "chunk_reads", 1 << 20))
        if rank != 0:
            return True
        for line in summary_lines(r, args.fasta_file):
            print(line, file=sys.stderr)
        for barcode, count in r["doc"].items():
            print("\t".join([barcode, str(count)]))
        return True
    except ValueError as ve:
        log.error("%s", ve)
    except Exception as e:  # noqa: BLE001 - mirrors the reference
        log.error("An unexpected error occurred: %s", e)
    return False  # (the reference's main returns None either way; the multi-rank launcher needs to know)


def _cli():
    p = argparse.ArgumentParser(description="Process Barcodes.")
    p.add_argument("fasta_file", type=str, help="List or FASTA file containing barcodes.")
    p.add_argument("file1", type=str, help="First reads file: FASTQ or raw reads.")
    p.add_argument("file2", type=str, nargs="?", default=None, help="Second reads file: FASTQ or raw reads (optional).")
    p.add_argument("--mismatches", type=int, default=0, help="count reads within this many substitutions (extension)")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--chunk-reads", type=int, default=1 << 20, help="reads per device pass (default 1048576)")
    p.add_argument("--gpus", type=int, default=1,
                   help="ranks (one per GPU) the bulk pass is sharded over; replaces the reference's Pool fan-out")
    p.add_argument("--timeout", type=float, default=-1.0,
                   help="with --gpus N: seconds after which ranks still running are ended (default BC_LAUNCH_TIMEOUT or 24 h)")
    a = p.parse_args()
    logging.basicConfig(level=logging.INFO, stream=sys.stderr, format="%(levelname)s %(message)s")
    from . import dist as D
    if a.gpus > 1 and not D.under_launcher():
        # start the ranks as fresh processes; this parent never touches the GPU
        env = dict(os.environ)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        raise SystemExit(D.launch_workers(["-m", "barcoder_amd.heuristicount"] + sys.argv[1:], a.gpus, env=env, timeout=a.timeout))
    if D.under_launcher():
        import contextlib
        import io
        buf = io.StringIO()
        with D.stdout_to_stderr():  # native libraries (gloo) print to fd 1: keep stdout for the TSV alone
            rank, world, local = D.init_from_env()
            if rank != 0:
                logging.getLogger().setLevel(logging.WARNING)
            with contextlib.redirect_stdout(buf):
                ok = main(a, device=local, rank=rank, world=world)
            if not ok and world > 1:
                # this rank failed (its error is logged): the others may be waiting in a collective it will never
                # join — leave without the barrier, non-zero, and launch_workers / torchrun ends the peers
                sys.stderr.flush()
                os._exit(1)
            D.barrier()
            D.shutdown()
        sys.stdout.write(buf.getvalue())
        sys.stdout.flush()
    else:
        main(a, device=a.device)


if __name__ == "__main__":
    _cli()
