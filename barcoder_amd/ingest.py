"""Native reads-file reader (bc_reader_*): the sequence lines of .fastq / .reads
(.gz, .zst) files as text + offsets buffers, the input form of bc_pack_windows.
Host-side replacement for the per-line Python loop of heuristicount.py:100-153."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as N


class NativeReader:
    def __init__(self, path: str, max_reads: int = 1 << 20, bytes_per_read: int = 192, byte_range=None):
        """byte_range=(begin, end): only the records that start in that byte range of an
        uncompressed file (one rank's shard, bc_reader_open_range)."""
        self._h = N.vp()
        if byte_range is None:
            N.check(N.load().bc_reader_open(path.encode(), C.byref(self._h)))
        else:
            N.check(N.load().bc_reader_open_range(path.encode(), int(byte_range[0]), int(byte_range[1]),
                                                  C.byref(self._h)))
        self.max_reads = int(max_reads)
        self._text = np.empty(self.max_reads * bytes_per_read, np.uint8)
        self._off = np.empty(self.max_reads + 1, np.uint64)

    def _next_once(self, m: int):
        n = C.c_uint64(0)
        while True:
            rc = N.load().bc_reader_next(self._h, m, N.ptr(self._text), self._text.size, N.ptr(self._off), C.byref(n))
            if rc == N.BC_ECAP:  # one read larger than the whole buffer: grow and retry
                self._text = np.empty(self._text.size * 4, np.uint8)
                continue
            N.check(rc)
            break
        k = int(n.value)
        return self._text[:int(self._off[k])], self._off[:k + 1]

    def tell(self) -> int:
        """Offset of the next unread byte in the decoded stream (bc_reader_tell)."""
        v = C.c_uint64(0)
        N.check(N.load().bc_reader_tell(self._h, C.byref(v)))
        return int(v.value)

    def next(self, max_reads: int | None = None):
        """(text uint8[...], off uint64[n+1]); n = min(max_reads, reads left in the
        file).  The views are valid until the next call; n == 0 at EOF."""
        m = self.max_reads if max_reads is None else min(int(max_reads), self.max_reads)
        text, off = self._next_once(m)
        got = off.size - 1
        if got == 0 or got == m:
            return text, off
        # the text buffer filled up before m reads: keep going and stitch the pieces
        texts, offs = [text.copy()], [off.copy()]
        while got < m:
            t, o = self._next_once(m - got)
            if o.size <= 1:
                break
            texts.append(t.copy())
            offs.append(o[1:] + offs[-1][-1])
            got += o.size - 1
        return np.concatenate(texts), np.concatenate(offs)

    def __iter__(self):
        while True:
            text, off = self.next()
            if off.size <= 1:
                return
            yield text, off

    def close(self):
        if self._h:
            N.load().bc_reader_close(self._h)
            self._h = N.vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_all(path: str):
    """Every sequence line of a file as a list of str (tests / small inputs)."""
    out = []
    with NativeReader(path, max_reads=1 << 16) as r:
        for text, off in r:
            b = text.tobytes()
            o = off.astype(np.int64)
            out.extend(b[o[i]:o[i + 1]].decode("ascii", errors="replace") for i in range(o.size - 1))
    return out


def record_boundary(path: str, offset: int) -> int:
    """Offset of the first record of an uncompressed reads file that starts at or after `offset`
    (bc_reader_open_range + bc_reader_tell): how nominal byte cuts become record-aligned ranges."""
    size = os.path.getsize(path)
    if offset <= 0:
        return 0
    if offset >= size:
        return size
    with NativeReader(path, max_reads=1, byte_range=(offset, size)) as r:
        return min(r.tell(), size)


class StreamResult:
    """What bc_stream_count_file leaves on the host: read count, undocumented table (packed words sorted,
    counts, smallest keys) and the odd windows' (key, text)."""

    def __init__(self, n_reads, seconds, u_words, u_counts, u_first, odd):
        self.n_reads, self.seconds = n_reads, seconds
        self.u_words, self.u_counts, self.u_first, self.odd = u_words, u_counts, u_first, odd


def stream_count_file(lib_handle, path: str, byte_begin: int, byte_end: int, start: int, L: str, R: str, reverse: bool,
                      max_mismatch: int, algo: str, key_base: int, counts: np.ndarray, first_seen: np.ndarray,
                      n_threads: int = 0, part=None) -> StreamResult:
    """The device path for a single-end file: raw bytes in (a .gz / .zst stream is decoded on the host, 16 MiB at a time,
    into the upload buffers), counts / first-seen (accumulated into the given uint64 arrays), undocumented table and
    odd windows out.  part = (p, n): only the chunks c of the (decoded) stream with c % n == p — how several ranks
    share a compressed file (bc_stream_count_file_part); the byte range must then be the whole file."""
    lib = N.load()
    h = N.vp()
    Lb, Rb = L.encode("ascii"), R.encode("ascii")
    if part is not None:
        N.check(lib.bc_stream_count_file_part(lib_handle, path.encode(), int(part[0]), int(part[1]), int(start), Lb, len(Lb),
                                              Rb, len(Rb), 1 if reverse else 0, int(max_mismatch), N.ALGOS[algo], int(key_base),
                                              int(n_threads), N.ptr(counts), N.ptr(first_seen), C.byref(h)))
    else:
        if path.endswith((".gz", ".zst")):
            byte_begin, byte_end = 0, 2**64 - 1  # a compressed stream has no byte ranges
        N.check(lib.bc_stream_count_file(lib_handle, path.encode(), int(byte_begin), int(byte_end), int(start), Lb, len(Lb),
                                         Rb, len(Rb), 1 if reverse else 0, int(max_mismatch), N.ALGOS[algo], int(key_base),
                                         int(n_threads), N.ptr(counts), N.ptr(first_seen), C.byref(h)))
    try:
        n, nu, no, ob, sec = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double()
        N.check(lib.bc_stream_result_info(h, C.byref(n), C.byref(nu), C.byref(no), C.byref(ob), C.byref(sec)))
        uw, uc, uf = (np.empty(nu.value, np.uint64) for _ in range(3))
        tk = C.c_int()
        N.check(lib.bc_stream_result_text_info(h, C.byref(tk)))
        if tk.value:  # barcodes of more than 32 nt: the undocumented windows come as rows of text (void dtype, one per row)
            rows = np.empty((nu.value, tk.value), np.uint8)
            if nu.value:
                N.check(lib.bc_stream_result_undoc_text(h, N.ptr(rows), N.ptr(uc), N.ptr(uf)))
            uw = rows.view(np.dtype((np.void, tk.value))).reshape(-1)
        elif nu.value:
            N.check(lib.bc_stream_result_undoc(h, N.ptr(uw), N.ptr(uc), N.ptr(uf)))
        odd = []
        if no.value:
            keys = np.empty(no.value, np.uint64)
            off = np.empty(no.value + 1, np.uint64)
            text = np.empty(max(ob.value, 1), np.uint8)
            N.check(lib.bc_stream_result_odd(h, N.ptr(keys), N.ptr(off), N.ptr(text)))
            raw = text.tobytes()
            o = off.astype(np.int64)
            odd = [(int(keys[i]), raw[o[i]:o[i + 1]].decode("ascii", errors="replace")) for i in range(no.value)]
        return StreamResult(int(n.value), float(sec.value), uw, uc, uf, odd)
    finally:
        lib.bc_stream_result_free(h)


def file_count_newlines(path: str, begin: int, end: int):
    """(newlines in [begin, end), file size, last byte of the file) — bc_file_count_newlines."""
    n, size, last = C.c_uint64(), C.c_uint64(), C.c_int()
    N.check(N.load().bc_file_count_newlines(path.encode(), int(begin), int(end), C.byref(n), C.byref(size), C.byref(last)))
    return int(n.value), int(size.value), int(last.value)


def file_seek_newline(path: str, begin: int, nth: int) -> int:
    """Offset of the byte after the nth newline at or after `begin` (nth = 0: `begin`) — bc_file_seek_newline."""
    off = C.c_uint64()
    N.check(N.load().bc_file_seek_newline(path.encode(), int(begin), int(nth), C.byref(off)))
    return int(off.value)


def paired_read_ranges(file_a: str, file_b: str, rank: int, world: int, gather):
    """Byte ranges of two uncompressed mate files that hold the SAME reads, one pair of ranges per rank: every rank
    counts the newlines of one slice of each file, `gather(obj)` (dist.gather_objects) makes the counts common
    knowledge, and a line number becomes a byte offset by a short scan inside the slice that holds it.  Returns
    (index of the rank's first read, (begin, end) in file_a, (begin, end) in file_b).  Files with different numbers
    of reads raise the reference's error (heuristicount.py:484-486) on every rank."""
    import bisect
    info = []
    for path in (file_a, file_b):
        size = os.path.getsize(path)
        cuts = [size * r // world for r in range(world + 1)]
        n, _size, last = file_count_newlines(path, cuts[rank], cuts[rank + 1])
        counts = [p[0] for p in gather((n, last))]
        cum = [0]
        for c in counts:
            cum.append(cum[-1] + c)  # cum[j] = newlines in front of slice j
        lines = cum[-1] + (1 if size and last != 10 else 0)
        per = 4 if path.endswith(".fastq") else 1
        reads = (lines + 2) // 4 if per == 4 else lines  # a FASTQ read = a line with index 1 mod 4
        info.append((path, size, cuts, cum, lines, per, reads))
    if info[0][6] != info[1][6]:
        raise ValueError("Length of reads1 and reads2 must be the same for paired-end data.")
    R = info[0][6]
    r0, r1 = R * rank // world, R * (rank + 1) // world
    ranges = []
    for path, size, cuts, cum, lines, per, _reads in info:
        def line_start(L):
            if L <= 0:
                return 0
            if L > cum[-1]:  # past the last newline: nothing starts there
                return size
            j = bisect.bisect_left(cum, L) - 1  # the L-th newline lies in slice j: cum[j] < L <= cum[j + 1]
            return file_seek_newline(path, cuts[j], L - cum[j])
        ranges.append((line_start(r0 * per), size if rank == world - 1 else line_start(r1 * per)))
    return r0, ranges[0], ranges[1]


def stream_count_pair(lib_handle, path_fwd: str, path_rev: str, start_fwd: int, L_fwd: str, R_fwd: str, start_rev: int,
                      L_rev: str, R_rev: str, max_mismatch: int, algo: str, key_base: int, counts: np.ndarray,
                      first_seen: np.ndarray, n_threads: int = 0, range_fwd=None, range_rev=None) -> StreamResult:
    """The device path for two mate files (bc_stream_count_pair; plain or compressed).  `odd` of the result holds, per
    pair with an odd window, (key, p1, st1, p2, st2, text1, text2): the packed word / status of each mate and the stripped
    text of each odd mate ('' for a packed one — except for barcodes of more than 32 nt, whose word is a hash: then the
    packed mate's barcode as the library spells it).  Such libraries return their undocumented windows as rows of text."""
    lib = N.load()
    h = N.vp()
    enc = [x.encode("ascii") for x in (L_fwd, R_fwd, L_rev, R_rev)]
    fa, fb = range_fwd or (0, 2**64 - 1), range_rev or (0, 2**64 - 1)
    N.check(lib.bc_stream_count_pair_range(lib_handle, path_fwd.encode(), int(fa[0]), int(fa[1]), path_rev.encode(),
                                           int(fb[0]), int(fb[1]), int(start_fwd), enc[0], len(enc[0]), enc[1], len(enc[1]),
                                           int(start_rev), enc[2], len(enc[2]), enc[3], len(enc[3]), int(max_mismatch),
                                           N.ALGOS[algo], int(key_base), int(n_threads), N.ptr(counts), N.ptr(first_seen),
                                           C.byref(h)))
    try:
        n, nu, no, ob, sec = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double()
        N.check(lib.bc_stream_result_info(h, C.byref(n), C.byref(nu), C.byref(no), C.byref(ob), C.byref(sec)))
        uw, uc, uf = (np.empty(nu.value, np.uint64) for _ in range(3))
        tk = C.c_int()
        N.check(lib.bc_stream_result_text_info(h, C.byref(tk)))
        if tk.value:  # barcodes of more than 32 nt: rows of text (void dtype, one per row)
            rows = np.empty((nu.value, tk.value), np.uint8)
            if nu.value:
                N.check(lib.bc_stream_result_undoc_text(h, N.ptr(rows), N.ptr(uc), N.ptr(uf)))
            uw = rows.view(np.dtype((np.void, tk.value))).reshape(-1)
        elif nu.value:
            N.check(lib.bc_stream_result_undoc(h, N.ptr(uw), N.ptr(uc), N.ptr(uf)))
        m, b1, b2 = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(lib.bc_stream_result_pair_info(h, C.byref(m), C.byref(b1), C.byref(b2)))
        odd = []
        if m.value:
            mm = m.value
            keys, p1, p2 = (np.empty(mm, np.uint64) for _ in range(3))
            st1, st2 = np.empty(mm, np.uint8), np.empty(mm, np.uint8)
            off1, off2 = np.empty(mm + 1, np.uint64), np.empty(mm + 1, np.uint64)
            t1, t2 = np.empty(max(b1.value, 1), np.uint8), np.empty(max(b2.value, 1), np.uint8)
            N.check(lib.bc_stream_result_pair_odd(h, N.ptr(keys), N.ptr(p1), N.ptr(st1), N.ptr(p2), N.ptr(st2), N.ptr(off1),
                                                  N.ptr(t1), N.ptr(off2), N.ptr(t2)))
            r1, r2 = t1.tobytes(), t2.tobytes()
            o1, o2 = off1.astype(np.int64), off2.astype(np.int64)
            odd = [(int(keys[i]), int(p1[i]), int(st1[i]), int(p2[i]), int(st2[i]),
                    r1[o1[i]:o1[i + 1]].decode("ascii", errors="replace"), r2[o2[i]:o2[i + 1]].decode("ascii", errors="replace"))
                   for i in range(mm)]
        return StreamResult(int(n.value), float(sec.value), uw, uc, uf, odd)
    finally:
        lib.bc_stream_result_free(h)
