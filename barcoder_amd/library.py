"""Barcode / guide library loaders with the reference's API (BarCodeLibrary.py)
plus the device-resident form the HIP kernels search.

Mirrors (reference file:line):
  BarCodeLibraryReader  BarCodeLibrary.py:14-45
  BarCodeLibrary        BarCodeLibrary.py:48-93
  BarCodeLibraryError   BarCodeLibrary.py:96-102
"""
from __future__ import annotations

import csv
import ctypes as C
import logging
import os

import numpy as np

from . import _native as N
from .encoding import pack_kmers

log = logging.getLogger("barcoder_amd")


class BarCodeLibraryError(Exception):
    """Raised for errors in the BarCodeLibrary (BarCodeLibrary.py:96-102)."""

    def __init__(self, message):
        super().__init__(message)
        self.message = message


def read_fasta_records(path: str):
    """Minimal FASTA reader: one string per record, sequence lines joined
    (what Bio.SeqIO.parse(..., "fasta") yields as str(record.seq),
    BarCodeLibrary.py:27-31; Biopython itself is not a dependency here)."""
    seqs, cur, seen = [], [], False
    with open(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if seen:
                    seqs.append("".join(cur))
                cur, seen = [], True
            elif seen:
                cur.append(line.strip())
    if seen:
        seqs.append("".join(cur))
    return seqs


class BarCodeLibraryReader:
    def __init__(self, filename, column=None):
        self.filename = filename
        self.column = column

    def read_barcodes(self):
        if self.filename.endswith(".fasta"):
            return self._read_fasta()
        if self.filename.endswith(".tsv"):
            return self._read_tsv()
        raise ValueError(f"Unsupported file format: {self.filename}")  # BarCodeLibrary.py:25

    def _read_fasta(self):
        return read_fasta_records(self.filename)

    def _read_tsv(self):
        if self.column is None:
            raise ValueError("A barcode column must be specified for TSV files")  # :35
        with open(self.filename, "r", newline="") as fh:
            reader = csv.reader(fh, delimiter="\t")
            header = next(reader)
            if self.column not in header:
                raise ValueError(f"Column '{self.column}' not found in file")  # :41
            at = header.index(self.column)
            return [row[at] for row in reader]


class BarCodeLibrary:
    """Same constructor and methods as the reference class.  ``barcodes`` is a
    set, as there; ``ordered`` additionally keeps first-occurrence order, which
    defines the guide index used on the device ("Library
    order").  Unlike the reference, ``size`` is never stale (its lru_cache
    quirk, BarCodeLibrary.py:90-93, is deliberately not reproduced)."""

    def __init__(self, filename=None, barcodes=None, **kwargs):
        self._barcodes = {}
        self.kwargs = kwargs
        self._device = {}
        if filename is not None:
            self.reader = BarCodeLibraryReader(filename, **self.kwargs)
            self.load()
        if barcodes is not None:
            self.load_from_list(barcodes)

    # -- reference API ---------------------------------------------------------
    @property
    def barcodes(self):
        return set(self._barcodes)

    @property
    def ordered(self):
        return list(self._barcodes)

    def add(self, sequence):
        self._barcodes.setdefault(sequence, None)
        self._device.clear()

    def remove(self, sequence):
        del self._barcodes[sequence]  # KeyError like set.remove
        self._device.clear()

    def load(self):
        log.warning("Genome circularity is not yet implemented. Barcodes spanning the origin will be missed!")
        try:
            for sequence in self.reader.read_barcodes():
                self.add(sequence)
            log.info("Loaded %d barcodes from %s ...", self.size, os.path.abspath(self.reader.filename))
        except Exception as e:
            raise BarCodeLibraryError("Failed to load barcodes") from e  # :82-83

    def load_from_list(self, barcodes):
        for sequence in barcodes:
            self.add(sequence)
        log.info("Loaded %d barcodes from list ...", self.size)

    @property
    def size(self):
        return len(self._barcodes)

    # -- device form -----------------------------------------------------------
    def to_device(self, device: int = 0) -> "DeviceLibrary":
        if device not in self._device:
            self._device[device] = DeviceLibrary.from_strings(self.ordered, device)
        return self._device[device]


class DeviceLibrary:
    """Owner of one ``bc_lib`` (guides, exact-match table and seed indices in HBM)."""

    def __init__(self, packed: np.ndarray, k: int, device: int = 0, labels=None):
        packed = np.ascontiguousarray(packed, np.uint64)
        self.packed, self.k, self.device = packed, int(k), int(device)
        self.labels = labels
        self.handle = N.vp()
        N.check(N.load().bc_lib_create(N.ptr(packed), packed.size, self.k, self.device, C.byref(self.handle)))

    @classmethod
    def from_strings(cls, seqs, device: int = 0):
        """De-duplicate keeping first occurrence; index = position in that order."""
        seqs = list(dict.fromkeys(seqs))
        if not seqs:
            raise ValueError("empty barcode library")
        k = len(seqs[0])
        if k > 32:  # exact match through a 64-bit text hash, hits verified against the text (LongDeviceLibrary)
            return LongDeviceLibrary(seqs, device)
        packed, flags = pack_kmers(seqs, k)
        if flags.any():
            bad = seqs[int(np.flatnonzero(flags)[0])]
            raise ValueError(f"library entry {bad!r} is not pure upper-case ACGT")
        return cls(packed, k, device, labels=seqs)

    @property
    def size(self) -> int:
        return int(self.packed.size)

    long = False  # LongDeviceLibrary: barcodes of more than 32 nt

    @property
    def last_kernel(self) -> str:
        """Name of the match kernel the last call on this library launched (bc_lib_last_kernel)."""
        return (N.load().bc_lib_last_kernel(self.handle) or b"").decode()

    def close(self):
        if self.handle:
            N.load().bc_lib_destroy(self.handle)
            self.handle = N.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LongDeviceLibrary(DeviceLibrary):
    """Barcodes of 33 ... 1000 nt (heuristicount.py:75-97 admits them; its set lookup, :449-453, does not care about
    length).  The device holds the 64-bit hash of every barcode as a pseudo 32-mer — exact table only — and the caller
    verifies every hit against `label_bytes`, so a hash collision can cost a lookup, never a wrong count."""
    long = True

    def __init__(self, seqs, device: int = 0):
        from .encoding import hash_kmers
        seqs = list(seqs)
        k = len(seqs[0])
        if k > 1000:
            raise ValueError(f"barcode length {k} > 1000")  # heuristicount.py:88-94
        blob = "".join(seqs).encode("ascii", errors="replace")
        self.label_bytes = np.frombuffer(blob, np.uint8).reshape(len(seqs), k)
        if not np.isin(self.label_bytes, np.frombuffer(b"ACGT", np.uint8)).all():
            bad = next(s for s in seqs if not set(s) <= set("ACGT"))
            raise ValueError(f"library entry {bad!r} is not pure upper-case ACGT")
        hashes = hash_kmers(seqs, k)
        if np.unique(hashes).size != hashes.size:
            raise ValueError("two barcodes share a 64-bit hash (a 2^-64 event per pair): cannot index this library")
        super().__init__(hashes, 32, device, labels=seqs)
        self.text_k = k
        # the texts next to the hashes in HBM: the device ingest path (bc_stream_count_file) verifies its own hits
        N.check(N.load().bc_lib_attach_text(self.handle, N.ptr(np.ascontiguousarray(self.label_bytes)), k))
