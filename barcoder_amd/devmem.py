"""Tiny RAII wrapper over bc_dev_alloc/free so the package can hold HBM buffers
without torch (torch tensors' data_ptr() work equally well with the C ABI)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class DevBuf:
    def __init__(self, nbytes: int, device: int = 0):
        L = N.load()
        N.check(L.bc_dev_set(device))
        self.nbytes = int(nbytes)
        self.device = device
        self._p = N.vp()
        N.check(L.bc_dev_alloc(C.byref(self._p), max(self.nbytes, 16)))

    @property
    def ptr(self):
        return self._p

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = 0) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device)
        if a.nbytes:
            N.check(N.load().bc_dev_h2d(b._p, N.ptr(a), a.nbytes, None))
        return b

    def zero(self):
        N.check(N.load().bc_dev_memset(self._p, 0, self.nbytes, None))
        return self

    def to_numpy(self, dtype, count: int | None = None) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(n, dtype)
        if n:
            N.check(N.load().bc_dev_d2h(N.ptr(out), self._p, n * dtype.itemsize, None))
        return out

    def free(self):
        if self._p:
            N.load().bc_dev_free(self._p)
            self._p = N.vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
