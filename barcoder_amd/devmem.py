"""Tiny RAII wrappers over bc_dev_alloc/free and bc_host_alloc_pinned so the package can hold HBM buffers and
page-locked host buffers without torch (torch tensors' data_ptr() work equally well with the C ABI).

Both kinds are pooled: hipMalloc / hipFree / hipHostMalloc cost tens to hundreds of microseconds each (hipFree waits for
the device), which is most of a small call such as pam.find_sites on a bacterial genome.  A freed buffer goes back to a
per-device pool (bounded) and serves the next request of a similar size."""
from __future__ import annotations

import atexit
import ctypes as C

import numpy as np

from . import _native as N

_POOL_MAX_BYTES = 1 << 30        # per kind: what the pool may hold idle
_POOL_MAX_ITEM = 256 << 20       # larger buffers are returned to the runtime at once
_dev_pool: dict[int, list] = {}  # device -> [(cap, ptr value)]
_pin_pool: list = []             # [(cap, ptr value)]
_idle = {"dev": 0, "pin": 0}


def _take(pool: list, nbytes: int, kind: str):
    best = -1
    for i, (cap, _p) in enumerate(pool):
        if nbytes <= cap <= max(2 * nbytes, 1 << 16) and (best < 0 or cap < pool[best][0]):
            best = i
    if best < 0:
        return None
    cap, p = pool.pop(best)
    _idle[kind] -= cap
    return cap, p


def _drain():
    try:
        L = N.load()
    except Exception:
        return
    for dev, pool in _dev_pool.items():
        if pool:
            L.bc_dev_set(dev)
        for _cap, p in pool:
            L.bc_dev_free(N.vp(p))
        pool.clear()
    for _cap, p in _pin_pool:
        L.bc_host_free_pinned(N.vp(p))
    _pin_pool.clear()
    _idle["dev"] = _idle["pin"] = 0


atexit.register(_drain)


class DevBuf:
    def __init__(self, nbytes: int, device: int = 0):
        L = N.load()
        N.check(L.bc_dev_set(device))
        self.nbytes = int(nbytes)
        self.device = device
        want = max(self.nbytes, 16)
        got = _take(_dev_pool.setdefault(device, []), want, "dev")
        if got:
            self._cap, self._p = got[0], N.vp(got[1])
        else:
            self._cap = want if want > (1 << 16) else 1 << 16
            self._p = N.vp()
            N.check(L.bc_dev_alloc(C.byref(self._p), self._cap))

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
            if self._cap <= _POOL_MAX_ITEM and _idle["dev"] + self._cap <= _POOL_MAX_BYTES:
                # (every entry point that filled this buffer has synchronised before returning to Python; the next
                # user's work is ordered behind it on the device anyway)
                _dev_pool.setdefault(self.device, []).append((self._cap, self._p.value))
                _idle["dev"] += self._cap
            else:
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


class PinnedBlock:
    """Page-locked host memory whose numpy views keep it alive: `array()` returns a view, and the block goes back to
    the pool when the last view (and the block object) is gone — results can be handed to the caller without a copy."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        want = max(self.nbytes, 16)
        got = _take(_pin_pool, want, "pin")
        if got:
            self._cap, self._p = got[0], N.vp(got[1])
        else:
            self._cap = want if want > (1 << 16) else 1 << 16
            self._p = N.vp()
            N.check(N.load().bc_host_alloc_pinned(C.byref(self._p), self._cap))

    @property
    def ptr(self):
        return self._p

    def array(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        cbuf = (C.c_char * self._cap).from_address(self._p.value)
        cbuf._owner = self  # the view references the ctypes object, which keeps the block alive (no cycle back)
        return np.frombuffer(cbuf, dtype=dtype, count=count, offset=offset)

    def __del__(self):
        try:
            if self._p:
                if self._cap <= _POOL_MAX_ITEM and _idle["pin"] + self._cap <= _POOL_MAX_BYTES:
                    _pin_pool.append((self._cap, self._p.value))
                    _idle["pin"] += self._cap
                else:
                    N.load().bc_host_free_pinned(self._p)
                self._p = N.vp()
        except Exception:
            pass
