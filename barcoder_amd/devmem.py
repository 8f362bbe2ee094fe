"""Tiny RAII wrappers over bc_dev_alloc/free and bc_host_alloc_pinned so the package can hold HBM buffers and
page-locked host buffers without torch (torch tensors' data_ptr() work equally well with the C ABI).

Both kinds are pooled: hipMalloc / hipFree / hipHostMalloc cost tens to hundreds of microseconds each (hipFree waits for
the device), which is most of a small call such as pam.find_sites on a bacterial genome.  A freed buffer goes back to a
per-device pool (bounded) and serves the next request of a similar size."""
from __future__ import annotations

import atexit
import ctypes as C
import threading

import numpy as np

from . import _native as N

_POOL_MAX_BYTES = 1 << 30        # per kind: what the pool may hold idle
_POOL_MAX_ITEM = 256 << 20       # larger buffers are returned to the runtime at once
_dev_pool: dict[int, list] = {}  # device -> [(cap, ptr value)]
_pin_pool: list = []             # [(cap, ptr value)]
_idle = {"dev": 0, "pin": 0}
_lock = threading.RLock()        # the pools and _idle (a PinnedBlock can be finalised on any thread)


def _take(pool: list, nbytes: int, kind: str):
    with _lock:
        best = -1
        for i, (cap, _p) in enumerate(pool):
            if nbytes <= cap <= max(2 * nbytes, 1 << 16) and (best < 0 or cap < pool[best][0]):
                best = i
        if best < 0:
            return None
        cap, p = pool.pop(best)
        _idle[kind] -= cap
        return cap, p


def _give(pool: list, cap: int, p: int, kind: str) -> bool:
    """Back to the pool if it has room; False = the caller returns the buffer to the runtime."""
    with _lock:
        if cap <= _POOL_MAX_ITEM and _idle[kind] + cap <= _POOL_MAX_BYTES:
            pool.append((cap, p))
            _idle[kind] += cap
            return True
    return False


def drain():
    """Return every idle pooled buffer (HBM and page-locked host memory) to the runtime.  The C side allocates on
    its own (stream slots, cascade workspace, window stores): callers that are about to start a large job —
    `heuristicount.count_barcodes` before its bulk pass — drain first, and an allocation that fails here drains and
    retries once, so a job near the HBM limit does not fail on memory this package is only holding idle."""
    try:
        L = N.load()
    except Exception:
        return
    with _lock:
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


_drain = drain
atexit.register(drain)


def _alloc_retry(fn, out, cap):
    """fn(byref(out), cap); on failure drain the pools and try once more"""
    rc = fn(C.byref(out), cap)
    if rc != 0:
        drain()
        rc = fn(C.byref(out), cap)
    N.check(rc)


class DevBuf:
    def __init__(self, nbytes: int, device: int = 0):
        L = N.load()
        N.check(L.bc_dev_set(device))
        self.nbytes = int(nbytes)
        self.device = device
        want = max(self.nbytes, 16)
        with _lock:
            pool = _dev_pool.setdefault(device, [])
        got = _take(pool, want, "dev")
        if got:
            self._cap, self._p = got[0], N.vp(got[1])
        else:
            self._cap = want if want > (1 << 16) else 1 << 16
            self._p = N.vp()
            _alloc_retry(L.bc_dev_alloc, self._p, self._cap)

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
            # (every entry point that filled this buffer has synchronised before returning to Python; the next
            # user's work is ordered behind it on the device anyway)
            with _lock:
                pool = _dev_pool.setdefault(self.device, [])
            if not _give(pool, self._cap, self._p.value, "dev"):
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
            _alloc_retry(N.load().bc_host_alloc_pinned, self._p, self._cap)

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
                if not _give(_pin_pool, self._cap, self._p.value, "pin"):
                    N.load().bc_host_free_pinned(self._p)
                self._p = N.vp()
        except Exception:
            pass
