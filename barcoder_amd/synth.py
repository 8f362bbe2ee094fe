"""Seeded synthetic workloads, numpy twin of the
splitmix64 streams in csrc/bc_common.h (bc_synth_reads_dev)."""
from __future__ import annotations

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
LIB_SEED = 0xBA5C0DE5


def sm64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draw(seed: int, i, j: int):
    with np.errstate(over="ignore"):
        return sm64(np.uint64(seed) ^ sm64(np.asarray(i, np.uint64) * np.uint64(8) + np.uint64(j)))


def _kmask(k: int) -> np.uint64:
    m = 0xFFFFFFFF if k >= 32 else (1 << k) - 1
    return np.uint64(m | (m << 32))


def library(G: int, k: int = 20, seed: int = LIB_SEED) -> np.ndarray:
    """G distinct packed k-mers, index = generation order (duplicates rejected)."""
    out = np.empty(0, np.uint64)
    nxt = 0
    while out.size < G:
        need = G - out.size
        cand = draw(seed, np.arange(nxt, nxt + need + 64, dtype=np.uint64), 0) & _kmask(k)
        nxt += need + 64
        allc = np.concatenate([out, cand])
        _, first = np.unique(allc, return_index=True)
        out = allc[np.sort(first)][:G]
    return out


def guide_index(seed: int, i, G: int):
    u0 = draw(seed, i, 0)
    return ((u0 >> np.uint64(32)) * np.uint64(G)) >> np.uint64(32)


def reads(lib: np.ndarray, k: int, seed: int, first: int, n: int, mixture: int) -> np.ndarray:
    """n packed reads starting at global index `first`.
    mixture 0: 90 % library reads, 10 % random; mixture 1: 60/20/10/5 % with
    0/1/2/3 substitutions at distinct positions, 5 % random; mixture 100 + J (read-mix
    sweep): J % random reads, the others with 0/1/2 substitutions in the proportion 6:2:1."""
    i = np.arange(first, first + n, dtype=np.uint64)
    G = lib.size
    g = lib[guide_index(seed, i, G).astype(np.int64)]
    c = (draw(seed, i, 1) & np.uint64(0xFFFF)).astype(np.int64)
    rnd = draw(seed, i, 2) & _kmask(k)
    if mixture == 0:
        return np.where(c < 58982, g, rnd)
    if mixture >= 100:
        T = (65536 * (200 - mixture)) // 100
        nsub = np.select([c < (T * 6) // 9, c < (T * 8) // 9, c < T], [0, 1, 2], default=-1)
    else:
        nsub = np.select([c < 39322, c < 52429, c < 58982, c < 62259], [0, 1, 2, 3], default=-1)
    u3 = draw(seed, i, 3)
    u4 = draw(seed, i, 4)
    r = g.copy()
    used = []  # sorted position arrays chosen so far (per read), as columns
    for s in range(3):
        act = nsub > s
        rr = ((u3 >> np.uint64(16 * s)) & np.uint64(0xFFFF)).astype(np.int64)
        pos = rr % (k - s)
        # map to the pos-th unused position, walking the used ones in ascending order
        if used:
            srt = np.sort(np.stack(used, axis=1), axis=1)
            for t in range(srt.shape[1]):
                pos = pos + (pos >= srt[:, t])
        used.append(pos.copy())
        x = 1 + ((u4 >> np.uint64(16 * s)) & np.uint64(0xFFFF)).astype(np.int64) % 3
        flip = ((x & 1).astype(np.uint64) << pos.astype(np.uint64)) | \
               ((x >> 1).astype(np.uint64) << (pos.astype(np.uint64) + np.uint64(32)))
        r = np.where(act, r ^ flip, r)
    return np.where(nsub < 0, rnd, r)


def genome(length: int, seed: int = 4) -> str:
    """Uniform random ACGT string."""
    i = np.arange((length + 31) // 32, dtype=np.uint64)
    w = draw(seed, i, 0)
    codes = ((w[:, None] >> (np.arange(32, dtype=np.uint64) * np.uint64(2))) & np.uint64(3)).astype(np.uint8)
    return np.frombuffer(b"ACGT", np.uint8)[codes.reshape(-1)[:length]].tobytes().decode()
