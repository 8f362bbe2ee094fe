#!/usr/bin/env python3
"""PCIe-inclusive rate of bc_match_host (host buffers in, host results out): the
number DESIGN.md quotes next to the HBM-resident `value` of bench.py."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import DeviceLibrary, GpuMatcher, synth as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
lib = S.library(100_000, 20)
m = GpuMatcher(DeviceLibrary(lib, 20, 0))
reads = np.concatenate([S.reads(lib, 20, 2, a, min(10_000_000, n - a), 1) for a in range(0, n, 10_000_000)])
out = []
for d, algo in ((2, "seed"), (0, "hash")):
    m.match(reads[:1_000_000], None, d, algo)
    t0 = time.perf_counter()
    res = m.match(reads, None, d, algo)
    dt = time.perf_counter() - t0
    out.append({"max_mismatch": d, "reads": n, "seconds": round(dt, 3), "Mreads_per_s": round(n / dt / 1e6, 1),
                "GB_per_s_over_pcie": round(n * 14 / dt / 1e9, 2), "matched": int((res.best_idx >= 0).sum())})
print(json.dumps(out))
