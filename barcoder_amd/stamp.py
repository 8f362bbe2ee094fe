"""Source stamp of the match kernels: what `profiles/traffic.json` was measured on and what `bench.py` is running.

A PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) cannot run inside bench.py, so the HBM-traffic figure of the
bench line comes from the tracked profile; the stamp ties that figure to the kernel sources it was measured on, and
bench.py reports a figure whose stamp differs from the tree it runs from as stale instead of quoting it."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
MATCH_SOURCES = ("bc_match.hip", "bc_match_dev.h", "bc_device.h", "bc_common.h", "bc_match_quad.hip", "bc_match_pairs.hip")


def match_source_stamp() -> str:
    h = hashlib.sha256()
    for name in MATCH_SOURCES:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
