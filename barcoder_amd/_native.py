"""ctypes binding of libbarcoder_hip.so (include/barcoder.h).

The library is the product: there is no Python / CPU fallback for matching or
scanning.  If the shared object is missing this module raises ImportError with
the build command; if it loads but no GPU is visible, the device entry points
return BC_ENODEV and ``check`` raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

# torch (when present) must load ITS libamdhip64 before ours is resolved, so
# that one HIP runtime serves both (same SONAME libamdhip64.so.7).
try:  # pragma: no cover - trivial
    import torch  # noqa: F401
except Exception:  # torch is plumbing only; the C ABI works without it
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# BC_NATIVE_SO: another build of the same library (the host-sanitizer build of tests/README: build/libbarcoder_hip_asan.so)
SO_PATH = os.environ.get("BC_NATIVE_SO") or os.path.join(_HERE, "libbarcoder_hip.so")
CSRC = os.path.join(_HERE, "csrc")

BC_OK, BC_EINVAL, BC_ENOMEM, BC_EHIP, BC_ENODEV, BC_ECAP = 0, -1, -2, -3, -4, -5
IDX_MISS, IDX_SKIP, DIST_NONE = -1, -2, 255
SKIP_BIT = np.uint64(1 << 63)
ALGO_AUTO, ALGO_HASH, ALGO_SEED, ALGO_BRUTE, ALGO_SEED_CSR = 0, 1, 2, 3, 4
ALGOS = {"auto": ALGO_AUTO, "hash": ALGO_HASH, "seed": ALGO_SEED, "brute": ALGO_BRUTE, "seed_csr": ALGO_SEED_CSR}
WIN_OK, WIN_SKIP, WIN_ODD = 0, 1, 2

vp = C.c_void_p


HIT_DTYPE = np.dtype([("query", np.uint32), ("guide", np.uint32), ("dist", np.uint32)])

# name -> (restype, argtypes); must list every symbol of include/barcoder.h
SIGNATURES = {
    "bc_version": (C.c_int, []),
    "bc_last_error": (C.c_char_p, []),
    "bc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bc_pack_kmers": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_size_t, vp, vp]),
    "bc_unpack_kmers": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "bc_revcomp_kmers": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "bc_pack_windows": (C.c_int, [vp, vp, C.c_size_t, C.c_int64, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, vp, vp]),
    "bc_hash_kmers": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "bc_hash_windows": (C.c_int, [vp, vp, C.c_size_t, C.c_int64, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "bc_reader_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "bc_reader_open_range": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "bc_reader_tell": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "bc_reader_next": (C.c_int, [vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]),
    "bc_reader_close": (C.c_int, [vp]),
    "bc_stream_count_file": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_char_p, C.c_int, C.c_char_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp, vp, C.POINTER(vp)]),
    "bc_stream_count_file_part": (C.c_int, [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_char_p, C.c_int, C.c_char_p,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp, vp, C.POINTER(vp)]),
    "bc_reader_read_raw": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bc_stream_result_info": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "bc_stream_result_undoc": (C.c_int, [vp, vp, vp, vp]),
    "bc_stream_result_odd": (C.c_int, [vp, vp, vp, vp]),
    "bc_stream_result_free": (C.c_int, [vp]),
    "bc_stream_count_pair": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                       C.c_int64, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                       C.c_int, vp, vp, C.POINTER(vp)]),
    "bc_stream_count_pair_range": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64,
                                             C.c_int64, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, C.c_char_p, C.c_int,
                                             C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp, vp, C.POINTER(vp)]),
    "bc_file_count_newlines": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_int)]),
    "bc_file_seek_newline": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bc_stream_result_pair_info": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bc_stream_result_pair_odd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bc_sample_data": (C.c_int, [C.c_char_p, C.c_char_p, vp, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "bc_sample_info": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "bc_sample_flanks": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp, C.POINTER(C.c_int), vp,
                                   C.POINTER(C.c_int)]),
    "bc_sample_free": (C.c_int, [vp]),
    "bc_lib_create": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]),
    "bc_lib_destroy": (C.c_int, [vp]),
    "bc_lib_info": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bc_lib_last_kernel": (C.c_char_p, [vp]),
    "bc_lib_release_stream": (C.c_int, [vp, vp]),
    "bc_lib_workspace_count": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "bc_pcie_probe": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    "bc_lib_attach_text": (C.c_int, [vp, vp, C.c_int]),
    "bc_stream_result_text_info": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "bc_stream_result_undoc_text": (C.c_int, [vp, vp, vp, vp]),
    "bc_multi_create": (C.c_int, [vp, C.c_uint32, C.c_int, vp, C.c_int, C.POINTER(vp)]),
    "bc_multi_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bc_multi_gpu_match": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint64]),
    "bc_multi_destroy": (C.c_int, [vp]),
    "bc_match_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "bc_match_host": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint64]),
    "bc_match_all_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp]),
    "bc_match_all_pam_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, vp, C.c_int64, C.c_uint64, C.c_uint64, vp,
                                       C.c_uint64, vp, vp]),
    "bc_genome_pack": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp]),
    "bc_pam_scan_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int, vp, vp, vp]),
    "bc_pam_scan_host": (C.c_int, [C.c_int, vp, vp, C.c_uint64, vp, C.c_int, vp, vp]),
    "bc_pam_scan_time_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int, vp, vp, C.c_int, vp, C.POINTER(C.c_float)]),
    "bc_pam_sites_count_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "bc_pam_sites_emit_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_uint64, vp, vp]),
    "bc_sites_expand_dev": (C.c_int, [vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp]),
    "bc_windows_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int64, C.c_int, C.c_int, vp, vp, vp]),
    "bc_score_mismatch_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp]),
    "bc_synth_reads_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, vp, vp]),
    "bc_dev_set": (C.c_int, [C.c_int]),
    "bc_dev_alloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "bc_dev_free": (C.c_int, [vp]),
    "bc_tally_export_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
    "bc_dev_memset": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
    "bc_dev_h2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "bc_dev_d2h": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "bc_dev_sync": (C.c_int, [vp]),
    "bc_host_alloc_pinned": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "bc_host_free_pinned": (C.c_int, [vp]),
    "bc_match_time_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp,
                                    C.POINTER(C.c_float)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libbarcoder_hip.so in-tree with hipcc for gfx950."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbarcoder_hip.so")
    cmd = ["make", "-C", CSRC, f"HIPCC={hipcc}"]
    if force:
        cmd.insert(1, "-B")
    subprocess.check_call(cmd)
    return SO_PATH


def load():
    """Load the shared library (once) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing. Build it with `make -C {CSRC}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "barcoder_amd has no CPU fallback.")
    L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class BarcoderNativeError(RuntimeError):
    pass


def check(rc: int):
    """Map a status code to the exception the reference's Python code would see."""
    if rc == BC_OK:
        return
    msg = (load().bc_last_error() or b"").decode(errors="replace")
    if rc == BC_EINVAL:
        raise ValueError(msg)
    if rc == BC_ENOMEM:
        raise MemoryError(msg)
    if rc == BC_ENODEV:
        raise BarcoderNativeError(f"no MI355X/HIP device available: {msg}")
    raise BarcoderNativeError(f"libbarcoder_hip error {rc}: {msg}")


def ptr(a):
    """void* of a numpy array (must be C-contiguous) or None."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(vp)


def device_count() -> int:
    n = C.c_int(0)
    rc = load().bc_device_count(C.byref(n))
    return n.value if rc == BC_OK else 0
