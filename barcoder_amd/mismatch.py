"""Mismatch-efficacy model with the call shapes of the reference's mismatch.py.

Mirrors (reference file:line):
  gc_content               mismatch.py:10-12
  calculate_y_pred         mismatch.py:15-35   (float64: intercept + per-position + per-substitution + GC term)
  read_parameters          mismatch.py:38-46
  find_closest_mismatch    mismatch.py:55-66
  generate_mismatches      mismatch.py:81-111  (all 3k single-base variants, closest to each desired score)
  print_mismatches         mismatch.py:69-78   (TSV: original, variant, change, y_pred %.4f)

`y_pred_from_mask` scores a (guide, hit) pair from the matcher's mismatch
position mask instead of a second string compare."""
from __future__ import annotations

import csv
import sys

import numpy as np


def gc_content(seq):
    return (seq.count("G") + seq.count("C")) / len(seq)


def read_parameters(file_path):
    try:
        with open(file_path, newline="") as fh:
            return {row["feature"]: float(row["weight"]) for row in csv.DictReader(fh)}
    except Exception as e:  # the reference logs and exits (mismatch.py:44-46)
        print(f"Error reading parameters file: {e}", file=sys.stderr)
        sys.exit(1)


def calculate_y_pred(original, variant, gc_weight, params):
    if original is None or variant is None or original is np.nan or variant is np.nan:
        return None
    if original == variant or len(original) != len(variant):
        return None
    y = params["intercept"]
    for pos, (o, v) in enumerate(zip(original, variant)):
        if o != v:
            y += params[f"{pos}"]
            y += params[f"{o}{v}"]
    return y + gc_weight * gc_content(original)


def y_pred_from_mask(original, variant, mask: int, params):
    """Same value, visiting only the positions set in `mask` (bit i = base i differs)."""
    if mask == 0 or len(original) != len(variant):
        return None
    y = params["intercept"]
    pos = 0
    while mask:
        if mask & 1:
            y += params[f"{pos}"]
            y += params[f"{original[pos]}{variant[pos]}"]
        mask >>= 1
        pos += 1
    return y + params["GC_content"] * gc_content(original)


def find_closest_mismatch(score, mismatches, mismatch_list):
    taken = {m for m, _ in mismatch_list}
    best, best_score = None, None
    for m, s in mismatches:
        if best_score is None or abs(s - score) < abs(best_score - score):
            if m not in taken:
                best, best_score = m, s
    return best, best_score


def mismatch_rows(spacer, min_score, max_score, step, parameters, spacer_original=None):
    """Rows (original, variant, change, y_pred) the reference prints for one spacer."""
    spacer_original = spacer if spacer_original is None else spacer_original
    scored = []
    for pos in range(len(spacer)):
        for nt in "ACGT":
            if nt == spacer[pos]:
                continue
            variant = spacer[:pos] + nt + spacer[pos + 1:]
            scored.append(((pos, nt), calculate_y_pred(spacer, variant, parameters["GC_content"], parameters)))
    chosen = []
    for score in np.arange(min_score, max_score + step, step):
        m, s = find_closest_mismatch(score, scored, chosen)
        if m is not None:
            chosen.append((m, s))
    rows = []
    for (pos, nt), s in chosen:
        variant = spacer_original[:pos] + nt + spacer_original[pos + 1:]
        rows.append((spacer_original, variant, f"{spacer_original[pos]}{pos + 1}{nt}", f"{s:.4f}"))
    return rows


def generate_mismatches(spacers, min_score, max_score, step, parameters, spacer_original):
    """Prints like the reference: the table of the LAST spacer in `spacers`
    (mismatch.py:108-111 prints after the loop)."""
    rows = []
    for spacer in spacers:
        rows = mismatch_rows(spacer, min_score, max_score, step, parameters, spacer_original)
    for r in rows:
        print("\t".join(r))


def generate_header():
    print("\t".join(["original", "variant", "change_description", "y_pred"]))


# ---------------------------------------------------------------------------- device scoring
def weights_vector(params: dict) -> np.ndarray:
    """The reference's {feature: weight} dict in the layout of bc_score_mismatch_dev
    (missing features weigh 0, like an absent CSV row would fail in the reference)."""
    w = np.zeros(50, np.float64)
    w[0] = params["intercept"]
    for p in range(32):
        w[1 + p] = params.get(f"{p}", 0.0)
    for o, a in enumerate("ACGT"):
        for v, b in enumerate("ACGT"):
            w[33 + 4 * o + v] = params.get(f"{a}{b}", 0.0)
    w[49] = params["GC_content"]
    return w


def y_pred_device(originals, variants, params: dict, device: int = 0):
    """calculate_y_pred for many equal-length pairs on the GPU; list of float / None."""
    from . import _native as N
    from .devmem import DevBuf
    from .encoding import pack_kmers
    originals, variants = list(originals), list(variants)
    if not originals:
        return []
    k = len(originals[0])
    po, fo = pack_kmers(originals, k)
    pv, fv = pack_kmers(variants, k)
    if fo.any() or fv.any():
        raise ValueError("sequences must be upper-case ACGT")
    n = po.size
    L = N.load()
    N.check(L.bc_dev_set(device))
    with DevBuf.from_numpy(po, device) as d_o, DevBuf.from_numpy(pv, device) as d_v, \
            DevBuf.from_numpy(weights_vector(params), device) as d_w, DevBuf(n * 8, device) as d_y, \
            DevBuf(n, device) as d_n:
        N.check(L.bc_score_mismatch_dev(d_o.ptr, d_v.ptr, n, k, d_w.ptr, d_y.ptr, d_n.ptr, None))
        y = d_y.to_numpy(np.float64, n)
        none = d_n.to_numpy(np.uint8, n)
    return [None if z else float(v) for v, z in zip(y, none)]
