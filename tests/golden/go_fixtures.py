"""Python restatement of the *input fixtures* the reference's container tests use.

The ten named patterns x three encodings come from roaring/roaring_helpers_test.go:12-306
(arrayEmpty ... runEvenBitsSet, setupContainerTests :257); the helper calls that appear
inside the table literals (MakeBitmap, MakeLastBitSet, getFullBitmap, make, bitmapXxx(),
NewContainerArray/Run) are resolved by `resolve()`.
"""
from __future__ import annotations

import numpy as np

W = 1 << 16
FULL = 0xFFFFFFFFFFFFFFFF

PATTERNS = ["empty", "full", "firstBitSet", "lastBitSet", "firstBitUnset", "lastBitUnset", "innerBitsSet", "outerBitsSet", "oddBitsSet", "evenBitsSet"]


def pattern_values(name: str) -> np.ndarray:
    """Set bit positions of a named pattern (roaring_helpers_test.go:12-79)."""
    allv = np.arange(W, dtype=np.int64)
    return {
        "empty": allv[:0],
        "full": allv,
        "firstBitSet": allv[:1],
        "lastBitSet": allv[-1:],
        "firstBitUnset": allv[1:],
        "lastBitUnset": allv[:-1],
        "innerBitsSet": allv[1:-1],
        "outerBitsSet": allv[[0, W - 1]],
        "oddBitsSet": allv[1::2],
        "evenBitsSet": allv[0::2],
    }[name]


def pattern_words(name: str) -> np.ndarray:
    bits = np.zeros(W, dtype=np.uint8)
    bits[pattern_values(name)] = 1
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def pattern_runs(name: str):
    """Run encodings as the reference writes them (roaring_helpers_test.go:166-229):
    maximal runs; odd/even patterns are 32768 single-bit runs."""
    v = pattern_values(name)
    if v.size == 0:
        return []
    breaks = np.nonzero(np.diff(v) != 1)[0]
    starts = np.concatenate([[v[0]], v[breaks + 1]])
    lasts = np.concatenate([v[breaks], [v[-1]]])
    return list(zip(starts.tolist(), lasts.tolist()))


def pad_words(words) -> np.ndarray:
    w = np.zeros(1024, dtype=np.uint64)
    src = np.asarray(words, dtype=np.uint64)
    w[: src.size] = src
    return w


def runs_of(lst):
    """[{Start,Last}] or [[s,l]] -> [(s,l)]"""
    out = []
    for r in lst or []:
        out.append((r["Start"], r["Last"]) if isinstance(r, dict) else (r[0], r[1]))
    return out


def resolve(v):
    """Resolve a symbolic helper call recorded by extract_go_tables.py to plain data."""
    if not (isinstance(v, dict) and "$call" in v):
        return v
    name, args = v["$call"], [resolve(a) for a in v["args"]]
    if name == "MakeBitmap":  # roaring_internal_test.go:1640
        return pad_words(args[0]).tolist()
    if name == "MakeLastBitSet":  # :1645
        return pattern_words("lastBitSet").tolist()
    if name == "getFullBitmap":
        return [FULL] * 1024
    if name == "make":
        return [0] * int(args[0])
    if name.startswith("bitmap") and name[6:7].isupper():  # bitmapFull() etc, helpers :81-164
        pat = name[6].lower() + name[7:]
        return pattern_words(pat).tolist()
    if name == "NewContainerArray":
        return {"type": "array", "data": args[0]}
    if name == "NewContainerRun":
        return {"type": "run", "data": runs_of(args[0])}
    raise KeyError(name)
