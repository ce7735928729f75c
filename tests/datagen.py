"""Seeded synthetic inputs shared by tests, smoke() and bench.py.

Shapes follow the reference's own generators where they exist: the container archetypes
of roaring/container_archetypes.go:67-124 (Run16..Run1024, RunSplit, RunFull, Ary*,
Bitmap*) and the rank-law row densities of fragment_internal_test.go:2786-2834.  The RNG
streams themselves (Go math/rand, apophenia) are not reproducible without Go, so seeds
here are our own (numpy PCG64 seeded from SEED + index).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

SEED = 0x5EED0001
SLOTS = 16
WORDS = 1024


def rng_for(*idx: int) -> np.random.Generator:
    return np.random.default_rng([SEED, *idx])


# ---- container shapes (values as sorted int arrays) -----------------------------------
def vals_random(rng, n: int) -> np.ndarray:
    return np.sort(rng.choice(65536, size=n, replace=False)).astype(np.int64)


def vals_runs(rng, n_runs: int, fill: float = 0.5) -> np.ndarray:
    """n_runs runs spread over the container (archetypes Run16..Run1024)."""
    period = 65536 // n_runs
    starts = np.arange(n_runs) * period + rng.integers(0, max(1, period // 4), n_runs)
    lens = np.maximum(1, (period * fill * rng.uniform(0.5, 1.0, n_runs)).astype(np.int64))
    lens = np.minimum(lens, period - (starts - np.arange(n_runs) * period) - 1)
    lens = np.maximum(lens, 1)
    return np.concatenate([np.arange(s, s + l) for s, l in zip(starts, lens)]).astype(np.int64)


def vals_density(rng, p: float) -> np.ndarray:
    return np.nonzero(rng.random(65536) < p)[0].astype(np.int64)


def words_of(vals: np.ndarray) -> np.ndarray:
    bits = np.zeros(65536, dtype=np.uint8)
    bits[vals] = 1
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def runs_of_vals(vals: np.ndarray):
    if vals.size == 0:
        return []
    br = np.nonzero(np.diff(vals) != 1)[0]
    return list(zip(np.concatenate([[vals[0]], vals[br + 1]]).tolist(), np.concatenate([vals[br], [vals[-1]]]).tolist()))


KINDS = ["array_small", "array_big", "bitmap_sparse", "bitmap_dense", "run_few", "run_many", "run_full", "run_split", "array_dense_runs", "bitmap_as_array_range", "empty_array", "single"]


def oracle_container(rng, kind: str):
    """An oracle container (oracle.pyoracle.OContainer) of a named shape.  The encoding is
    fixed by the kind, NOT by optimize(): kernels must accept any legal encoding, e.g.
    arrays longer than 4096 (roaring.go:5054) and bitmaps with few bits (roaring.go:4976)."""
    from oracle import pyoracle as O

    if kind == "array_small":
        return O.OContainer.array(vals_random(rng, int(rng.integers(1, 64))))
    if kind == "array_big":
        return O.OContainer.array(vals_random(rng, int(rng.integers(1000, 4096))))
    if kind == "array_dense_runs":  # an array that would optimize() to runs; > 4096 long
        return O.OContainer.array(vals_runs(rng, 32, 0.6))
    if kind == "bitmap_sparse":  # intersectBitmapBitmap never down-converts
        return O.OContainer.bitmap(words_of(vals_random(rng, int(rng.integers(1, 300)))))
    if kind == "bitmap_dense":
        return O.OContainer.bitmap(words_of(vals_density(rng, float(rng.uniform(0.2, 0.9)))))
    if kind == "bitmap_as_array_range":
        return O.OContainer.bitmap(words_of(vals_density(rng, 0.06)))  # n ~ 3900..4100
    if kind == "run_few":
        return O.OContainer.run(runs_of_vals(vals_runs(rng, int(rng.choice([1, 2, 16, 32])), 0.7)))
    if kind == "run_many":
        return O.OContainer.run(runs_of_vals(vals_runs(rng, int(rng.choice([128, 1024, 2048])), 0.5)))
    if kind == "run_full":
        return O.OContainer.run([(0, 65535)])
    if kind == "run_split":  # archetype RunSplit: everything but a hole in the middle
        h = int(rng.integers(1, 65534))
        return O.OContainer.run([(0, h - 1), (h + 1, 65535)])
    if kind == "empty_array":
        return O.OContainer.array([])
    if kind == "single":
        v = int(rng.choice([0, 63, 64, 65535, int(rng.integers(0, 65536))]))
        t = int(rng.integers(0, 3))
        return [O.OContainer.array([v]), O.OContainer.bitmap(words_of(np.array([v]))), O.OContainer.run([(v, v)])][t]
    raise ValueError(kind)


def random_row(rng, row_id: int, p_missing: float = 0.15) -> Dict[int, object]:
    """One shard row: container key (row_id*16 + slot) -> oracle container, a mix of every
    encoding, with some slots missing (nil containers, containers_slice.go:238)."""
    row = {}
    for s in range(SLOTS):
        if rng.random() < p_missing:
            continue
        row[row_id * SLOTS + s] = oracle_container(rng, KINDS[int(rng.integers(0, len(KINDS)))])
    return row


def to_fbk(c):
    """oracle container -> featurebase_amd.roaring.Container (same encoding, same bytes)."""
    from featurebase_amd.roaring import Container
    from oracle import pyoracle as O

    if c.typ == O.ARRAY:
        return Container.array(c.data())
    if c.typ == O.BITMAP:
        return Container.bitmap(c.data(), c.n)
    return Container.run([tuple(x) for x in c.data().tolist()], c.n)


def to_fbk_row(row) -> Dict[int, object]:
    return {k: to_fbk(c) for k, c in row.items()}


# ---- BASELINE.json configs ---------------------------------------------------------------
def dense_rows(n_rows: int, p: float, seed_idx: int) -> np.ndarray:
    """n_rows x 16 x 1024 uint64, each bit set i.i.d. with probability p (config 1: p=0.10,
    config 2: p=0.50; all containers are bitmaps after optimize(): N >= 4096)."""
    rng = rng_for(seed_idx)
    if p == 0.5:
        return rng.integers(0, 2**64, (n_rows, SLOTS, WORDS), dtype=np.uint64)
    out = np.zeros((n_rows, SLOTS, WORDS), dtype=np.uint64)
    # build from 8 independent uniform bytes per bit would be slow; compose probabilities
    # from AND/OR of fair words instead (exact for p = k/256)
    k = int(round(p * 256))
    acc = np.zeros_like(out)
    for bit in range(8):  # p = sum b_i 2^-(i+1): standard bit-serial Bernoulli construction
        r = rng.integers(0, 2**64, out.shape, dtype=np.uint64)
        if (k >> bit) & 1:
            acc = acc | r
        else:
            acc = acc & r
    return acc


def zipf_density(r: int) -> float:
    """Row density rank law of config 3: clamp(0.5*(r+1)^-1.1, 0.001, 0.5)."""
    return float(min(0.5, max(0.001, 0.5 * (r + 1) ** -1.1)))


def mixed_vals_for_density(rng, d: float, run_structured: bool) -> np.ndarray:
    """The values of one container of a config-3 row (same draws as mixed_container_for_density)."""
    if run_structured:
        nr = int(rng.choice([16, 32, 128, 1024]))
        return vals_runs(rng, nr, min(0.95, max(0.02, d * 2)))
    return vals_density(rng, d)


def fbk_container_of_vals(vals: np.ndarray):
    """Sorted values -> featurebase_amd.roaring.Container in the encoding Container.optimize()
    picks (roaring.go:3412-3461: run if runs <= 2048 and runs <= n/2, else array if n < 4096,
    else bitmap), with numpy only: the benchmark scripts generate their inputs with this, so that
    nothing under oracle/ runs while they measure (tests/test_datagen.py checks it against the
    oracle's optimize())."""
    from featurebase_amd.roaring import Container

    n = int(vals.size)
    if n == 0:
        return None
    rs = runs_of_vals(vals)
    if len(rs) <= 2048 and len(rs) <= n // 2:
        return Container.run(rs, n)
    if n < 4096:
        return Container.array(vals.astype(np.uint16))
    return Container.bitmap(words_of(vals), n)


def mixed_container_for_density(rng, d: float, run_structured: bool):
    """One container of a config-3 row: encoding chosen by optimize() (roaring.go:3412)."""
    from oracle import pyoracle as O

    if run_structured:
        nr = int(rng.choice([16, 32, 128, 1024]))
        vals = vals_runs(rng, nr, min(0.95, max(0.02, d * 2)))
    else:
        vals = vals_density(rng, d)
    if vals.size == 0:
        return None
    c = O.OContainer.array(vals) if vals.size < 4096 else O.OContainer.bitmap(words_of(vals))
    return O.optimize(c)
