"""Seeded synthetic inputs shared by tests, smoke() and bench.py.

Shapes follow the reference's own generators where they exist: the container archetypes
of roaring/container_archetypes.go:67-124 (Run16..Run1024, RunSplit, RunFull, Ary*,
Bitmap*) and the rank-law row densities of fragment_internal_test.go:2786-2834.  The RNG
streams themselves (Go math/rand, apophenia) are not reproducible without Go, so seeds
here are our own (numpy PCG64 seeded from SEED + index).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

# every seeded generator derives from this; FBK_TEST_SEED=<int> re-rolls all randomised tests (scripts/fuzz_parity.sh)
SEED = int(os.environ.get("FBK_TEST_SEED", "0x5EED0001"), 0)
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


# ---- fast flat generators for bench.py's secondary configurations -------------------------
DESC_DTYPE = np.dtype(
    [("key", "<u8"), ("off", "<u8"), ("row", "<u4"), ("len", "<u4"), ("n", "<i4"), ("type", "u1"), ("pad", "u1", (3,))]
)  # == fbk_container_desc (include/fbk.h), 32 bytes


class FlatRows:
    """Rows in the flattened form fbk_batch_upload takes (descriptor table + one payload buffer),
    built without one Python object per container.  Encodings follow Container.optimize()
    (roaring.go:3412-3461), as fbk_container_of_vals does."""

    def __init__(self):
        self.key, self.row, self.len, self.n, self.type, self.off = [], [], [], [], [], []
        self.chunks: List[np.ndarray] = []
        self.bytes = 0
        self.n_rows = 0

    def add(self, row: int, key: int, typ: int, data: np.ndarray, n: int) -> None:
        self.key.append(key)
        self.row.append(row)
        self.type.append(typ)
        self.n.append(n)
        self.len.append(1024 if typ == 2 else (data.size if typ == 1 else data.size // 2))
        self.off.append(self.bytes)
        b = data.view(np.uint8).reshape(-1)
        self.chunks.append(b)
        self.bytes += b.size

    def add_vals(self, row: int, key: int, vals: np.ndarray) -> None:
        """sorted distinct values (int64) of one container"""
        n = int(vals.size)
        if n == 0:
            return
        br = np.nonzero(np.diff(vals) != 1)[0]
        runs = br.size + 1
        if runs <= 2048 and runs <= n // 2:
            iv = np.empty((runs, 2), dtype=np.uint16)
            iv[0, 0] = vals[0]
            iv[1:, 0] = vals[br + 1]
            iv[:-1, 1] = vals[br]
            iv[-1, 1] = vals[-1]
            self.add(row, key, 3, iv, n)
        elif n < 4096:
            self.add(row, key, 1, vals.astype(np.uint16), n)
        else:
            self.add(row, key, 2, words_of(vals), n)

    def add_runs(self, row: int, key: int, starts: np.ndarray, lens: np.ndarray) -> None:
        """disjoint, non-adjacent runs [start, start + len)"""
        n, runs = int(lens.sum()), int(starts.size)
        if runs <= 2048 and runs <= n // 2:
            iv = np.empty((runs, 2), dtype=np.uint16)
            iv[:, 0] = starts
            iv[:, 1] = starts + lens - 1
            self.add(row, key, 3, iv, n)
        else:
            first = np.cumsum(lens) - lens  # index of each run's first value in the flattened list
            self.add_vals(row, key, (np.repeat(starts - first, lens) + np.arange(n)).astype(np.int64))

    def extend(self, other: "FlatRows", row_off: int) -> None:
        self.key += other.key
        self.row += [r + row_off for r in other.row]
        self.len += other.len
        self.n += other.n
        self.type += other.type
        self.off += [o + self.bytes for o in other.off]
        self.chunks += other.chunks
        self.bytes += other.bytes

    def descs(self) -> np.ndarray:
        d = np.zeros(len(self.key), dtype=DESC_DTYPE)
        d["key"], d["off"], d["row"], d["len"], d["n"], d["type"] = self.key, self.off, self.row, self.len, self.n, self.type
        return d

    def payload(self) -> np.ndarray:
        return np.concatenate(self.chunks) if self.chunks else np.zeros(1, dtype=np.uint8)


def sparse_positions(rng, d: float, span: int) -> np.ndarray:
    """Sorted positions in [0, span), each present with probability d: geometric gaps (O(d * span))."""
    m = int(span * d * 1.2) + 64
    pos = np.cumsum(rng.geometric(d, size=m)) - 1
    while pos[-1] < span:  # rare: extend
        more = np.cumsum(rng.geometric(d, size=m)) + pos[-1]
        pos = np.concatenate([pos, more])
    return pos[pos < span]


def bernoulli_words(rng, d: float, shape) -> np.ndarray:
    """uint64 words whose bits are set with probability round(d * 256) / 256 (bit-serial construction)."""
    k = max(1, min(255, int(round(d * 256))))
    acc = np.zeros(shape, dtype=np.uint64)
    for bit in range(8):
        r = rng.integers(0, 2**64, shape, dtype=np.uint64)
        acc = (acc | r) if (k >> bit) & 1 else (acc & r)
    return acc


def _config3_chunk(args):
    s0, s1, k, seed_idx, dens, run_frac = args
    rows, _, filt = config3_flat(s1 - s0, k, seed_idx, first_shard=s0, workers=1, densities=dens, run_frac=run_frac)
    return rows, filt


def config4_densities(n_rows: int = 64, seed_idx: int = 4000):
    """SURVEY.md 8d, configs[3] as written: "field A: 32 rows, field B: 32 rows, densities log-uniform [0.001, 0.5]" — one
    density per FIELD ROW (the same in every shard: a row of a field is one attribute value), drawn once from the seeded
    generator.  i.i.d. bits at density d never make a run container (runs ~ n (1 - d) > n / 2) and make an array below
    4096 / 65536 = 0.0625, so ln(62.5) / ln(500) = 66.5 % of the rows are ARRAY rows — the mix the dense-only runs of rounds
    1-4 never touched."""
    rng = rng_for(seed_idx, 0xD5)
    return np.exp(rng.uniform(np.log(0.001), np.log(0.5), n_rows)).tolist()


def config4_flat(n_shards: int, n_a: int = 32, n_b: int = 32, seed_idx: int = 4000, first_shard: int = 0, workers: int = 0, mp: str = "spawn"):
    """BASELINE.json configs[3] on SURVEY 8d's input: per shard n_a rows of field A then n_b rows of field B (row r of
    the shard = field row r; rows 0 .. n_a-1 are A), densities config4_densities, encodings by optimize()'s rule, plus
    one filter row of density 0.5.  Returns (rows FlatRows, rows_a [n_shards, n_a], rows_b [n_shards, n_b], filter
    FlatRows, densities)."""
    dens = config4_densities(n_a + n_b, seed_idx)
    rows, groups, filt = config3_flat(n_shards, n_a + n_b, seed_idx, first_shard, workers, mp, densities=dens, run_frac=0.0)
    return rows, np.ascontiguousarray(groups[:, :n_a]), np.ascontiguousarray(groups[:, n_a:]), filt, dens


def config3_flat(n_shards: int, k: int = 64, seed_idx: int = 3000, first_shard: int = 0, workers: int = 0, mp: str = "spawn", densities=None, run_frac: float = 0.25):
    """BASELINE.json configs[2]: per shard k rows of rank-law density clamp(0.5 (r+1)^-1.1, 0.001, 0.5)
    — a quarter of the containers run-structured — plus one filter row of density 0.5.
    Returns (rows FlatRows, groups [n_shards, k], filter FlatRows).  Shard s is seeded by (seed_idx,
    s), so the data does not depend on how the shards are split over `workers` processes.
    densities / run_frac: the same generator on another density law (config4_flat)."""
    rows, filt = FlatRows(), FlatRows()
    groups = np.arange(n_shards * k, dtype=np.uint32).reshape(n_shards, k)
    if workers == 0:
        import os

        workers = max(1, min(16 if n_shards < 1024 else 64, (os.cpu_count() or 1) // 2, n_shards // 8))
    if workers > 1:
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor

        per = (n_shards + workers - 1) // workers
        jobs = [(first_shard + a, first_shard + min(n_shards, a + per), k, seed_idx, densities, run_frac) for a in range(0, n_shards, per)]
        # mp="fork" only before the process has initialised the HIP runtime (bench.py generates first)
        with ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context(mp)) as ex:
            for (a, *_), (r, f) in zip(jobs, ex.map(_config3_chunk, jobs)):
                rows.extend(r, (a - first_shard) * k)
                filt.extend(f, a - first_shard)
        rows.n_rows, filt.n_rows = n_shards * k, n_shards
        return rows, groups, filt
    for s_local in range(n_shards):
        s = first_shard + s_local
        rng = rng_for(seed_idx, s)
        for r in range(k):
            d = zipf_density(r) if densities is None else float(densities[r])
            row = s_local * k + r
            is_run = rng.random(SLOTS) < run_frac
            plain = np.nonzero(~is_run)[0]
            if plain.size:
                if d >= 0.07:  # bitmaps (n >= 4096 with overwhelming probability)
                    w = bernoulli_words(rng, d, (plain.size, WORDS))
                    cnt = np.bitwise_count(w).sum(axis=1)
                    for i, slot in enumerate(plain):
                        if cnt[i] >= 4096:
                            rows.add(row, s * 16 + int(slot), 2, w[i], int(cnt[i]))
                        else:
                            rows.add_vals(row, s * 16 + int(slot), np.nonzero(np.unpackbits(w[i].view(np.uint8), bitorder="little"))[0].astype(np.int64))
                else:
                    # i.i.d. bits at density d < 0.07: runs ~ n (1 - d) > n / 2, never a run container;
                    # n ~ 65536 d < 4096 except in the tail of the distribution (checked)
                    pos = sparse_positions(rng, d, plain.size * 65536)
                    cut = np.searchsorted(pos, np.arange(1, plain.size + 1) * 65536)
                    v16 = (pos & 0xFFFF).astype(np.uint16)
                    lo = 0
                    for i, slot in enumerate(plain):
                        hi = int(cut[i])
                        if hi - lo >= 4096:
                            rows.add_vals(row, s * 16 + int(slot), pos[lo:hi] - i * 65536)
                        elif hi > lo:
                            rows.add(row, s * 16 + int(slot), 1, v16[lo:hi], hi - lo)
                        lo = hi
            for slot in np.nonzero(is_run)[0]:
                nr = int(rng.choice([16, 32, 128, 1024]))
                fill = min(0.95, max(0.02, d * 2))
                period = 65536 // nr
                starts = np.arange(nr) * period + rng.integers(0, max(1, period // 4), nr)
                lens = np.maximum(1, (period * fill * rng.uniform(0.5, 1.0, nr)).astype(np.int64))
                lens = np.maximum(np.minimum(lens, period - (starts - np.arange(nr) * period) - 1), 1)
                rows.add_runs(row, s * 16 + int(slot), starts, lens)
        w = rng.integers(0, 2**64, (SLOTS, WORDS), dtype=np.uint64)
        for slot in range(SLOTS):
            filt.add(s_local, s * 16 + slot, 2, w[slot], int(np.bitwise_count(w[slot]).sum()))
    rows.n_rows, filt.n_rows = n_shards * k, n_shards
    return rows, groups, filt


def config3_flat_subprocess(n_shards: int, k: int = 64, seed_idx: int = 3000, config4: bool = False, first_shard: int = 0):
    """config3_flat (config4 = True: config4_flat with k = n_a + n_b rows, half each) run in a child process (which forks
    its generator workers): for callers that have already initialised the HIP runtime, where a fork of THIS process
    would be unsafe and a spawn would re-import the caller's main module.  Returns (descs, payload, n_rows, groups,
    fdescs, fpayload, encoded_bytes)."""
    import subprocess
    import sys
    import tempfile

    with tempfile.TemporaryDirectory(prefix="fbk_cfg3_") as tmp:
        gen = (f"rows, _, _, filt, _ = D.config4_flat({n_shards}, {k // 2}, {k - k // 2}, {seed_idx}, first_shard={first_shard}, mp='fork')\n" if config4 else
               f"rows, groups, filt = D.config3_flat({n_shards}, {k}, {seed_idx}, first_shard={first_shard}, mp='fork')\n")
        code = (
            "import sys, numpy as np\n"
            f"sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
            "import datagen as D\n" + gen +
            f"np.save({tmp!r} + '/d.npy', rows.descs()); np.save({tmp!r} + '/p.npy', rows.payload())\n"
            f"np.save({tmp!r} + '/fd.npy', filt.descs()); np.save({tmp!r} + '/fp.npy', filt.payload())\n"
        )
        env = dict(os.environ, FBK_TEST_SEED=hex(SEED))
        subprocess.check_call([sys.executable, "-c", code], env=env)
        d, p = np.load(tmp + "/d.npy"), np.load(tmp + "/p.npy")
        fd, fp = np.load(tmp + "/fd.npy"), np.load(tmp + "/fp.npy")
    groups = np.arange(n_shards * k, dtype=np.uint32).reshape(n_shards, k)
    return d, p, n_shards * k, groups, fd, fp, int(p.size + fp.size)
