"""Differential fuzzing of chained operations, in the spirit of the reference's roaring/fuzzer.go
(:26-120: random op sequences checked against roaring/naive.go): a population of device
batches (uploaded flat, dense, from a serialised image, or produced by earlier operations, with
and without optimize()) is mutated by random set-ops / folds / serialisation round trips, and
after every step compared bit for bit with an independent numpy bitset model.  Catches state
bugs the per-kernel tests cannot: stale descriptor copies, the dense flag surviving a sparse
result, recycled pool blocks, keys carried through chains."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2, 3, 0], ids=["pair-kernels-r2", "pair-kernels-r3", "pair-kernels-ring", "pair-kernels-auto"], autouse=True)
def pair_kernel_generation(request, gpu_ctx):
    """Every test of this file runs with the round-2 pair kernels (k_icount / k_setop), with the round-3 ones (k_icount2 /
    k_setop2: table + probe, interior-map run decode, one-wave blocks) and with the library's own choice by payload size:
    each generation is checked against the oracle on every input of the file, not only on the rows the dispatch would
    hand it."""
    try:
        gpu_ctx.set_option("pair_kernels", request.param)
    except Exception:
        if request.param != 3:
            raise
        pytest.skip("k_icount3 exists in -DFBK_EXPERIMENTS builds only")
    yield request.param
    gpu_ctx.set_option("pair_kernels", 0)

N_ROWS = 6  # rows per batch; row r is "shard r": keys r*16 + slot


def model_of_rows(rows):
    """[n_rows][16][1024] uint64 bitset model of a list of {key: oracle container} rows"""
    m = np.zeros((len(rows), 16, 1024), dtype=np.uint64)
    for r, row in enumerate(rows):
        for k, c in row.items():
            m[r, k & 15] = c.words()
    return m


def check(batch, model, what):
    rows = batch.download()
    assert len(rows) == model.shape[0], what
    got = np.zeros_like(model)
    for r, row in enumerate(rows):
        for k, c in row.items():
            assert k >> 4 == r, (what, "key moved", k, r)
            assert c.n == int(np.bitwise_count(c.words()).sum()) and c.n > 0, (what, "stored n", k)
            got[r, k & 15] = c.words()
    assert (got == model).all(), what
    cnt = batch.count(np.arange(model.shape[0]))
    assert cnt.tolist() == np.bitwise_count(model).sum(axis=(1, 2)).tolist(), what


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_chained_ops_vs_bitset_model(gpu_ctx, oracle, seed):
    O = oracle
    ctx = gpu_ctx
    rng = D.rng_for(seed)
    pop = []  # (batch, model)

    def add(batch, model, what):
        check(batch, model, what)
        pop.append((batch, model))
        if len(pop) > 8:  # retire the oldest: its blocks go back to the pool and get reused
            old, _ = pop.pop(int(rng.integers(0, 3)))
            old.free()

    # seed population: flat upload (mixed encodings), dense upload, serialised image
    rows = [{r * 16 + (k & 15): c for k, c in D.random_row(rng, 0).items()} for r in range(N_ROWS)]
    add(ctx.upload([D.to_fbk_row(r) for r in rows]), model_of_rows(rows), "flat")
    w = D.dense_rows(N_ROWS, 0.4, 700 + seed)
    add(ctx.upload_dense(w), w.reshape(N_ROWS, 16, 1024).copy(), "dense")
    rows2 = [{r * 16 + (k & 15): c for k, c in D.random_row(rng, 0).items()} for r in range(N_ROWS)]
    for r in rows2:
        r.setdefault((rows2.index(r)) * 16, O.OContainer.array([1]))  # every row id present
    img = O.OBitmap.from_containers([kv for r in rows2 for kv in r.items()]).marshal(True)
    b, ids = ctx.upload_roaring(img)
    assert ids.tolist() == list(range(N_ROWS))
    add(b, model_of_rows(rows2), "roaring image")
    w2 = D.dense_rows(N_ROWS, 0.5, 800 + seed)
    add(ctx.upload_dense(w2), w2.reshape(N_ROWS, 16, 1024).copy(), "dense2")

    npop = {L.OP_AND: np.bitwise_and, L.OP_OR: np.bitwise_or, L.OP_XOR: np.bitwise_xor, L.OP_ANDNOT: lambda a, b: a & ~b}
    idx = np.arange(N_ROWS)
    for step in range(60):
        kind = rng.integers(0, 10)
        ia, ib = rng.integers(0, len(pop), size=2)
        (A, ma), (B, mb) = pop[ia], pop[ib]
        if kind < 5:  # pairwise set-op, random row permutation on the B side
            op = int(rng.integers(0, 4))
            flags = L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0
            out, cnt = ctx.setop(op, A, idx, B, idx, flags)
            m = npop[op](ma, mb)
            assert cnt.tolist() == np.bitwise_count(m).sum(axis=(1, 2)).tolist(), ("setop counts", step)
            ic = ctx.intersection_count(A, idx, B, idx)
            assert ic.tolist() == np.bitwise_count(ma & mb).sum(axis=(1, 2)).tolist(), ("icount", step)
            add(out, m, f"step {step}: setop {op} flags {flags}")
        elif kind < 8:  # n-way fold over rows of ONE batch: group g = rows (g, g+1, g+2) mod N
            op = int(rng.integers(0, 4))
            k = int(rng.integers(1, 4))
            groups = np.array([[(g + j) % N_ROWS for j in range(k)] for g in range(N_ROWS)], dtype=np.uint32)
            out, cnt = ctx.fold_n(op, A, groups, L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0)
            m = ma[groups[:, 0]].copy()
            if op == L.OP_ANDNOT:
                for j in range(1, k):
                    m &= ~ma[groups[:, j]]
            else:
                for j in range(1, k):
                    m = npop[op](m, ma[groups[:, j]])
            assert cnt.tolist() == np.bitwise_count(m).sum(axis=(1, 2)).tolist(), ("fold counts", step, op, k)
            # fold outputs carry the key high bits of each group's first row: re-key to row order
            rows_out = out.download()
            got = np.zeros_like(m)
            for r, row in enumerate(rows_out):
                for kk, c in row.items():
                    got[r, kk & 15] = c.words()
            assert (got == m).all(), ("fold", step, op, k)
            out.free()
        elif kind < 9:  # serialise -> upload again (only batches whose keys are row-ordered)
            img = A.to_roaring()
            b2, ids = ctx.upload_roaring(img)
            keep = [r for r in range(N_ROWS) if np.bitwise_count(ma[r]).sum() > 0]
            assert ids.tolist() == keep, ("roundtrip rows", step)
            m2 = ma[keep]
            rows_out = b2.download()
            got = np.zeros_like(m2)
            for r, row in enumerate(rows_out):
                for kk, c in row.items():
                    got[r, kk & 15] = c.words()
            assert (got == m2).all(), ("roundtrip", step)
            assert O.OBitmap.unmarshal(img).count() == int(np.bitwise_count(ma).sum())
            b2.free()
        else:  # count ranges
            s, e = sorted(int(x) for x in rng.integers(0, (1 << 20) + 1, size=2))
            ctx.set_option("count_range_reference_quirk", 0)  # (the bit count; the reference-identical default is tested in test_gpu_parity.py)
            got = ctx.count_range(A, idx, s, e)
            ctx.set_option("count_range_reference_quirk", 1)
            bits = np.unpackbits(ma.reshape(N_ROWS, -1).view(np.uint8), axis=1, bitorder="little")
            assert got.tolist() == bits[:, s:e].sum(axis=1).tolist(), ("count_range", step, s, e)
    for b, _ in pop:
        b.free()
