"""GPU parity tests proper: the HIP path (through the C ABI) vs the CPU oracle on the same
seeded inputs, bit-exact.  Run on the MI355X box with `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest

import datagen as D
import go_fixtures as G
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2, 3, 0], ids=["pair-kernels-r2", "pair-kernels-r3", "pair-kernels-ring", "pair-kernels-auto"], autouse=True)
def pair_kernel_generation(request, gpu_ctx):
    """Every test of this file runs with the round-2 pair kernels (k_icount / k_setop), with the round-3 ones (k_icount2 /
    k_setop2: table + probe, interior-map run decode, one-wave blocks; array x run by probing the run table) with the round-6
    persistent loader / decoder count (k_icount3: payloads through an LDS ring; set-ops as round 3 — experiments builds
    only, skipped on the product library) and with the
    library's own choice by payload size: each generation is checked against the oracle on every input of the file, not
    only on the rows the dispatch would hand it."""
    try:
        gpu_ctx.set_option("pair_kernels", request.param)
    except Exception:
        if request.param != 3:
            raise
        pytest.skip("k_icount3 (parity-green, 1.5 x slower than k_icount2) exists in -DFBK_EXPERIMENTS builds only")
    yield request.param
    gpu_ctx.set_option("pair_kernels", 0)

OPS = [(L.OP_AND, "intersect"), (L.OP_OR, "union"), (L.OP_XOR, "xor"), (L.OP_ANDNOT, "difference")]
ZERO = np.zeros(1024, dtype=np.uint64)
COMBOS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "container_combinations.json")))["ops"]


def expected_setop(O, name, a, b):
    """Bitmap-level result for one key (roaring.go:736-759, 1292-1315, 1573-1595, 1598-1623):
    a/b are oracle containers or None (key absent on that side)."""
    if a is not None and b is not None:
        return O.OPS[name](a, b)
    if name == "intersect":
        return O.OContainer(None)  # keys present on one side only are dropped (:743-757)
    if name == "difference":
        return a if a is not None else O.OContainer(None)
    return a if a is not None else b  # union / xor copy the unmatched container (:1299-1306)


def check_rows(O, ctx, rows_a, rows_b):
    n = len(rows_a)
    A, B = ctx.upload([D.to_fbk_row(r) for r in rows_a]), ctx.upload([D.to_fbk_row(r) for r in rows_b])
    idx = np.arange(n)
    # Row.Count (row.go:446) = sum of stored N
    assert A.count(idx).tolist() == [sum(c.n for c in r.values()) for r in rows_a]
    # Bitmap.IntersectionCount (roaring.go:711-733)
    got = ctx.intersection_count(A, idx, B, idx)
    for r in range(n):
        exp = sum(O.intersection_count(rows_a[r][k], rows_b[r][k]) for k in rows_a[r] if k in rows_b[r])
        assert int(got[r]) == exp, ("intersection_count", r)
    for op, name in OPS:
        out, cnt = ctx.setop(op, A, idx, B, idx)
        res = out.download()
        assert out.count(idx).tolist() == cnt.tolist()
        for r in range(n):
            tot = 0
            keys = set(rows_a[r]) | set(rows_b[r])
            for k in keys:
                e = expected_setop(O, name, rows_a[r].get(k), rows_b[r].get(k))
                tot += e.n
                g = res[r].get(k)
                gw = g.words() if g is not None else ZERO
                assert (gw == e.words()).all(), (name, r, k)
                if g is not None:
                    assert g.n == e.n
            assert set(res[r]) <= keys
            assert int(cnt[r]) == tot, (name, r)
        out.free()
    A.free()
    B.free()


def test_golden_container_combinations_on_gpu(gpu_ctx, oracle):
    """The reference's golden combination table (roaring_internal_test.go:2974-3771: 638 op entries), every
    op x 3x3 encodings, evaluated by the HIP kernels: one shard row per triple."""
    O = oracle
    mk = {
        1: lambda p: O.OContainer.array(G.pattern_values(p).astype(np.uint16)),
        2: lambda p: O.OContainer.bitmap(G.pattern_words(p)),
        3: lambda p: O.OContainer.run(G.pattern_runs(p)),
    }
    opmap = {"intersect": L.OP_AND, "union": L.OP_OR, "difference": L.OP_ANDNOT, "xor": L.OP_XOR}
    for tx in (1, 2, 3):
        for ty in (1, 2, 3):
            by_op = {}
            for t in COMBOS:
                if t["op"] in opmap:
                    by_op.setdefault(t["op"], []).append(t)
            for opname, triples in by_op.items():
                rows_a = [{i * 16 + (i % 16): D.to_fbk(mk[tx](t["x"]))} for i, t in enumerate(triples)]
                rows_b = [{i * 16 + (i % 16): D.to_fbk(mk[ty](t["y"]))} for i, t in enumerate(triples)]
                A, B = gpu_ctx.upload(rows_a), gpu_ctx.upload(rows_b)
                idx = np.arange(len(triples))
                out, cnt = gpu_ctx.setop(opmap[opname], A, idx, B, idx)
                res = out.download()
                ic = gpu_ctx.intersection_count(A, idx, B, idx)
                for i, t in enumerate(triples):
                    expw = G.pattern_words(t["exp"])
                    g = res[i].get(i * 16 + (i % 16))
                    gw = g.words() if g is not None else ZERO
                    assert (gw == expw).all(), (t, tx, ty)
                    assert int(cnt[i]) == int(np.bitwise_count(expw).sum())
                    if opname == "intersect":
                        assert int(ic[i]) == int(cnt[i])
                for b in (out, A, B):
                    b.free()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mixed_rows_vs_oracle(gpu_ctx, oracle, seed):
    rng = D.rng_for(100 + seed)
    n = 48
    rows_a = [D.random_row(rng, r) for r in range(n)]
    rows_b = [D.random_row(rng, r) for r in range(n)]
    check_rows(oracle, gpu_ctx, rows_a, rows_b)


def test_every_type_pair_every_kind(gpu_ctx, oracle):
    """All ordered pairs of container shapes, incl. arrays > 4096 and near-threshold sizes."""
    rng = D.rng_for(7)
    kinds = D.KINDS
    rows_a, rows_b = [], []
    r = 0
    for ka in kinds:
        for kb in kinds:
            rows_a.append({r * 16 + 3: D.oracle_container(rng, ka)})
            rows_b.append({r * 16 + 3: D.oracle_container(rng, kb)})
            r += 1
    check_rows(oracle, gpu_ctx, rows_a, rows_b)


def test_empty_and_ragged_inputs(gpu_ctx, oracle):
    O = oracle
    # completely empty rows, rows with one container, disjoint slots, row self-intersection
    rows_a = [{}, {5: O.OContainer.array([1, 2, 3])}, {16 + 0: O.OContainer.run([(0, 65535)])}, {32 + 15: O.OContainer.array([65535])}]
    rows_b = [{}, {6: O.OContainer.array([1, 2, 3])}, {16 + 0: O.OContainer.run([(0, 65535)])}, {}]
    check_rows(O, gpu_ctx, rows_a, rows_b)
    # zero pairs
    A = gpu_ctx.upload([D.to_fbk_row(r) for r in rows_a])
    assert gpu_ctx.intersection_count(A, [], A, []).size == 0
    out, cnt = gpu_ctx.setop(L.OP_OR, A, [], A, [])
    assert cnt.size == 0 and out.info()[0] == 0
    out.free()
    # a batch with zero rows
    E = gpu_ctx.upload([])
    assert E.info() == (0, 0, 0)
    E.free()
    A.free()


def test_invalid_inputs_are_rejected(gpu_ctx):
    from featurebase_amd.roaring import Container

    with pytest.raises(L.FbkError):  # unsorted array
        gpu_ctx.upload([{0: Container(L.TYPE_ARRAY, np.array([5, 3], dtype=np.uint16), 2)}])
    with pytest.raises(L.FbkError):  # overlapping runs
        gpu_ctx.upload([{0: Container(L.TYPE_RUN, np.array([[0, 10], [5, 20]], dtype=np.uint16), 27)}])
    with pytest.raises(L.FbkError):  # two containers in one slot
        gpu_ctx.upload([{0: Container.array([1]), 16: Container.array([2])}])
    A = gpu_ctx.upload([{0: Container.array([1])}])
    with pytest.raises(L.FbkError):  # row out of range
        gpu_ctx.intersection_count(A, [1], A, [0])
    A.free()


def test_recount_on_device(gpu_ctx, oracle):
    """n == -1 containers are counted on the device (Container.count, roaring.go:3052)."""
    from featurebase_amd.roaring import Container

    rng = D.rng_for(11)
    row_o = D.random_row(rng, 0, p_missing=0.0)
    row = {}
    for k, c in row_o.items():
        f = D.to_fbk(c)
        if f.typ != L.TYPE_ARRAY:
            f = Container(f.typ, f.data, -1)
        row[k] = f
    A = gpu_ctx.upload([row])
    assert int(A.count([0])[0]) == sum(c.n for c in row_o.values())
    back = A.download()[0]
    for k, c in row_o.items():
        if c.n:
            assert back[k].n == c.n
    A.free()


def test_dense_config1_single_shard(gpu_ctx, oracle):
    """BASELINE config 1: one shard, 2 rows x 1M columns at ~10 % density, Intersect + Count;
    checked against the oracle's Bitmap.Intersect/Count and a numpy model."""
    O = oracle
    w = D.dense_rows(2, 0.10, 1)
    A = gpu_ctx.upload_dense(w)
    ic = gpu_ctx.intersection_count(A, [0], A, [1])
    out, cnt = gpu_ctx.setop(L.OP_AND, A, [0], A, [1])
    a = O.OBitmap.from_containers([(s, O.OContainer.bitmap(w[0, s])) for s in range(16)])
    b = O.OBitmap.from_containers([(s, O.OContainer.bitmap(w[1, s])) for s in range(16)])
    exp_bm = a.intersect(b)
    assert int(ic[0]) == a.intersection_count(b) == exp_bm.count() == int(cnt[0])
    assert int(ic[0]) == int(np.bitwise_count(w[0] & w[1]).sum())
    res = out.download()[0]
    for k, c in exp_bm.items():
        assert (res[k].words() == c.words()).all()
    out.free()
    A.free()


def test_dense_many_shards_properties(gpu_ctx):
    """Size-independent properties at a larger size (256 shards x 2 rows, 50 %):
    |A∩B| + |A\\B| == |A|; |A∪B| == |A| + |B| - |A∩B|; |A⊕B| == |A∪B| - |A∩B|;
    A∩A == A; totals == numpy popcounts."""
    n = 256
    wa, wb = D.dense_rows(n, 0.5, 21), D.dense_rows(n, 0.5, 22)
    A, B = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb)
    idx = np.arange(n)
    ca, cb = A.count(idx), B.count(idx)
    assert ca.tolist() == np.bitwise_count(wa).reshape(n, -1).sum(1).tolist()
    iab = gpu_ctx.intersection_count(A, idx, B, idx)
    assert iab.tolist() == np.bitwise_count(wa & wb).reshape(n, -1).sum(1).tolist()
    outs = {}
    for op, name in OPS:
        o, c = gpu_ctx.setop(op, A, idx, B, idx)
        outs[name] = c
        o.free()
    assert (outs["intersect"] == iab).all()
    assert (outs["intersect"] + outs["difference"] == ca).all()
    assert (outs["union"] == ca + cb - iab).all()
    assert (outs["xor"] == outs["union"] - iab).all()
    assert (gpu_ctx.intersection_count(A, idx, A, idx) == ca).all()
    # plan path: per-pair counts + device-side total
    plan = gpu_ctx.plan(A, idx, B, idx[::-1].copy())
    plan.intersection_count()
    plan.total()
    cnt, tot = plan.read(want_total=True)
    assert cnt.tolist() == np.bitwise_count(wa & wb[::-1]).reshape(n, -1).sum(1).tolist()
    assert tot == int(cnt.sum())
    plan.setop(L.OP_AND)
    plan.total()
    cnt2, tot2 = plan.read(want_total=True)
    assert (cnt2 == cnt).all() and tot2 == tot
    # the set-op output of a dense plan feeds the dense kernels again: (A∩B)∩B == A∩B
    O2 = plan.output()
    assert (gpu_ctx.intersection_count(O2, idx, B, idx[::-1].copy()) == cnt).all()
    plan.free()
    A.free()
    B.free()


def test_fused_count_and_total_plan(gpu_ctx):
    """fbk_plan_intersection_count_total: counts + per-node sum in one launch (last-block
    reduce across all XCDs), repeated back to back, dense and mixed batches."""
    import torch

    n = 700
    wa, wb = D.dense_rows(n, 0.5, 31), D.dense_rows(n, 0.3, 32)
    A, B = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb)
    rows = np.arange(n)
    plan = gpu_ctx.plan(A, rows, B, rows[::-1].copy())
    exp = np.bitwise_count(wa & wb[::-1]).sum(axis=(1, 2)).astype(np.uint64) if wa.ndim == 3 else None
    if exp is None:
        exp = np.bitwise_count(wa.reshape(n, -1) & wb.reshape(n, -1)[::-1]).sum(axis=1).astype(np.uint64)
    for _ in range(25):  # the ticket counter must be back at 0 after every launch
        plan.intersection_count_total()
    counts, total = plan.read(want_total=True)
    assert (counts == exp).all() and total == int(exp.sum())
    # into a caller-owned device cell
    cell = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()  # (the fill runs on torch's stream, the kernel on the context's)
    plan.intersection_count_total(cell.data_ptr())
    gpu_ctx.synchronize()
    assert int(cell.item()) == int(exp.sum())
    # the accumulate form: every workgroup adds into a zeroed cell; three launches add up
    acc = torch.zeros(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        plan.intersection_count_accumulate(acc.data_ptr() + 8)
    gpu_ctx.synchronize()
    assert acc.tolist() == [0, 3 * int(exp.sum())]
    assert (plan.read() == exp).all()
    plan.free()
    # mixed batch: generic kernel + sum kernel behind the same entry point
    rng = D.rng_for(33)
    ra = [D.random_row(rng, r) for r in range(40)]
    M = gpu_ctx.upload([D.to_fbk_row(r) for r in ra])
    p2 = gpu_ctx.plan(M, np.arange(40), A, np.arange(40))
    p2.intersection_count_total()
    c2, t2 = p2.read(want_total=True)
    assert t2 == int(c2.sum()) and t2 > 0
    acc = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    p2.intersection_count_accumulate(acc.data_ptr())
    gpu_ctx.synchronize()
    assert int(acc.item()) == t2
    p2.free()
    for b in (A, B, M):
        b.free()


def test_reference_bitmap_level_vectors_through_the_abi(gpu_ctx, oracle):
    """TestBitmap_IntersectionCount_* and testBM() (roaring/roaring_test.go:1283-1387, 1661-1684),
    the reference's own known answers, through fbk_intersection_count / fbk_count / fbk_setop:
    both orders, as the reference checks them."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import go_bitmap_vectors as V
    from test_oracle_bitmap_vectors import file_bitmap

    O = oracle
    rows, want = [], []
    for name, a, b, n in V.CASES:
        rows.append(dict(file_bitmap(O, *a)))  # keys 0..15: one shard row each
        rows.append(dict(file_bitmap(O, *b)))
        want.append(n)
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    ia = np.arange(0, len(rows), 2)
    got = gpu_ctx.intersection_count(batch, ia, batch, ia + 1)
    rev = gpu_ctx.intersection_count(batch, ia + 1, batch, ia)
    assert got.tolist() == want and rev.tolist() == want
    out, cnt = gpu_ctx.setop(L.OP_AND, batch, ia, batch, ia + 1)
    assert cnt.tolist() == want
    out.free()
    tb = [i for i, c in enumerate(V.CASES) if c[0] == "Mixed/self"][0]
    assert int(batch.count([2 * tb])[0]) == V.TEST_BM_COUNT  # "count 75007"
    batch.free()


def test_reference_bitmap_level_setop_vectors_through_the_abi(gpu_ctx, oracle):
    """TestBitmap_Intersection / _Union1 / _Intersect* (incl. the *InPlace forms, :499-780) /
    _Difference* / _Union / _Xor* (roaring/roaring_test.go:483-1216) through fbk_setop: the bitmaps are cut into shard rows
    (16 container keys each, as fragment.row does), one row pair per shard either operand touches;
    result cardinalities summed over the shards and, where the reference checks them, the
    columns of the result."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import go_bitmap_vectors as V
    from test_oracle_bitmap_vectors import file_bitmap

    O = oracle
    ops = {"and": L.OP_AND, "or": L.OP_OR, "andnot": L.OP_ANDNOT, "xor": L.OP_XOR}
    cache = {}

    def shard_rows(spec):
        key = id(spec[0]), spec[1]
        if key not in cache:
            rows = {}
            for k, c in file_bitmap(O, *spec):
                rows.setdefault(k >> 4, {})[k] = c
            cache[key] = rows
        return cache[key]

    for name, op, a, b, want, want_slice in V.SETOP_CASES:
        ra, rb = shard_rows(a), shard_rows(b)
        shards = sorted(set(ra) | set(rb)) or [0]
        batch = gpu_ctx.upload([D.to_fbk_row(ra.get(s, {})) for s in shards] + [D.to_fbk_row(rb.get(s, {})) for s in shards])
        n = len(shards)
        for flags in (0, L.SETOP_OPTIMIZE):
            out, cnt = gpu_ctx.setop(ops[op], batch, np.arange(n), batch, np.arange(n) + n, flags)
            assert int(cnt.sum()) == want, (name, flags)
            if want_slice is not None:
                cols = []
                for i, row in enumerate(out.download()):
                    for k, c in row.items():
                        vals = np.nonzero(np.unpackbits(c.words().view(np.uint8), bitorder="little"))[0]
                        cols.extend((((shards[i] * 16 + (k & 15)) << 16) + vals).tolist())
                assert sorted(cols) == want_slice, (name, flags)
            out.free()
        if op == "and":  # the count-only form of the same intersections
            assert int(gpu_ctx.intersection_count(batch, np.arange(n), batch, np.arange(n) + n).sum()) == want, name
        batch.free()
    # bm0.IntersectInPlace(bm11, bm12): a three-way fold through fbk_fold_n
    for name, op, specs, want, want_slice in V.FOLD_CASES:
        parts = [shard_rows(sp) for sp in specs]
        shards = sorted(set().union(*[set(p) for p in parts])) or [0]
        n = len(shards)
        batch = gpu_ctx.upload([D.to_fbk_row(p.get(s, {})) for p in parts for s in shards])
        groups = np.array([[j * n + i for j in range(len(parts))] for i in range(n)], dtype=np.uint32)
        out, cnt = gpu_ctx.fold_n(ops[op], batch, groups)
        assert int(cnt.sum()) == want, name
        cols = []
        for i, row in enumerate(out.download()):
            for k, c in row.items():
                vals = np.nonzero(np.unpackbits(c.words().view(np.uint8), bitorder="little"))[0]
                cols.extend((((shards[i] * 16 + (k & 15)) << 16) + vals).tolist())
        if want_slice is not None:
            assert sorted(cols) == want_slice, name
        out.free()
        batch.free()


def test_reference_bitmap_level_count_range_vectors_through_the_abi(gpu_ctx, oracle):
    """TestBitmap_BitmapCountRangeEdgeCase / _BitmapCountRange / _ArrayCountRange / _RunCountRange
    (roaring/roaring_test.go:368-482) through fbk_count_range: the bitmap cut into shard rows, the
    range clipped to every shard it overlaps (what Row.CountRange over its segments amounts to);
    start > end counts nothing, as in the reference's loop (the ABI rejects it)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import go_bitmap_vectors as V
    from test_oracle_bitmap_vectors import file_bitmap

    O = oracle
    SW = 1 << 20
    for name, spec, ranges in V.COUNT_RANGE_CASES:
        rows = {}
        for k, c in file_bitmap(O, *spec):
            rows.setdefault(k >> 4, {})[k] = c
        shards = sorted(rows)
        batch = gpu_ctx.upload([D.to_fbk_row(rows[s]) for s in shards])
        for s, e, want in ranges:
            got = 0
            if s < e:
                for i, sh in enumerate(shards):
                    lo, hi = max(s, sh * SW), min(e, (sh + 1) * SW)
                    if lo < hi:
                        got += int(gpu_ctx.count_range(batch, [i], lo - sh * SW, hi - sh * SW)[0])
            assert got == want, (name, s, e)
        if name == "EdgeCase":
            assert int(batch.count(np.arange(len(shards))).sum()) == ranges[0][2]
        batch.free()


def test_count_range_counts_bits_where_run_count_range_double_counts(gpu_ctx, oracle):
    """RunCountRange's double count (roaring.go:3216-3227, pinned on the oracle in
    tests/test_oracle_bitmap_vectors.py::test_run_count_range_overcount_is_pinned): the DEFAULT
    (count_range_reference_quirk = 1, ABI 5) returns the reference's number on the same call, option = 0 the bits
    of [start, end)."""
    from featurebase_amd.roaring import Container

    b = gpu_ctx.upload([{0: Container.run([(10, 20), (30, 40)])}])
    bm = oracle.OBitmap.from_containers([(0, oracle.OContainer.run([(10, 20), (30, 40)]))])
    assert bm.count_range(15, 40) == 27  # the reference's answer
    assert gpu_ctx.get_option("count_range_reference_quirk") == 1
    assert gpu_ctx.count_range(b, [0], 15, 40).tolist() == [27]
    gpu_ctx.set_option("count_range_reference_quirk", 0)
    try:
        assert gpu_ctx.count_range(b, [0], 15, 40).tolist() == [16]
    finally:
        gpu_ctx.set_option("count_range_reference_quirk", 1)
    assert gpu_ctx.count_range(b, [0], 12, 35).tolist() == [bm.count_range(12, 35)]
    b.free()


def test_count_range_reference_quirk_mode_matches_the_reference_everywhere(gpu_ctx, oracle):
    """Option count_range_reference_quirk = 1: fbk_count_range returns what Bitmap.CountRange returns in the
    reference ON THE SAME INPUTS, RunCountRange's over-count (roaring.go:3216-3227) included (the default since
    ABI 5); = 0 returns the number of bits in [start, end).  Random mixed rows and ranges chosen to hit run ends."""
    from featurebase_amd.roaring import Container

    b = gpu_ctx.upload([{0: Container.run([(10, 20), (30, 40)])}])
    gpu_ctx.set_option("count_range_reference_quirk", 1)
    try:
        assert gpu_ctx.count_range(b, [0], 15, 40).tolist() == [27]
        assert gpu_ctx.count_range(b, [0], 15, 41).tolist() == [17]
        b.free()
        rng = D.rng_for(811)
        rows = [D.random_row(rng, 0) for _ in range(24)]
        batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
        obms = [oracle.OBitmap.from_containers(sorted(r.items())) for r in rows]
        words = []
        for r in rows:
            w = np.zeros((16, 1024), dtype=np.uint64)
            for k, c in r.items():
                w[k & 15] = c.words()
            words.append(w.reshape(-1))
        idx = np.arange(len(rows))
        ends = []
        for r in rows:  # range ends on, one before and one past a run's last value
            for k, c in r.items():
                if c.typ == oracle.RUN and c.n:
                    last = int(c.data()[int(rng.integers(0, len(c.data())))][1])
                    ends += [((k & 15) << 16) + last + d for d in (0, 1, 2) if ((k & 15) << 16) + last + d <= 1 << 20]
        ranges = [(int(rng.integers(0, e + 1)), e) for e in ends[:60]] + [(0, 1 << 20), (0, 0), (65536, 131072), (5, 65536 * 3 + 9)]
        n_quirk = 0
        for s, e in ranges:
            want_ref = [bm.count_range(s, e) for bm in obms]
            gpu_ctx.set_option("count_range_reference_quirk", 1)
            assert gpu_ctx.count_range(batch, idx, s, e).tolist() == want_ref, (s, e)
            gpu_ctx.set_option("count_range_reference_quirk", 0)
            bits = [int(np.bitwise_count(_mask_range(w, s, e)).sum()) for w in words]
            assert gpu_ctx.count_range(batch, idx, s, e).tolist() == bits, (s, e)
            n_quirk += int(want_ref != bits)
        assert n_quirk > 0, "no range hit the quirk: the test does not exercise the strict mode"
        batch.free()
    finally:
        gpu_ctx.set_option("count_range_reference_quirk", 1)


def _mask_range(w, s, e):
    """words of a row with every bit outside [s, e) cleared"""
    bits = np.unpackbits(w.view(np.uint8), bitorder="little").copy()
    bits[:s] = 0
    bits[e:] = 0
    return np.packbits(bits, bitorder="little").view(np.uint64)


def test_count_array_forms_odd_lengths_and_value_zero(gpu_ctx, oracle):
    """The count path's array forms of round 6 (fbk_pair_kernels.hip.h: floor(len / 2) full dwords + an odd last value by itself;
    the probe takes no per-value predicate and corrects `junk x [0 in the table]` on its scalar partial; the scatter predicates
    rows): probing and scattered arrays of every length class — 1, 2, odd, just under / at / over a 128-value row, a half batch
    (512), a batch (1024), 1025, 2047, 4095 — against tables that DO and do NOT hold value 0: the other array, a bitmap
    (copied into the table), a run container of <= 600 runs (boundary masks: [0, 5]; a full first dword: [0, 200]) and a long run
    list (the streaming path).  Both orders; numpy is the expectation."""
    from featurebase_amd.roaring import Container

    rng = D.rng_for(6262)
    lens = [1, 2, 3, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 3001, 4095]

    def arr(n, zero):
        v = np.sort(rng.choice(np.arange(1, 65536), size=n - (1 if zero else 0), replace=False)).astype(np.uint16)
        return np.concatenate([[0], v]).astype(np.uint16) if zero else v

    def bits(vals):
        w = np.zeros(65536, dtype=np.uint8)
        w[np.asarray(vals, dtype=np.int64)] = 1
        return w

    others = []  # (container, bit vector)
    for zero in (False, True):
        for n in (1, 7, 130, 700, 1500, 4000):
            a = arr(n, zero)
            others.append((Container.array(a), bits(a)))
        bm = rng.random(65536) < 0.3
        bm[0] = zero
        others.append((Container.bitmap(np.packbits(bm.astype(np.uint8), bitorder="little").view(np.uint64)), bm.astype(np.uint8)))
        for runs in ([(0, 5), (40, 100), (3000, 3000), (65000, 65535)] if zero else [(1, 5), (40, 100), (3000, 3000), (65000, 65535)],
                     [(0, 200), (300, 301), (9000, 20000)] if zero else [(2, 200), (300, 301), (9000, 20000)],
                     [(0 if zero else 1, 1)] + [(10 + 64 * i, 10 + 64 * i + 20) for i in range(900)]):
            b = np.zeros(65536, dtype=np.uint8)
            for s, l in runs:
                b[s:l + 1] = 1
            others.append((Container.run(runs), b))
    rows_a, rows_b, want = [], [], []
    for n in lens:
        for zero in (False, True):
            a = arr(n, zero)
            ba = bits(a)
            for k0 in range(0, len(others), 16):
                grp = others[k0:k0 + 16]
                rows_a.append({s: Container.array(a) for s in range(len(grp))})
                rows_b.append({s: c for s, (c, _) in enumerate(grp)})
                want.append(int(sum(int((ba & bv).sum()) for _, bv in grp)))
    A, B = gpu_ctx.upload(rows_a), gpu_ctx.upload(rows_b)
    idx = np.arange(len(rows_a))
    assert gpu_ctx.intersection_count(A, idx, B, idx).tolist() == want
    assert gpu_ctx.intersection_count(B, idx, A, idx).tolist() == want
    plan = gpu_ctx.plan(A, idx, B, idx)  # (the launch-only form resolves item records: the one-wave-per-item kernel)
    plan.intersection_count()
    assert plan.read().tolist() == want
    plan.free()
    A.free()
    B.free()
