"""GPU parity for round 2's operator additions: fbk_topn (fragment.top's MinThreshold / Tanimoto
rules), fbk_topk_bsi (the BSI-encoded TopK counts) and fbk_flip (Bitmap.Flip + the golden flip
triples of the reference's container combination table)."""
import json
import os

import numpy as np
import pytest

import datagen as D
import go_fixtures as G
from featurebase_amd import lib as L
from featurebase_amd.roaring import Container

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "topn_vectors.json")))["cases"]
COMBOS = json.load(open(os.path.join(HERE, "golden", "container_combinations.json")))["ops"]


def row_of_columns(cols, key_base=0):
    """columns (0 .. 2^20) of one shard row -> {key: Container} with optimize()-style encodings left to arrays/bitmaps"""
    row = {}
    cols = np.asarray(sorted(cols), dtype=np.int64)
    for slot in np.unique(cols >> 16):
        v = (cols[(cols >> 16) == slot] & 0xFFFF).astype(np.uint16)
        row[key_base + int(slot)] = Container.array(v) if v.size < 4096 else Container.bitmap(D.words_of(v.astype(np.int64)), int(v.size))
    return row


def case_rows(c):
    if "generator" in c:
        return {i: list(range(i)) for i in range(c["generator"]["n"])}
    return {int(k): v for k, v in c["rows"].items()}


@pytest.mark.parametrize("c", VEC, ids=lambda c: c["test"])
def test_fragment_top_vectors_through_fbk_topn(gpu_ctx, c):
    """TestFragment_Top / _TopN_Intersect / _Intersect_Large / _IDs / _Tanimoto / _Zero_Tanimoto
    (fragment_internal_test.go:1148-1273, 1490-1538) through fbk_topn, both ordering paths."""
    o = c["options"]
    rows = case_rows(c)
    ids = sorted(rows) if not o["RowIDs"] else o["RowIDs"]
    n = 0 if o["RowIDs"] else o["N"]
    batch = gpu_ctx.upload([row_of_columns(rows[i]) if rows.get(i) else {} for i in ids])
    F = gpu_ctx.upload([row_of_columns(c["src"])]) if o["Src"] else None
    fargs = (F, np.zeros(1, dtype=np.uint32)) if F is not None else (None, None)
    try:
        for mode in (0, 1):
            gpu_ctx.set_option("topk_device_sort", mode)
            idx, cnt = gpu_ctx.topn(batch, np.arange(len(ids)).reshape(1, -1), n, *fargs, min_threshold=o["MinThreshold"], tanimoto_threshold=o["TanimotoThreshold"])
            assert [[ids[i], int(x)] for i, x in zip(idx.tolist(), cnt.tolist())] == c["expected"], (c["test"], mode)
    finally:
        gpu_ctx.set_option("topk_device_sort", -1)
    batch.free()
    if F is not None:
        F.free()


def test_topn_thresholds_multi_shard_vs_oracle(gpu_ctx):
    """Random rows over several shards: MinThreshold and Tanimoto rules applied per shard, counts summed over shards.
    Under topn_semantics = 1 (default) against oracle/pytopn.execute_topn (executeTopN's two passes, pinned to the
    reference's executor vectors), under 0 against top_exact; one-shot call, prepared query, and a member's partials."""
    from oracle import pytopn as T

    differ = []
    rng = np.random.default_rng(77)
    n_shards, n_a = 4, 40
    shards, srcs = [], []
    for s in range(n_shards):
        rows = {}
        for r in range(n_a):
            k = int(rng.integers(0, 4))
            m = [0, int(rng.integers(1, 30)), int(rng.integers(100, 3000)), int(rng.integers(5000, 40000))][k]
            rows[r] = sorted(set(rng.integers(0, 1 << 17, m).tolist()))
        shards.append(rows)
        srcs.append(sorted(set(rng.integers(0, 1 << 17, 20000).tolist())))
    batch = gpu_ctx.upload([row_of_columns(shards[s][r]) for s in range(n_shards) for r in range(n_a)])
    F = gpu_ctx.upload([row_of_columns(srcs[s]) for s in range(n_shards)])
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rf = np.arange(n_shards)
    ids = list(range(n_a))
    try:
        for use_src in (True, False):
            for mt, tt, n in [(0, 0, 0), (5, 0, 0), (300, 0, 7), (2000, 0, 0), (0, 10, 0), (0, 30, 5), (0, 60, 0), (0, 100, 0), (7, 20, 0), (0, 0, 1), (0, 0, 3), (40, 0, 2),
                              (0, 20, 2), (0, 0, n_a - 1), (0, 0, n_a), (0, 0, n_a + 5)]:
                if tt and not use_src:
                    continue
                ss = srcs if use_src else None
                fa = (F, rf) if use_src else (None, None)
                # topn_semantics = 1 (default): executeTopN's two passes, candidates per SHARD; = 0: the exact top n
                exp_ref = T.execute_topn(shards, n, ss, None, mt, tt)
                exp_exact = T.top_exact(shards, ids, n, ss, mt, tt)
                for sem, exp in ((1, exp_ref), (0, exp_exact)):
                    gpu_ctx.set_option("topn_semantics", sem)
                    idx, cnt = gpu_ctx.topn(batch, ra, n, *fa, min_threshold=mt, tanimoto_threshold=tt)
                    assert list(zip(idx.tolist(), [int(x) for x in cnt])) == exp, (use_src, mt, tt, n, sem)
                    q = gpu_ctx.prepare_topn(batch, ra, n, *fa, min_threshold=mt, tanimoto_threshold=tt)
                    for _ in range(2):
                        q.run()
                        qi, qc = q.read()
                        assert list(zip(qi.tolist(), [int(x) for x in qc])) == exp, ("prepared", use_src, mt, tt, n, sem)
                    q.free()
                    # a member's share: totals of its shards + the candidate flags of its shards' first pass
                    tot, cand = gpu_ctx.topn_partials(batch, ra[1:3], n_a, n, *((F, rf[1:3]) if use_src else (None, None)), min_threshold=mt, tanimoto_threshold=tt)
                    sub, subsrc = shards[1:3], (srcs[1:3] if use_src else None)
                    et = np.zeros(n_a, dtype=np.uint64)
                    for r, c in T.top_exact(sub, ids, 0, subsrc, mt, tt):
                        et[r] = c
                    assert (tot == et).all(), ("partials", use_src, mt, tt, n, sem)
                    if sem == 1 and 0 < n < n_a:
                        assert np.nonzero(cand)[0].tolist() == T.topn_candidates(sub, n, subsrc, mt, tt), ("candidates", use_src, mt, tt, n)
                    else:
                        assert ((cand != 0) == (et != 0)).all()
                if 0 < n < n_a and exp_ref != exp_exact:
                    differ.append((use_src, mt, tt, n))
    finally:
        gpu_ctx.set_option("topn_semantics", 1)
    assert differ, "the inputs never separated the reference's answer from the exact one"
    with pytest.raises(L.FbkError):
        gpu_ctx.topn(batch, ra, 0, F, rf, tanimoto_threshold=101)
    batch.free()
    F.free()


EXEC = json.load(open(os.path.join(HERE, "golden", "executor_topn_vectors.json")))


@pytest.mark.parametrize("c", EXEC["cases"], ids=lambda c: c["test"])
def test_executor_topn_vectors_through_fbk_topn(gpu_ctx, c):
    """TestExecutor_Execute_TopN / _fill / _fill_small / _Src (executor_test.go:1846-2200) through fbk_topn on the default
    semantics: _fill_small's five shards each rank another row first, n = 1 -> the candidates are {0..4} -> {0: 5}.  Also
    through a two-member group on one device (shards dealt round-robin) and the partials + reduce of the one-process-per-GPU
    deployment: the answer must not depend on the dealing."""
    from featurebase_amd import dist as fd
    from featurebase_amd.roaring import Group

    w = EXEC["shard_width"]
    n_shards = max(col // w for _, col in c["bits"]) + 1
    ids = sorted({r for r, _ in c["bits"]})
    per = [[[] for _ in ids] for _ in range(n_shards)]
    for r, col in c["bits"]:
        per[col // w][ids.index(r)].append(col % w)
    batch = gpu_ctx.upload([row_of_columns(per[s][i]) for s in range(n_shards) for i in range(len(ids))])
    ra = np.arange(n_shards * len(ids)).reshape(n_shards, len(ids))
    F, rf = None, None
    if c["src_bits"] is not None:
        F = gpu_ctx.upload([row_of_columns([x % w for x in c["src_bits"] if x // w == s]) for s in range(n_shards)])
        rf = np.arange(n_shards)
    idx, cnt = gpu_ctx.topn(batch, ra, c["n"], F, rf, min_threshold=1)
    assert [[ids[i], int(x)] for i, x in zip(idx.tolist(), cnt.tolist())] == c["expected"], c["test"]
    # two ranks' partials, reduced as featurebase_amd.dist.topn_reduce does without a process group: add them by hand
    parts = [gpu_ctx.topn_partials(batch, ra[m::2], len(ids), c["n"], F, rf[m::2] if rf is not None else None, min_threshold=1) if ra[m::2].size else
             (np.zeros(len(ids), dtype=np.uint64), np.zeros(len(ids), dtype=np.uint64)) for m in range(2)]
    i2, c2 = fd.topn_reduce(parts[0][0] + parts[1][0], parts[0][1] + parts[1][1], c["n"])
    assert [[ids[i], int(x)] for i, x in zip(i2.tolist(), c2.tolist())] == c["expected"], c["test"]
    batch.free()
    if F is not None:
        F.free()


def test_topk_counts_as_bsi_planes(gpu_ctx):
    """fbk_topk_bsi: plane p holds row id i iff bit p of the total count of row i is set
    (bsiBuilder.Insert(rowID, count), bsi.go:251-284); adding two such results with fbk_bsi_add is
    AddBSI of the per-shard results (bsi.go:83-175) = the totals over both shard sets."""
    rng = np.random.default_rng(78)
    n_shards, n_a = 3, 300
    w = D.dense_rows(n_shards * n_a, 0.5, 7801)
    w[5::7] = 0  # some empty rows
    w[[5, n_a + 5, 2 * n_a + 5]] = 0  # row index 5 is empty in every shard
    wf = D.dense_rows(n_shards, 0.5, 7802)
    A, F = gpu_ctx.upload_dense(w), gpu_ctx.upload_dense(wf)
    ra, rf = np.arange(n_shards * n_a).reshape(n_shards, n_a), np.arange(n_shards)
    tot = np.zeros(n_a, dtype=np.uint64)
    for s in range(n_shards):
        tot += np.bitwise_count(w[ra[s]] & wf[s]).sum(axis=(1, 2)).astype(np.uint64)

    def decode(batch, depth):
        got = np.zeros(n_a, dtype=np.uint64)
        rows = batch.download()
        assert len(rows) == depth
        for p, row in enumerate(rows):
            for k, c in row.items():
                assert k >> 4 == p
                bits = np.nonzero(np.unpackbits(c.words().view(np.uint8), bitorder="little"))[0] + ((k & 15) << 16)
                got[bits] |= np.uint64(1 << p)
        return got

    for flags in (0, L.SETOP_OPTIMIZE):
        out, depth = gpu_ctx.topk_bsi(A, ra, F, rf, flags)
        assert depth == int(tot.max()).bit_length()
        assert (decode(out, depth) == tot).all()
        out.free()
    # per-shard results merged with the BSI adder = the result over all shards
    o1, d1 = gpu_ctx.topk_bsi(A, ra[:1], F, rf[:1])
    o2, d2 = gpu_ctx.topk_bsi(A, ra[1:], F, rf[1:])
    s = gpu_ctx.bsi_add(o1, np.arange(d1).reshape(1, -1), o2, np.arange(d2).reshape(1, -1))
    got = np.zeros(n_a, dtype=np.uint64)
    for p, row in enumerate(s.download()):
        for k, c in row.items():
            bits = np.nonzero(np.unpackbits(c.words().view(np.uint8), bitorder="little"))[0] + ((k & 15) << 16)
            got[bits] |= np.uint64(1 << p)
    assert (got == tot).all()
    out, depth = gpu_ctx.topk_bsi(A, ra[:, 5:6], None, None)  # a field whose only row is empty: no planes
    assert depth == 0 and out.info()[1] == 0
    for b in (o1, o2, s, out, A, F):
        b.free()


def test_flip_golden_triples_and_random_ranges(gpu_ctx, oracle):
    """The 10 flip triples of TestContainerCombinations (roaring_internal_test.go:3639-3649) in all three
    encodings — the container-level flip is fbk_flip over one slot's range — and Bitmap.Flip
    (roaring.go:2769) on random rows and ranges against a numpy bit model."""
    O = oracle
    mk = {
        1: lambda p: O.OContainer.array(G.pattern_values(p).astype(np.uint16)),
        2: lambda p: O.OContainer.bitmap(G.pattern_words(p)),
        3: lambda p: O.OContainer.run(G.pattern_runs(p)),
    }
    flips = [t for t in COMBOS if t["op"] == "flip"]
    assert len(flips) == 10
    for enc in (1, 2, 3):
        for slot in (0, 7, 15):
            rows = [{i * 16 + slot: D.to_fbk(mk[enc](t["x"]))} if t["x"] != "empty" else {} for i, t in enumerate(flips)]
            batch = gpu_ctx.upload(rows)
            for flags in (0, L.SETOP_OPTIMIZE):
                out, cnt = gpu_ctx.flip(batch, np.arange(len(flips)), slot << 16, (slot << 16) + 65535, flags)
                res = out.download()
                for i, t in enumerate(flips):
                    expw = G.pattern_words(t["exp"])
                    got = [c for k, c in res[i].items() if (k & 15) == slot]
                    gw = got[0].words() if got else np.zeros(1024, dtype=np.uint64)
                    assert (gw == expw).all(), (t, enc, slot)
                    assert int(cnt[i]) == int(np.bitwise_count(expw).sum())
                    assert len(res[i]) == (1 if expw.any() else 0)  # nothing outside the flipped slot
                out.free()
            batch.free()
    rng = D.rng_for(79)
    n = 12
    rows = [D.random_row(rng, r) for r in range(n)]
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])

    def bits_of(row):
        b = np.zeros(1 << 20, dtype=np.uint8)
        for k, c in row.items():
            w = c.words() if hasattr(c, "words") else None
            b[(k & 15) << 16: ((k & 15) + 1) << 16] = np.unpackbits(w.view(np.uint8), bitorder="little")
        return b

    for start, end in [(0, (1 << 20) - 1), (0, 0), (65535, 65536), (70000, 70001), (123, 900000), ((1 << 20) - 1, (1 << 20) - 1), (65536 * 3, 65536 * 5 - 1)]:
        out, cnt = gpu_ctx.flip(batch, np.arange(n), start, end, L.SETOP_OPTIMIZE)
        res = out.download()
        for i in range(n):
            exp = bits_of(rows[i])
            exp[start: end + 1] ^= 1
            got = bits_of(res[i])
            assert (got == exp).all(), (start, end, i)
            assert int(cnt[i]) == int(exp.sum())
        out.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.flip(batch, [0], 5, 4)
    with pytest.raises(L.FbkError):
        gpu_ctx.flip(batch, [0], 0, 1 << 20)
    batch.free()


def _interval_row(lo, hi):
    """columns [lo, hi) of a shard as run containers, one interval per touched slot"""
    row = {}
    for slot in range(lo >> 16, ((hi - 1) >> 16) + 1 if hi > lo else 0):
        a, b = max(lo, slot << 16), min(hi, (slot + 1) << 16)
        row[slot] = Container.run([(a & 0xFFFF, (b - 1) & 0xFFFF)])
    return row


def test_topn_tanimoto_rule_at_shard_scale_counts_vs_float64(gpu_ctx):
    """The device's TopnRule (integer arithmetic, fbk_query_kernels.hip.h) against the LITERAL float64 expressions of fragment.top
    (fragment.go:1334-1385: float64(srcCount*t)/100, float64(srcCount*100)/float64(t), math.Ceil(float64(count*100)/float64(cnt+
    srcCount-count))) on rows whose counts reach 2^20: one shard, every row an interval of columns (run containers), the source row
    an interval too, so cnt, count and srcCount are chosen freely — including the integer neighbours of all three boundaries.
    Exact semantics, n = 0: every row is judged by the rule and reported with its count (the candidate pass of the reference's
    semantics applies the same TopnRule; its heap walk is pinned at small scale by the executor vectors)."""
    from test_oracle_topn import _literal_float64_rule

    W = 1 << 20
    rng = np.random.default_rng(0x70b9)
    for t, src_n in ((30, 600_000), (50, 333_333), (7, 1 << 20), (99, 1000), (10, 65_536)):
        rows = []  # (lo, hi)
        # boundary neighbours: cnt around src * t / 100 and src * 100 / t, count around t (cnt + src) / (100 + t)
        for cnt0 in {src_n * t // 100, -(-src_n * t // 100), src_n * 100 // t, -(-src_n * 100 // t)}:
            for cnt in (cnt0 - 1, cnt0, cnt0 + 1):
                if not 1 <= cnt <= W:
                    continue
                lo_c, hi_c = max(0, cnt + src_n - W), min(cnt, src_n)
                c0 = t * (cnt + src_n) // (100 + t)
                for count in {lo_c, hi_c, c0 - 1, c0, c0 + 1}:
                    if lo_c <= count <= hi_c:
                        rows.append((src_n - count, src_n - count + cnt))  # [lo, hi) overlaps the source [0, src_n) in `count` columns
        while len(rows) < 160:
            cnt = int(rng.integers(1, W + 1))
            lo_c, hi_c = max(0, cnt + src_n - W), min(cnt, src_n)
            count = int(rng.integers(lo_c, hi_c + 1))
            rows.append((src_n - count, src_n - count + cnt))
        rows = [r for r in rows if 0 <= r[0] and r[1] <= W]
        n_a = len(rows)
        batch = gpu_ctx.upload([_interval_row(lo, hi) for lo, hi in rows])
        F = gpu_ctx.upload([_interval_row(0, src_n)])
        ra, rf = np.arange(n_a).reshape(1, -1), np.zeros(1, dtype=np.uint32)
        exp = {}
        for i, (lo, hi) in enumerate(rows):
            cnt, count = hi - lo, max(0, min(hi, src_n) - lo)
            if _literal_float64_rule(cnt, count, src_n, True, 0, t):
                exp[i] = count
        try:
            gpu_ctx.set_option("topn_semantics", 0)
            idx, cnt = gpu_ctx.topn(batch, ra, 0, F, rf, tanimoto_threshold=t)
            assert dict(zip(idx.tolist(), [int(x) for x in cnt])) == exp, (t, src_n)
        finally:
            gpu_ctx.set_option("topn_semantics", 1)
        batch.free()
        F.free()
