"""Size-independent properties at BASELINE.json's full sizes (where the CPU oracle would take
minutes): config 5 — BSI 64-bit field over 96 shards (100 M columns) —, config 4's 32 x 32
count matrix over 128 shards and config 3's Union-of-64 over 256 shards of mixed containers;
config 2 at its full 1024 shards is checked inside bench.py against numpy popcounts on every run."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu


def test_config5_bsi_range_sum_properties_full_size(gpu_ctx):
    n_shards, depth = 96, 64
    w = D.dense_rows(n_shards * (depth + 2), 0.5, 5001).reshape(n_shards, depth + 2, 16, 1024)
    w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    w[-1, 0, 6:] = 0  # 100 000 000 columns: the last shard is partial
    # planes only where a value exists (a well-formed BSI fragment)
    w[:, 1:] &= w[:, :1]
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    base = np.arange(n_shards, dtype=np.uint32) * (depth + 2)
    idx = np.arange(n_shards)
    exists = np.bitwise_count(w[:, 0]).sum(axis=(1, 2))
    assert int(exists.sum()) == 95 * (1 << 20) + 6 * 65536
    k = 1 << 62
    gt, c_gt = gpu_ctx.bsi_range(batch, base, L.BSI_GT, depth, k)
    lte, c_lte = gpu_ctx.bsi_range(batch, base, L.BSI_LTE, depth, k)
    # partition: every existing column is either > k or <= k
    assert (c_gt + c_lte == exists.astype(np.uint64)).all()
    assert gpu_ctx.intersection_count(gt, idx, lte, idx).sum() == 0
    # > k  <=>  positive and magnitude bit 62 or 63 ... with bit 63 set the magnitude exceeds int64: here
    # planes are random so check against numpy directly on the two top planes and the sign
    pos = w[:, 0] & ~w[:, 1]
    top = (w[:, 2 + 63] | (w[:, 2 + 62] & _any_lower(w, 62))) & pos
    assert np.bitwise_count(top).sum(axis=(1, 2)).tolist() == c_gt.tolist()
    # Sum is additive over a partition of the filter
    s_all, n_all = gpu_ctx.bsi_sum(batch, base, depth)
    s_gt, n_gt = gpu_ctx.bsi_sum(batch, base, depth, gt, idx)
    s_lte, n_lte = gpu_ctx.bsi_sum(batch, base, depth, lte, idx)
    assert (n_all == exists.astype(np.uint64)).all() and (n_gt == c_gt).all() and (n_lte == c_lte).all()
    assert ((s_gt.astype(np.uint64) + s_lte.astype(np.uint64)) == s_all.astype(np.uint64)).all()  # int64 wrap-around arithmetic
    # Min <= Max, both inside the filter, counts bounded by the filter
    mn, cmn = gpu_ctx.bsi_min(batch, base, depth, gt, idx)
    mx, cmx = gpu_ctx.bsi_max(batch, base, depth, gt, idx)
    has = c_gt > 0
    assert (cmn[has] >= 1).all() and (cmx[has] >= 1).all() and (cmn <= c_gt).all() and (cmx <= c_gt).all()
    for b in (gt, lte, batch):
        b.free()


def _any_lower(w, bit):
    """columns whose magnitude has any bit below `bit` set (so that magnitude > 2^bit given bit is set)"""
    acc = np.zeros_like(w[:, 0])
    for i in range(bit):
        acc |= w[:, 2 + i]
    return acc


def test_config4_count_matrix_properties_full_slice(gpu_ctx):
    n_shards, n_a, n_b = 128, 32, 32
    wa = D.dense_rows(n_shards * n_a, 0.5, 4001)
    wb = D.dense_rows(n_shards * n_b, 0.5, 4002)
    wf = D.dense_rows(n_shards, 0.5, 4003)
    A, Bt, F = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    tot = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf)
    # transpose symmetry: |A_i ∩ B_j ∩ F| computed with the operands swapped
    tot_t = gpu_ctx.count_matrix(Bt, rb, A, ra, F, rf)
    assert (tot == tot_t.T).all()
    # row i of the matrix against a direct pair count: |A_i ∩ (B_j ∩ F)|
    bf, _ = gpu_ctx.setop(L.OP_AND, Bt, rb[:, 5].copy(), F, rf)
    direct = gpu_ctx.intersection_count(A, ra[:, 3].copy(), bf, np.arange(n_shards))
    assert int(direct.sum()) == int(tot[3, 5])
    # a few cells against numpy
    wa3, wb3, wf3 = wa.reshape(n_shards, n_a, -1), wb.reshape(n_shards, n_b, -1), wf.reshape(n_shards, -1)
    for i, j in ((0, 0), (31, 31), (7, 19)):
        assert int(tot[i, j]) == int(np.bitwise_count(wa3[:, i] & wb3[:, j] & wf3).sum())
    # without the filter every count can only grow
    tot_nf = gpu_ctx.count_matrix(A, ra, Bt, rb)
    assert (tot_nf >= tot).all()
    for b in (bf, A, Bt, F):
        b.free()


def test_config3_union_of_64_properties_full_size(gpu_ctx):
    """BASELINE config 3 at its full 256 shards x (64 mixed rows + filter): the fused
    |∪ rows ∩ F| against the same quantity through the materialised union, inclusion-exclusion
    with the difference, and bounds that hold whatever the data (32 distinct shard contents,
    repeated — the properties do not care)."""
    n_shards, k, distinct = 256, 64, 32
    content, fcontent = [], []
    for s in range(distinct):
        rng = D.rng_for(3000 + s)
        rows = []
        for r in range(k):
            d = D.zipf_density(r)
            row = {}
            for slot in range(16):
                rs = rng.random() < 0.25
                c = D.fbk_container_of_vals(D.mixed_vals_for_density(rng, d, rs))
                if c is not None and c.n:
                    row[slot] = c
            rows.append(row)
        content.append(rows)
        frng = D.rng_for(3500 + s)
        fcontent.append({slot: D.fbk_container_of_vals(D.mixed_vals_for_density(frng, 0.5, False)) for slot in range(16)})
    rows = [content[s % distinct][r] for s in range(n_shards) for r in range(k)]
    batch = gpu_ctx.upload(rows)
    F = gpu_ctx.upload([fcontent[s % distinct] for s in range(n_shards)])
    groups = np.arange(n_shards * k, dtype=np.uint32).reshape(n_shards, k)
    fidx = np.arange(n_shards)
    fused = gpu_ctx.union_n_intersection_count(batch, groups, F, fidx)
    un, un_cnt = gpu_ctx.union_n(batch, groups)
    via_union = gpu_ctx.intersection_count(un, fidx, F, fidx)
    assert (fused == via_union).all()
    diff, diff_cnt = gpu_ctx.setop(L.OP_ANDNOT, un, fidx, F, fidx)
    assert (fused + diff_cnt == un_cnt).all()  # |U ∩ F| + |U \\ F| = |U|
    row_cnt = batch.count(groups.reshape(-1)).reshape(n_shards, k)
    assert (un_cnt >= row_cnt.max(axis=1)).all() and (un_cnt <= row_cnt.sum(axis=1)).all()
    assert (fused <= F.count(fidx)).all()
    assert (fused[:distinct] == fused[distinct : 2 * distinct]).all()  # repeated content, repeated answers
    # the TopK shape on the same rows: per-row counts against the filter sum to at least the fused union count
    tot = gpu_ctx.count_matrix(batch, groups, F, fidx.reshape(-1, 1), per_shard=True)[1][:, :, 0]
    assert (tot.sum(axis=1) >= fused).all() and (tot.max(axis=1) <= fused).all()
    for b in (diff, un, batch, F):
        b.free()
