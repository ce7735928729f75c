"""Size-independent properties at BASELINE.json's full sizes (where the CPU oracle would take
minutes): config 5 — BSI 64-bit field over 96 shards (100 M columns) — and config 4's 32 x 32
count matrix over 128 shards; config 2 at its full 1024 shards is checked inside bench.py
against numpy popcounts on every run."""
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
