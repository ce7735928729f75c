"""Device memory of materialised results (round 4's advisor finding): kernels that apply Container.optimize() themselves write
the encoded bytes into the head of 8 KiB cells, so a result batch used to keep n_rows x 16 x 8 KiB on the device however small
its containers were.  One-shot calls now hand out a compacted batch (option setop_compact = 1, fbk_batch_compact for the rest);
outputs of plans stay borrowed cells.  Also: fbk_batch_memory reports a batch's heavy-row shadows and why it has none."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L
from oracle import pybatch as PB

pytestmark = pytest.mark.gpu


def _words(batch):
    d, p, n_rows = batch.download_flat()
    return PB.RowSet.from_flat(d, p, n_rows).words(), d, p


def _check_arena(batch, cells, n_containers_max):
    """compacted unless there is less than 1 MiB to gain (compact_cells' rule): then the arena is the payload + 16-byte padding"""
    arena, payload = batch.memory()[0], batch.info()[2]
    if payload + (1 << 20) + 16 * n_containers_max >= cells:
        assert arena in (cells, ) or payload <= arena <= payload + 16 * n_containers_max, (arena, cells, payload)
    else:
        assert payload <= arena <= payload + 16 * n_containers_max, (arena, cells, payload)
    return arena


def test_one_shot_results_are_compacted_and_identical(gpu_ctx):
    rows, g, filt = D.config3_flat(6, 16, seed_idx=8300, workers=1)
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    ia, ib = g[:, :8].reshape(-1), g[:, 8:].reshape(-1)
    n = ia.size
    cells = n * 16 * 8192
    try:
        for op in (L.OP_AND, L.OP_OR, L.OP_XOR, L.OP_ANDNOT):
            gpu_ctx.set_option("setop_compact", 0)
            o0, c0 = gpu_ctx.setop(op, batch, ia, batch, ib, L.SETOP_OPTIMIZE)
            gpu_ctx.set_option("setop_compact", 1)
            o1, c1 = gpu_ctx.setop(op, batch, ia, batch, ib, L.SETOP_OPTIMIZE)
            w0, d0, p0 = _words(o0)
            w1, d1, p1 = _words(o1)
            assert (c0 == c1).all() and (w0 == w1).all() and d0.tobytes() == d1.tobytes() and p0.tobytes() == p1.tobytes(), op
            a0, a1 = o0.memory()[0], o1.memory()[0]
            assert a0 == cells, (op, a0, cells)  # the kernel's cells
            _check_arena(o1, cells, n * 16)
            if op == L.OP_AND:
                assert a1 < a0 // 2, (a0, a1)  # intersections of mixed rows are small: this one must have shrunk
            # the explicit call on the uncompacted batch: same arena size, same content; a second call is a no-op
            assert o0.compact() == a1 and o0.compact() == a1
            w2, d2, p2 = _words(o0)
            assert (w2 == w1).all() and d2.tobytes() == d1.tobytes() and p2.tobytes() == p1.tobytes()
            # a compacted batch is a normal operand
            assert (gpu_ctx.intersection_count(o0, np.arange(n), o1, np.arange(n)) == c1).all()
            o0.free()
            o1.free()
        # the n-way fold and Flip hand out compacted batches too
        un, _ = gpu_ctx.union_n(batch, g, L.SETOP_OPTIMIZE)
        _check_arena(un, g.shape[0] * 16 * 8192, g.shape[0] * 16)
        un.free()
        fl, _ = gpu_ctx.flip(batch, np.arange(8), 5, 70000, L.SETOP_OPTIMIZE)
        _check_arena(fl, 8 * 16 * 8192, 8 * 16)
        fl.free()
    finally:
        gpu_ctx.set_option("setop_compact", 1)
    # a plan's output is borrowed: rewritten in place by the next run, never compacted
    plan = gpu_ctx.plan(batch, ia, batch, ib)
    plan.setop(L.OP_AND, L.SETOP_OPTIMIZE)
    O = plan.output()
    with pytest.raises(L.FbkError):
        O.compact()
    plan.free()
    batch.free()


def test_batch_memory_reports_shadows_and_the_arena_rule(gpu_ctx):
    """Run-heavy batch: a few bytes per container, 8 KiB per shadow (128 rows x 16 run containers: 16 MiB of shadows for a
    32 KB arena).  Under the default rule (8 x max(arena, 1 MiB)) it is refused and the state says so (2); with the rule off
    (matrix_shadow_arena_x = 0) the same content is shadowed (1), the bytes are reported, and the counts are the same."""
    from featurebase_amd.roaring import Container

    rows = [{s: Container.run([(100 * k + r, 100 * k + r + 50) for k in range(3)]) for s in range(16)} for r in range(128)]
    ra, rb = np.arange(64).reshape(1, 64), np.arange(64, 128).reshape(1, 64)
    b = gpu_ctx.upload(rows)
    assert b.memory()[1:] == (0, 0)
    ref = gpu_ctx.count_matrix(b, ra, b, rb)
    arena, sh, st = b.memory()
    assert st == 2 and sh == 0 and arena <= 128 * 16 * 16
    gpu_ctx.set_option("matrix_shadow_arena_x", 0)
    try:
        b2 = gpu_ctx.upload(rows)
        got = gpu_ctx.count_matrix(b2, ra, b2, rb)
        arena2, sh2, st2 = b2.memory()
        assert st2 == 1 and sh2 >= 128 * 16 * 8192 and arena2 == arena and (got == ref).all()
        b2.free()
    finally:
        gpu_ctx.set_option("matrix_shadow_arena_x", 8)
    assert int(ref[0, 0]) == 16 * 3 * 51 - 0 or ref.sum() > 0
    b.free()
