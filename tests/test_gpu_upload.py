"""The bulk upload path (fbk_batch_upload / fbk_batch_upload_dense): the arena image is assembled by several host threads
in two pinned buffers while the other buffer's DMA runs, and the container CONTENT is validated on the device after the
copy.  Round trips must be byte-exact whatever the buffer size and thread count (containers straddle buffer and slice
boundaries), and malformed content must still be refused."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture()
def upload_opts(gpu_ctx):
    yield gpu_ctx
    gpu_ctx.set_option("upload_chunk_mb", 64)
    gpu_ctx.set_option("upload_threads", 0)


@pytest.mark.parametrize("chunk_mb,threads", [(1, 1), (1, 7), (2, 3), (64, 0)])
def test_flat_upload_round_trip_across_buffer_and_slice_boundaries(upload_opts, chunk_mb, threads):
    ctx = upload_opts
    rows, groups, filt = D.config3_flat(3, 40, seed_idx=9100, workers=1)  # ~4.5 MB of mixed containers
    d, p = rows.descs(), rows.payload()
    ctx.set_option("upload_chunk_mb", chunk_mb)
    ctx.set_option("upload_threads", threads)
    b = ctx.upload_flat(d, p, rows.n_rows)
    d2, p2, n_rows = b.download_flat()
    assert n_rows == rows.n_rows and len(d2) == len(d)
    # same containers in the same (row, key) order with the same bytes (offsets differ: the download packs without padding)
    o = np.lexsort((d["key"], d["row"]))
    assert (d2["row"] == d["row"][o]).all() and (d2["key"] == d["key"][o]).all() and (d2["type"] == d["type"][o]).all()
    assert (d2["len"] == d["len"][o]).all() and (d2["n"] == d["n"][o]).all()
    for i, j in enumerate(o[:: max(1, len(o) // 400)]):  # a sample of payloads byte for byte, plus the totals below
        k = i * max(1, len(o) // 400)
        nb = {1: 2 * int(d["len"][j]), 2: 8192, 3: 4 * int(d["len"][j])}[int(d["type"][j])]
        assert p2[int(d2["off"][k]) : int(d2["off"][k]) + nb].tobytes() == p[int(d["off"][j]) : int(d["off"][j]) + nb].tobytes(), (k, j)
    assert int(b.count(np.arange(rows.n_rows)).sum()) == int(d["n"].sum())
    # and the bits are what they should be: every row against numpy
    got = np.zeros((rows.n_rows, 16, 1024), dtype=np.uint64)
    for r, row in enumerate(b.download()):
        for key, c in row.items():
            got[r, key & 15] = c.words()
    from oracle import pybatch as PB

    assert (got == PB.RowSet.from_flat(d, p, rows.n_rows).words()).all()
    b.free()


@pytest.mark.parametrize("chunk_mb,threads", [(1, 5), (64, 0)])
def test_dense_upload_round_trip(upload_opts, chunk_mb, threads):
    ctx = upload_opts
    ctx.set_option("upload_chunk_mb", chunk_mb)
    ctx.set_option("upload_threads", threads)
    w = D.dense_rows(37, 0.5, 9200)  # 4.6 MB: several 1 MB buffers, a ragged last one
    w[5, 3] = 0
    b = ctx.upload_dense(w)
    assert (b.count(np.arange(37)) == np.bitwise_count(w).sum(axis=(1, 2))).all()
    got = np.zeros_like(w)
    for r, row in enumerate(b.download()):
        for key, c in row.items():
            got[r, key & 15] = c.words()
    assert (got == w).all()
    b.free()


def test_malformed_content_is_refused_by_the_device_check(gpu_ctx):
    from featurebase_amd.roaring import Container

    ok = Container.array([1, 5, 9])
    bad_arr = Container(L.TYPE_ARRAY, np.array([5, 5, 9], dtype=np.uint16), 3)       # not strictly ascending
    bad_run = Container(L.TYPE_RUN, np.array([[10, 20], [20, 30]], dtype=np.uint16), 22)  # overlapping
    bad_run2 = Container(L.TYPE_RUN, np.array([[30, 20]], dtype=np.uint16), 1)         # last < start
    for bad, msg in ((bad_arr, "ascending"), (bad_run, "unordered"), (bad_run2, "unordered")):
        with pytest.raises(L.FbkError) as e:
            gpu_ctx.upload([{0: ok, 1: bad}, {16: ok}])
        assert e.value.code == L.FBK_E_INVALID and msg in str(e.value)
    # a caller's wrong cardinality for a bitmap / run is replaced by the recount, as bitmapRepair would (roaring.go:4193)
    c = Container.run([(3, 9)], 1234)
    b = gpu_ctx.upload([{0: c}])
    assert int(b.count([0])[0]) == 7
    b.free()


def test_a_pool_cap_smaller_than_the_working_set_evicts_and_stays_correct(monkeypatch):
    """FBK_POOL_MAX_BYTES: freed device blocks are cached up to the cap; the block just freed is kept and older, larger ones are
    given back.  With a cap of a few MiB every call below frees more than fits, so blocks are evicted and re-allocated
    between calls; the results must not change."""
    from featurebase_amd.roaring import Context

    monkeypatch.setenv("FBK_POOL_MAX_BYTES", str(6 << 20))
    ctx = Context(0)
    try:
        want = None
        for it in range(6):
            n = 24 + 8 * (it % 3)  # 3, 4, 5 MiB of bitmap cells: different pool buckets on successive calls
            w = D.dense_rows(n, 0.5, 9300 + (it % 3))
            b = ctx.upload_dense(w)
            out, counts = ctx.setop(L.OP_AND, b, np.arange(n - 1), b, np.arange(1, n))
            ref = np.bitwise_count(w[:-1] & w[1:]).sum(axis=(1, 2))
            assert (counts == ref).all(), it
            if it % 3 == 0:
                if want is None:
                    want = counts.copy()
                assert (counts == want).all()
            out.free()
            b.free()
    finally:
        ctx.close()
