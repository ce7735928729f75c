"""GPU parity for fbk_batch_upload_rbf: RBF file image -> device resident rows, against the
oracle's page reader; the reference-written fixture file; malformed trees are rejected."""
import json
import os

import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L
from test_oracle_rbf import FIX, file_image, fixture_file, random_fragment

pytestmark = pytest.mark.gpu


def test_reference_written_file_on_gpu(gpu_ctx):
    f = fixture_file()
    root = gpu_ctx.rbf_find_root(f, FIX["bitmap"])
    assert root == FIX["expect"]["root_pgno"]
    batch, ids = gpu_ctx.upload_rbf(f, root)
    rows = batch.download()
    assert ids.tolist() == [0] and list(rows[0]) == [0]
    c = rows[0][0]
    assert c.typ == L.TYPE_ARRAY and c.n == 1 and c.data.tolist() == [100]
    batch.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.rbf_find_root(f, "nope")  # ErrBitmapNotFound


def test_every_reference_written_file_on_gpu(gpu_ctx):
    """The four database files the reference ships, through fbk_rbf_find_root + fbk_batch_upload_rbf:
    three hold bitmap "x" = {key 0: array [100]}; bad-bitmap's branch cell points outside the file."""
    for name, d in FIX["files"].items():
        f = file_image(name)
        root = gpu_ctx.rbf_find_root(f, FIX["bitmap"])
        assert root == 3, name
        if "expect" in d:
            batch, ids = gpu_ctx.upload_rbf(f, root)
            rows = batch.download()
            assert ids.tolist() == [0] and list(rows[0]) == [0], name
            c = rows[0][0]
            assert c.typ == L.TYPE_ARRAY and c.n == 1 and c.data.tolist() == [100], name
            batch.free()
        else:
            with pytest.raises(L.FbkError, match="65537"):
                gpu_ctx.upload_rbf(f, root)


def test_bad_bitmap_file_is_rejected(gpu_ctx):
    """rbf/testdata/check/bad-bitmap: a branch cell points at pgno 65537 of a 4-page file
    (rbf/tx_test.go:1293-1305)."""
    f = b"".join(bytes.fromhex(h).ljust(8192, b"\0") for h in FIX["bad_bitmap_pages_hex_prefix"])
    root = gpu_ctx.rbf_find_root(f, "x")
    with pytest.raises(L.FbkError, match="out of bounds"):
        gpu_ctx.upload_rbf(f, root)


def test_fragment_trees_vs_oracle_reader(gpu_ctx, oracle):
    from oracle import pyrbf

    rng = D.rng_for(99)
    frag = random_fragment(rng, 60, oracle)
    other = random_fragment(rng, 2, oracle)
    f = pyrbf.write_db({"i/f/standard/0": frag, "i/g/standard/7": other}, leaf_cells_per_page=7, branch_fanout=5)
    for name, conts in (("i/f/standard/0", frag), ("i/g/standard/7", other)):
        root = gpu_ctx.rbf_find_root(f, name)
        assert root == pyrbf.find_root(f, name)
        batch, ids = gpu_ctx.upload_rbf(f, root)
        assert ids.tolist() == sorted({k >> 4 for k, _, _, _ in conts})
        rows = batch.download()
        got = {k: c for row in rows for k, c in row.items()}
        assert sorted(got) == [k for k, _, _, _ in conts]
        for k, t, n, payload in conts:
            c = got[k]
            assert (c.typ, c.n) == (t, n), k
            assert np.array_equal(np.asarray(c.data).reshape(-1), np.asarray(payload).reshape(-1)), k
        # uploaded rows behave like any other batch rows: row counts = sum of BitN
        cnt = batch.count(np.arange(len(ids)))
        for i, rid in enumerate(ids):
            assert int(cnt[i]) == sum(n for k, _, n, _ in conts if k >> 4 == int(rid))
        # and the RBF image re-serialises as the Pilosa roaring image of the same containers
        items = []
        for k, t, n, payload in conts:
            oc = {1: oracle.OContainer.array, 3: lambda p: oracle.OContainer.run([tuple(x) for x in np.asarray(p).reshape(-1, 2)]),
                  2: oracle.OContainer.bitmap}[t](payload)
            items.append((k, oc))
        assert batch.to_roaring() == oracle.OBitmap.from_containers(items).marshal(False)
        batch.free()


def test_fragment_cache_hit_miss_invalidate_evict(gpu_ctx, oracle):
    """Device fragment cache: resident fragments keyed by the RBF bitmap name + a write
    version; pin / release, stale versions, prefix invalidation, LRU eviction by bytes."""
    from oracle import pyrbf

    ctx = gpu_ctx
    ctx.cache_invalidate("")
    base = ctx.cache_stats()
    rng = D.rng_for(101)
    frags = {f"i/f/standard/{s}": random_fragment(rng, 6, oracle) for s in range(4)}
    f = pyrbf.write_db(frags)
    sizes = {}
    for name in frags:
        batch, ids = ctx.upload_rbf(f, ctx.rbf_find_root(f, name))
        ctx.cache_put(name, 7, batch, ids)
    st = ctx.cache_stats()
    assert st["entries"] == 4 and st["bytes"] > 0
    # hit: same version; the pinned batch works like any batch
    got = ctx.cache_get("i/f/standard/2", 7)
    assert got is not None
    b2, ids2 = got
    assert ids2.tolist() == sorted({k >> 4 for k, _, _, _ in frags["i/f/standard/2"]})
    cnt = b2.count(np.arange(len(ids2)))
    assert int(cnt.sum()) == sum(n for _, _, n, _ in frags["i/f/standard/2"])
    # miss: unknown key, and a newer version (the fragment was written) drops the stale entry
    assert ctx.cache_get("i/f/standard/9", 7) is None
    assert ctx.cache_get("i/f/standard/1", 8) is None
    assert ctx.cache_stats()["entries"] == 3
    # invalidating a pinned entry keeps it alive until release
    assert ctx.cache_invalidate("i/f/standard/2") == 1
    assert int(b2.count(np.arange(len(ids2))).sum()) == int(cnt.sum())
    ctx.cache_release(b2)
    with pytest.raises(L.FbkError):
        ctx.cache_release(b2)  # no longer a cache entry
    assert ctx.cache_get("i/f/standard/2", 7) is None
    # prefix invalidation of the rest of the field
    assert ctx.cache_invalidate("i/f/") == 2
    assert ctx.cache_stats()["entries"] == 0
    # LRU eviction: cap below two entries keeps only the most recently used
    for name in list(frags)[:3]:
        batch, ids = ctx.upload_rbf(f, ctx.rbf_find_root(f, name))
        ctx.cache_put(name, 1, batch, ids)
    one = ctx.cache_stats()["bytes"] // 3
    g0 = ctx.cache_get("i/f/standard/0", 1)  # touch 0: now most recent
    ctx.cache_release(g0[0])
    ctx.cache_configure(int(one * 1.5))
    st = ctx.cache_stats()
    assert st["entries"] == 1 and st["evictions"] - base["evictions"] == 2
    assert ctx.cache_get("i/f/standard/1", 1) is None
    g0 = ctx.cache_get("i/f/standard/0", 1)
    assert g0 is not None
    ctx.cache_release(g0[0])
    ctx.cache_configure(128 << 30)
    ctx.cache_invalidate("")


def test_corrupt_trees_and_untrusted_headers(gpu_ctx, oracle):
    """a branch cell that points back at an ancestor must be rejected (not walked
    cell_n^16 times); a leaf cell whose BitN under-reports is repaired by the device recount, an
    unsorted array cell is rejected by the device validator."""
    import struct

    from oracle import pyrbf

    rng = D.rng_for(103)
    frag = random_fragment(rng, 30, oracle)
    f = bytearray(pyrbf.write_db({"i/f/standard/0": frag}, leaf_cells_per_page=4, branch_fanout=3))
    root = pyrbf.find_root(bytes(f), "i/f/standard/0")
    assert struct.unpack_from(">I", f, root * 8192 + 4)[0] == pyrbf.BRANCH
    # first branch cell of the root: childPgno (little endian u32 at cell + 12) -> the root itself
    cell0 = struct.unpack_from(">H", f, root * 8192 + 10)[0]
    cyc = bytearray(f)
    struct.pack_into("<I", cyc, root * 8192 + cell0 + 12, root)
    with pytest.raises(L.FbkError, match="reachable twice"):
        gpu_ctx.upload_rbf(bytes(cyc), root)
    # find a leaf page with an array cell and damage it
    # (walk the tree: a bitmap data page has no header and its bytes can look like one)
    leaves, todo = [], [root]
    while todo:
        pg = todo.pop()
        flags, cell_n = struct.unpack_from(">IH", f, pg * 8192 + 4)
        if flags == pyrbf.BRANCH:
            todo.extend(struct.unpack_from("<I", f, pg * 8192 + struct.unpack_from(">H", f, pg * 8192 + 10 + 2 * i)[0] + 12)[0] for i in range(cell_n))
        else:
            assert flags == pyrbf.LEAF
            leaves.append(pg)
    for pg in sorted(leaves):
        cell_n = struct.unpack_from(">H", f, pg * 8192 + 8)[0]
        for i in range(cell_n):
            off = struct.unpack_from(">H", f, pg * 8192 + 10 + 2 * i)[0]
            key, typ, elem_n, bit_n = struct.unpack_from("<QIHI", f, pg * 8192 + off)
            if typ == 1 and elem_n >= 4:
                under = bytearray(f)
                struct.pack_into("<I", under, pg * 8192 + off + 14, 1)  # BitN = 1: the array holds more
                batch, ids = gpu_ctx.upload_rbf(bytes(under), root)
                got = {k: c for row in batch.download() for k, c in row.items()}
                assert got[key].n == elem_n  # recounted on the device
                batch.free()
                swapped = bytearray(f)
                a, b = struct.unpack_from("<HH", f, pg * 8192 + off + 18)
                struct.pack_into("<HH", swapped, pg * 8192 + off + 18, b, a)
                with pytest.raises(L.FbkError, match="ascending"):
                    gpu_ctx.upload_rbf(bytes(swapped), root)
                return
    raise AssertionError("no array cell found")


def test_image_written_by_the_restated_reference_writer(gpu_ctx, oracle):
    """fbk_rbf_find_root + fbk_batch_upload_rbf on a database image produced by the line-by-line restatement of the
    reference's page WRITER (oracle/pyrbf_writer.py, pinned byte for byte to the file the reference ships): a three-level
    b-tree whose root turned into a branch of branches, leaf pages left by putLeafCell's splits and in-place rewrites, RLE
    and BitmapPtr cells, bitmap pages reused after a free, root records chained over an overflow page.  The device result
    must equal the oracle's page reader AND what was put in (the cursor-level contents)."""
    from oracle import pyrbf
    from test_oracle_rbf import build_multi_page_db

    db, expect = build_multi_page_db(oracle, n_names=60)
    img = db.image()
    for name, conts in expect.items():
        root = gpu_ctx.rbf_find_root(img, name)  # (names on the second root-record page included)
        assert root == pyrbf.find_root(img, name) == db.records[name]
        ref = pyrbf.read_bitmap(img, root)
        batch, ids = gpu_ctx.upload_rbf(img, root)
        assert ids.tolist() == sorted({k >> 4 for k in conts})
        got = {k: c for row in batch.download() for k, c in row.items()}
        assert sorted(got) == sorted(conts) == [k for k, _, _, _ in ref]
        for k, t, n, payload in ref:
            c = got[k]
            assert (c.typ, c.n) == (t, n), (name, k)
            assert np.array_equal(np.asarray(c.data).reshape(-1), np.asarray(payload).reshape(-1)), (name, k)
            assert c.n == conts[k].n and (c.words() == conts[k].words()).all(), (name, k)
        batch.free()
    types = {t for _, t, _, _ in pyrbf.read_bitmap(img, db.records["i/f/standard/0"])}
    assert types == {1, 2, 3}
