"""Bytes the reference itself wrote, as shipped in its tree (tests/golden/extract_binary_fixtures.py):

  sample_view_0.roaring   testdata/sample_view/0: a 297 322-byte Pilosa fragment image (cookie 12348), 14 207
                          containers in 1000 rows — by far the largest piece of reference-written roaring data
  migrate_*_222           two 21-byte fragment files whose container table is empty and whose OPS LOG holds one add op
  cursor_add_roaring      the (name, wantChanged) table of rbf/cursor_test.go TestCursor_AddRoaring: pins which CELL a
                          container becomes (ConvertToLeafArgs: RLE cell, BitmapPtr for oversized arrays / run lists)
                          and when a merge rewrites it — the policy oracle/pyrbf.py's images follow

CPU part: the oracle's wire parser (oracle/wire_oracle.c) against an independent numpy reading of the same bytes and
against byte identity of its own writer; the RBF cursor policy restated in oracle/pyrbf.py against the reference's
expectations.  GPU part (marked): the same bytes through fbk_batch_upload_roaring / fbk_batch_download_roaring /
fbk_batch_upload_rbf and the query kernels, against the oracle."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import datagen as D

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "binary_fixtures.json")))


def sample_view() -> bytes:
    raw = open(os.path.join(HERE, "golden", "sample_view_0.roaring"), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == META["files"]["sample_view_0.roaring"]["sha256"]
    return raw


def naive_parse(raw: bytes):
    """An independent reading of the Pilosa roaring layout (roaring.go:1738-1817): 8-byte header {magic u16 = 12348,
    version/flags u16, container count u32}, count x {key u64, type u16, N-1 u16}, count x offset u32, payloads.
    -> {key: sorted values} (arrays only, which is all this file holds), payload end."""
    magic, _, n = struct.unpack_from("<HHI", raw, 0)
    assert magic == 12348
    hdr = np.frombuffer(raw, dtype=np.dtype([("key", "<u8"), ("type", "<u2"), ("n1", "<u2")]), count=n, offset=8)
    offs = np.frombuffer(raw, dtype="<u4", count=n, offset=8 + 12 * n)
    out, end = {}, 0
    for h, o in zip(hdr, offs):
        assert h["type"] == 1
        cnt = int(h["n1"]) + 1
        out[int(h["key"])] = np.frombuffer(raw, dtype="<u2", count=cnt, offset=int(o))
        end = max(end, int(o) + 2 * cnt)
    return out, end


def test_sample_view_oracle_parse_equals_naive_reading_and_rewrites_byte_identically(oracle):
    raw = sample_view()
    naive, end = naive_parse(raw)
    assert len(naive) == 14207 and end == len(raw)  # no ops log behind the containers
    bm = oracle.OBitmap.unmarshal(raw)
    items = bm.items()
    assert [k for k, _ in items] == sorted(naive) and bm.count() == 35001 == sum(v.size for v in naive.values())
    for k, c in items:
        assert c.typ == oracle.ARRAY and np.array_equal(np.asarray(c.data()), naive[k]) and (np.diff(naive[k].astype(np.int64)) > 0).all()
    assert len({k >> 4 for k in naive}) == 1000 and max(naive) >> 4 == 999
    # the reference wrote this image with Bitmap.WriteTo after Optimize(): the oracle's writer must give the same bytes
    assert bm.marshal(optimize_first=True) == raw and bm.marshal(optimize_first=False) == raw


def test_migrate_fragment_files_ops_log(oracle):
    from oracle import pywire_ops as W

    for name, want_bits in (("migrate__exists_222", {1}), ("migrate_language_222", {(1 << 20) + 1})):
        raw = bytes.fromhex(META["files"][name]["hex"])
        assert hashlib.sha256(raw).hexdigest() == META["files"][name]["sha256"]
        assert struct.unpack_from("<HHI", raw, 0) == (12348, 0, 0)  # no containers: everything is in the log
        ops = W.ops_parse(raw[8:])
        assert len(ops) == 1 and ops[0][0] == 0  # one opTypeAdd
        assert W.apply_ops(set(), ops, None) == want_bits


# ---- RBF: the cursor's cell policy ------------------------------------------------------------------------------------
def _add_roaring_cases(O):
    """the containers of TestCursor_AddRoaring's table (rbf/cursor_test.go:297-424), in order, after "no view\""""
    amax, rmax = META["ArrayMaxSize"], META["RLEMaxSize"]
    bm = lambda vals: O.OContainer.bitmap(D.words_of(np.array(vals)))  # noqa: E731  makeBitmap
    big_runs, x = [], 0
    for _ in range(rmax + 2):
        big_runs.append((x, x + 1))
        x += 3
    return [
        ("initial Array", 0, O.OContainer.array([1, 2])),
        ("initial RLE", 1, O.OContainer.run([(10, 20000)])),
        ("initial Bitmap", 3, bm([4, 8, 12])),
        ("merge Array exist", 0, O.OContainer.array([1, 2])),
        ("merge Array present", 0, O.OContainer.array([3, 4])),
        ("merge Bitmap exist", 3, bm([4, 8, 12])),
        ("merge Bitmap ", 3, bm([75])),
        ("merge BitmapArray ", 0, bm([75])),
        ("too Big Array ", 10, O.OContainer.array(list(range(amax + 2)))),
        ("too Big RLE ", 10, O.OContainer.run(big_runs)),
        ("empty container ", 11, O.OContainer.array([])),
        ("merge RLE", 1, O.OContainer.run([(1, 12)])),
    ]


def build_cursor_model(O):
    from oracle import pyrbf

    want = {c["name"]: c["wantChanged"] for c in META["cursor_add_roaring"]["cases"]}
    model = pyrbf.CursorModel()
    cases = _add_roaring_cases(O)
    assert [n for n, _, _ in cases] == [c["name"] for c in META["cursor_add_roaring"]["cases"]][1:]  # the transcription keeps the table's order
    for name, key, cont in cases:
        assert model.add_roaring([(key, cont)]) == want[name], name
    return model


def test_cursor_add_roaring_table_and_cell_types(oracle):
    from oracle import pyrbf

    model = build_cursor_model(oracle)
    cells = {k: (t, n) for k, t, n, _ in model.containers()}
    rmax = META["RLEMaxSize"]
    # key 0: array {1,2,3,4} merged with bitmap {75}: roaring.Union optimize()s {1-4, 75} (2 runs <= 5 / 2) into a RUN
    # container -> RLE cell; key 1: run [1, 20000]; key 3: {4, 8, 12, 75}: 4 runs > 4 / 2 -> array
    assert cells[0] == (3, 5) and cells[1] == (3, 20000) and cells[3] == (1, 4) and 11 not in cells
    # key 10: "too Big Array" alone is a bitmap page (4081 values > ArrayMaxSize); merged with "too Big RLE" the union is one
    # long run [0, 4081] plus 680 short ones: optimize() makes it a run container of 681 runs <= RLEMaxSize -> an RLE cell
    want10 = set(range(META["ArrayMaxSize"] + 2)) | {v for i in range(rmax + 2) for v in (3 * i, 3 * i + 1)}
    assert cells[10] == (3, len(want10)) and len([c for c in model.containers() if c[0] == 10][0][3]) == len(D.runs_of_vals(np.array(sorted(want10))))
    only_big = pyrbf.CursorModel()
    only_big.add_roaring([(10, _add_roaring_cases(oracle)[8][2])])
    assert only_big.containers()[0][1] == 2  # BitmapPtr
    # written as an RBF image and read back: same cells, RLE / BitmapPtr cell types on the page
    f = pyrbf.write_db({"x": model.containers()})
    back = pyrbf.read_bitmap(f, pyrbf.find_root(f, "x"))
    assert [(k, t, n) for k, t, n, _ in back] == [(k, t, n) for k, t, n, _ in model.containers()]
    for (_, _, _, p0), (_, _, _, p1) in zip(model.containers(), back):
        assert np.array_equal(np.asarray(p0).reshape(-1), p1.reshape(-1))


def test_cursor_rle_conversion_boundary(oracle):
    """TestCursor_RLEConversion (rbf/cursor_test.go:601-640): exactly RLEMaxSize runs stay an RLE cell
    (`CurrentPageType() == ContainerTypeRLE`) holding exactly those values; one run more becomes a bitmap."""
    from oracle import pyrbf

    rmax = META["RLEMaxSize"]
    runs = [(1 + 3 * i, 2 + 3 * i) for i in range(rmax)]
    leaf = pyrbf.convert_to_leaf(0, oracle.OContainer.run(runs))
    assert leaf[1] == 3 and leaf[2] == 2 * rmax and np.asarray(leaf[3]).tolist() == [list(r) for r in runs]
    assert 7 in {v for r in runs for v in r}  # c.Contains(0x7) in the reference test
    over = pyrbf.convert_to_leaf(0, oracle.OContainer.run(runs + [(1 + 3 * rmax, 2 + 3 * rmax)]))
    assert over[1] == 2 and over[2] == 2 * rmax + 2
    amax = META["ArrayMaxSize"]
    assert pyrbf.convert_to_leaf(0, oracle.OContainer.array(list(range(amax))))[1] == 1
    assert pyrbf.convert_to_leaf(0, oracle.OContainer.array(list(range(amax + 1))))[1] == 2


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_sample_view_through_the_abi(gpu_ctx, oracle):
    """fbk_batch_upload_roaring of the reference-written fragment image: every one of the 14 207 containers against the
    oracle's parse, the image written back byte-identically by fbk_batch_download_roaring, and the row queries a
    fragment serves (TopN counts against a filter row, row x row IntersectionCount, row counts) against the oracle."""
    from oracle import pybatch as PB

    raw = sample_view()
    batch, ids = gpu_ctx.upload_roaring(raw)
    assert ids.tolist() == list(range(1000))
    d, p, n_rows = batch.download_flat()
    assert n_rows == 1000 and len(d) == 14207 and (d["type"] == 1).all()
    naive, _ = naive_parse(raw)
    for e in d:
        assert np.array_equal(p[int(e["off"]): int(e["off"]) + 2 * int(e["len"])].view(np.uint16), naive[int(e["key"])])
    assert batch.to_roaring() == raw
    # the oracle's Bitmaps of the same rows, straight from the parse
    bm = oracle.OBitmap.unmarshal(raw)
    fr = D.FlatRows()
    for k, c in bm.items():
        fr.add(k >> 4, k, 1, np.asarray(c.data(), dtype=np.uint16), c.n)
    OA = PB.RowSet.from_flat(fr.descs(), fr.payload(), 1000)
    rows = np.arange(1000)
    assert (batch.count(rows) == OA.counts()).all()
    # row x row: each row against the next one and against row 0
    ra, rb = np.concatenate([rows[:-1], rows]), np.concatenate([rows[1:], np.zeros(1000, dtype=np.int64)])
    assert (gpu_ctx.intersection_count(batch, ra, batch, rb) == PB.intersection_count(OA, ra, OA, rb)).all()
    # TopN shape: every row against a filter row (the union of rows 0..9), one shard
    un, _ = gpu_ctx.union_n(batch, rows[:10].reshape(1, 10))
    OU, _ = PB.union_n(OA, rows[:10].reshape(1, 10))
    tot, ps = gpu_ctx.count_matrix(batch, rows.reshape(1, -1), un, np.zeros((1, 1), dtype=np.int64), per_shard=True)
    assert (ps[0, :, 0] == PB.topk_counts(OA, rows.reshape(1, -1), OU, [0])[0]).all()
    ids_k, cnt_k = gpu_ctx.topk(batch, rows.reshape(1, -1), 10, un, [0])
    exp = PB.topk_counts(OA, rows.reshape(1, -1), OU, [0])[0]
    order = sorted(range(1000), key=lambda r: (-int(exp[r]), r))[:10]
    assert sorted(cnt_k.tolist(), reverse=True) == [int(exp[r]) for r in order]
    un.free()
    batch.free()


@pytest.mark.gpu
def test_migrate_fragment_files_through_the_abi(gpu_ctx):
    for name, row, col in (("migrate__exists_222", 0, 1), ("migrate_language_222", 1, 1)):
        raw = bytes.fromhex(META["files"][name]["hex"])
        batch, ids = gpu_ctx.upload_roaring(raw)  # empty table + a one-op log, replayed on the device
        assert ids.tolist() == [row]
        r = batch.download()
        assert list(r[0]) == [row * 16] and r[0][row * 16].n == 1 and np.nonzero(np.unpackbits(r[0][row * 16].words().view(np.uint8), bitorder="little"))[0].tolist() == [col]
        batch.free()


@pytest.mark.gpu
def test_cursor_policy_image_through_upload_rbf(gpu_ctx, oracle):
    """The RBF image of the cells TestCursor_AddRoaring leaves behind (array, RLE and BitmapPtr cells as the reference's
    cursor would have written them) through fbk_rbf_find_root + fbk_batch_upload_rbf, bit for bit."""
    from oracle import pyrbf

    model = build_cursor_model(oracle)
    f = pyrbf.write_db({"x": model.containers()})
    batch, ids = gpu_ctx.upload_rbf(f, gpu_ctx.rbf_find_root(f, "x"))
    assert ids.tolist() == [0]  # keys 0, 1, 3, 10: all in row 0
    got = batch.download()[0]
    want = {k: pyrbf.leaf_to_container((k, t, n, p)) for k, t, n, p in model.containers()}
    assert sorted(got) == sorted(want)
    for k, c in got.items():
        assert c.n == want[k].n and (c.words() == want[k].words()).all(), k
    assert got[0].typ == 3 and got[1].typ == 3 and got[3].typ == 1 and got[10].typ == 3  # RLE, RLE, array, RLE
    # and a BitmapPtr cell: the oversized array on its own
    only_big = pyrbf.CursorModel()
    only_big.add_roaring([(10, _add_roaring_cases(oracle)[8][2])])
    f2 = pyrbf.write_db({"x": only_big.containers()})
    b2, _ = gpu_ctx.upload_rbf(f2, gpu_ctx.rbf_find_root(f2, "x"))
    c10 = b2.download()[0][10]
    assert c10.typ == 2 and c10.n == META["ArrayMaxSize"] + 2 and (c10.words() == D.words_of(np.arange(META["ArrayMaxSize"] + 2))).all()
    b2.free()
    batch.free()


def test_cursor_add_roaring_table_through_the_restated_writer(oracle):
    """The same (name, wantChanged) table of TestCursor_AddRoaring, this time through the line-by-line restatement of
    the reference's page WRITER (oracle/pyrbf_writer.py: Seek, putLeafCell fast and slow paths, bitmap-page allocation
    and release, merge): every call reports what the reference's test expects, and the pages it leaves hold the cells the
    policy model predicts — read back by the oracle's page reader."""
    from oracle import pyrbf, pyrbf_writer as W

    want = {c["name"]: c["wantChanged"] for c in META["cursor_add_roaring"]["cases"]}
    db = W.RbfDb()
    db.create_bitmap("x")
    for name, key, cont in _add_roaring_cases(oracle):
        assert db.add_roaring("x", [(key, cont)]) == want[name], name
    db.commit()
    img = db.image()
    back = pyrbf.read_bitmap(img, pyrbf.find_root(img, "x"))
    model = build_cursor_model(oracle)
    assert [(k, t, n) for k, t, n, _ in back] == [(k, t, n) for k, t, n, _ in model.containers()]
    for (_, _, _, p0), (_, _, _, p1) in zip(model.containers(), back):
        assert np.array_equal(np.asarray(p0).reshape(-1), p1.reshape(-1))
    # "too Big Array" took a bitmap page that "too Big RLE" (the union is an RLE cell) gave back: it is on the freelist or
    # was truncated off the end of the file by Commit
    assert db.page_n <= 6
