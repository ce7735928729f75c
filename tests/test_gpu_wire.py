"""GPU parity for the serialised-roaring entry points: fbk_batch_upload_roaring reads what
Bitmap.WriteTo writes (and the official RoaringBitmap format, with the reference's own
fixtures), fbk_batch_download_roaring writes byte for byte what the oracle's restatement of
Bitmap.WriteTo writes."""
import json
import os

import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "wire_fixtures.json")))


def batch_bits(batch, row_ids):
    out = []
    for r, row in enumerate(batch.download()):
        for k, c in sorted(row.items()):
            assert k >> 4 == int(row_ids[r])
            bits = np.unpackbits(c.words().view(np.uint8), bitorder="little")
            out.extend((k << 16) + int(v) for v in np.nonzero(bits)[0])
    return out


@pytest.mark.parametrize("fx", FIX["ok"], ids=lambda f: f"{len(f['hex']) // 2}B")
def test_reference_fixtures_upload(gpu_ctx, oracle, fx):
    """TestUnmarshalRoaringWithNoErrors (roaring_internal_test.go:3793): official format, with
    and without runs; count and bits as the reference expects."""
    raw = bytes.fromhex(fx["hex"])
    batch, ids = gpu_ctx.upload_roaring(raw)
    got = batch_bits(batch, ids)
    assert len(got) == fx["count"]
    assert int(batch.count(np.arange(len(ids))).sum()) == fx["count"]
    assert got == (fx["bits"] if fx["bits"] is not None else oracle.OBitmap.unmarshal(raw).slice())
    batch.free()


def test_malformed_images_are_rejected(gpu_ctx):
    for e in FIX["errors"]:
        with pytest.raises(L.FbkError):
            gpu_ctx.upload_roaring(bytes.fromhex(e["hex"]))
    for bad in (b"\x3c\x30", bytes.fromhex("3C30010000000000"), bytes.fromhex("FFFF000000000000"),
                bytes.fromhex("3C30000001000000") + b"\0" * 8):  # 1 container announced, truncated
        with pytest.raises(L.FbkError):
            gpu_ctx.upload_roaring(bad)
    batch, ids = gpu_ctx.upload_roaring(bytes.fromhex(FIX["pilosa_empty_ok"][0]))
    assert len(ids) == 0 and batch.info()[1] == 0
    batch.free()


def test_pilosa_round_trip_and_ops_on_uploaded_image(gpu_ctx, oracle):
    """Fragment-storage shaped image (key = row*16 + slot, mixed encodings, ragged rows):
    upload, operate, download; the downloaded bytes equal the oracle's Bitmap.WriteTo."""
    O = oracle
    rng = D.rng_for(93)
    n_rows = 9
    items = []
    rows = []
    for r in range(n_rows):
        row = D.random_row(rng, 0)  # keys 0..15
        rid = r * 3 + 1  # sparse row ids
        rows.append({rid * 16 + (k & 15): c for k, c in row.items()})
        items.extend(rows[-1].items())
    bm = O.OBitmap.from_containers(items)
    raw = bm.marshal(True)
    batch, ids = gpu_ctx.upload_roaring(raw)
    assert ids.tolist() == sorted({k >> 4 for k, c in items if c.n})
    assert batch_bits(batch, ids) == bm.slice()
    # byte-identical re-serialisation (containers arrive already Optimize()d)
    assert batch.to_roaring() == raw
    # the uploaded rows are ordinary batch rows: |row_i ∩ row_j| against the oracle
    n = len(ids)
    ia, ib = np.arange(n), (np.arange(n) + 1) % n
    got = gpu_ctx.intersection_count(batch, ia, batch, ib)
    by_row = {}
    for k, c in bm.items():
        by_row.setdefault(k >> 4, {})[k & 15] = c
    for i in range(n):
        a, b = by_row[int(ids[ia[i]])], by_row[int(ids[ib[i]])]
        exp = sum(O.intersection_count(a[s], b[s]) for s in a if s in b)
        assert int(got[i]) == exp
    # a set-op result re-encoded with optimize() serialises to the oracle's WriteTo of the same
    # result: second image with the same row ids (as two fields of one shard have)
    items2 = []
    for r in range(n_rows):
        row = D.random_row(rng, 0)
        items2.extend(((r * 3 + 1) * 16 + (k & 15), c) for k, c in row.items())
    for rid in ids:  # make sure every row id of image 1 exists in image 2
        items2 = [kv for kv in items2 if kv[0] != int(rid) * 16] + [(int(rid) * 16, O.OContainer.array([3, 4]))]
    bm2 = O.OBitmap.from_containers(items2)
    batch2, ids2 = gpu_ctx.upload_roaring(bm2.marshal(True))
    assert ids2.tolist() == ids.tolist()
    out, cnt = gpu_ctx.setop(L.OP_OR, batch, ia, batch2, ia, L.SETOP_OPTIMIZE)
    exp = bm.union(bm2)
    assert int(cnt.sum()) == exp.count()
    assert out.to_roaring() == exp.marshal(True)
    out2, cnt2 = gpu_ctx.setop(L.OP_ANDNOT, batch, ia, batch2, ia, L.SETOP_OPTIMIZE)
    assert out2.to_roaring() == bm.difference(bm2).marshal(True)
    out.free()
    out2.free()
    batch2.free()
    batch.free()


def test_unoptimized_and_large_images(gpu_ctx, oracle):
    """writeToUnoptimized images (arrays >= 4096 values, bitmaps with few bits, long run lists)
    and odd payload alignments survive the device unpack."""
    O = oracle
    rng = D.rng_for(95)
    big_array = np.sort(rng.choice(65536, size=5000, replace=False))
    items = [
        (0, O.OContainer.array([7])),  # 2-byte payload: everything after it is only 2-byte aligned
        (1, O.OContainer.run([(i * 20, i * 20 + 3) for i in range(3000)])),
        (2, O.OContainer.bitmap(np.array([1, 0, 1 << 63] + [0] * 1021, dtype=np.uint64))),
        (5, O.OContainer.array(big_array)),
        (16, O.OContainer.array([1, 2, 3])),
        (17, O.OContainer.run([(0, 65535)])),
        ((1 << 44) + 3, O.OContainer.array([65535])),
    ]
    bm = O.OBitmap.from_containers(items)
    raw = bm.marshal(False)
    batch, ids = gpu_ctx.upload_roaring(raw)
    assert ids.tolist() == [0, 1, 1 << 40]
    assert batch_bits(batch, ids) == bm.slice()
    assert batch.to_roaring() == raw
    rows = batch.download()
    assert rows[0][5].typ == L.TYPE_ARRAY and rows[0][5].n == 5000 and rows[1][17].n == 65536
    batch.free()


# ---- ops log behind the containers (Bitmap.UnmarshalBinary replays it, unmarshal_binary.go:66-95) ----
def _replay_case(gpu_ctx, oracle, base_raw, ops):
    """image + encoded ops through fbk_batch_upload_roaring vs the set model of oracle/pywire_ops.py."""
    from oracle import pywire_ops as W

    to_set = lambda img: set(oracle.OBitmap.unmarshal(img).slice())
    model = W.apply_ops(to_set(base_raw), ops, to_set)
    raw = base_raw + b"".join(W.op_encode(t, p) for t, p in ops)
    assert W.ops_parse(raw[len(raw) - sum(len(W.op_encode(t, p)) for t, p in ops):]) == list(ops)
    batch, ids = gpu_ctx.upload_roaring(raw)
    touched = {p >> 20 for p in to_set(base_raw)}
    for t, p in ops:
        touched |= {v >> 20 for v in ([p] if t < 2 else p if t < 4 else to_set(p))}
    assert ids.tolist() == sorted(touched)
    assert batch_bits(batch, ids) == sorted(model)
    assert int(batch.count(np.arange(len(ids))).sum()) == len(model)
    # every container comes out Optimize()d: the re-serialised bytes are WriteTo of the replayed bitmap
    assert batch.to_roaring() == oracle.bitmap_from_values(sorted(model)).marshal(True)
    batch.free()
    return model


def test_ops_log_reference_ops(gpu_ctx, oracle):
    """The twelve ops of TestOpLogWriteUnmarshal (roaring_internal_test.go:4007), behind an empty bitmap and behind a
    bitmap that already holds some of their positions."""
    ops = [(o["type"], o["value"] if "value" in o else o["values"])
           for o in json.load(open(os.path.join(HERE, "golden", "literal_vectors.json")))["op_log_ops"]]
    empty = oracle.OBitmap().marshal(True)
    assert _replay_case(gpu_ctx, oracle, empty, ops) == {27}
    for i in range(len(ops)):  # "test each one separately"
        _replay_case(gpu_ctx, oracle, empty, ops[i : i + 1])
    base = oracle.bitmap_from_values([0, 1, 2, 28, 44, 100, 51234567890, (3 << 20) + 5]).marshal(True)
    _replay_case(gpu_ctx, oracle, base, ops)


def test_ops_log_replay_mixed(gpu_ctx, oracle):
    """Point, batch and roaring ops in one log over a fragment-shaped image: order of operations on the same position,
    rows that only the log names, containers emptied by removals, nested images in Pilosa and official format."""
    from oracle import pywire_ops as W

    O = oracle
    rng = D.rng_for(97)
    items = []
    for r in (0, 2, 5):
        row = D.random_row(rng, 0)
        items.extend(((r * 16 + (k & 15)), c) for k, c in row.items())
    items = [kv for kv in items if kv[0] != 2 * 16 + 3] + [(2 * 16 + 3, O.OContainer.array([10, 11, 12]))]
    base = O.OBitmap.from_containers(items)
    raw = base.marshal(True)
    have = base.slice()
    pick = lambda n: [int(v) for v in rng.choice(have, size=n, replace=False)]
    nested_add = O.OBitmap.from_containers([
        (2 * 16 + 3, O.OContainer.run([(0, 9), (13, 5000)])),  # joins the small array into one long run
        (7 * 16 + 1, O.OContainer.array([1, 2, 3])),  # a row nothing else names
        (5 * 16 + 0, O.OContainer.bitmap(rng.integers(0, 1 << 63, size=1024, dtype=np.uint64))),
    ])
    nested_rm = O.OBitmap.from_containers([
        (2 * 16 + 3, O.OContainer.run([(0, 65535)])),  # empties the container again
        (0 * 16 + 1, O.OContainer.run([(100, 40000)])),
        (9 * 16 + 0, O.OContainer.array([5])),  # removal from a row that does not exist
    ])
    official = bytes.fromhex(FIX["ok"][0]["hex"])
    p0 = (11 << 20) + 77
    ops = [
        (W.ADD, p0), (W.REMOVE, p0), (W.ADD, p0),  # last one wins
        (W.REMOVE_N, pick(500)),
        (W.ADD_N, [int(v) for v in rng.integers(0, 6 << 20, size=3000)]),
        (W.ADD, (2 << 20) + (3 << 16) + 12), (W.REMOVE, (2 << 20) + (3 << 16) + 10),
        (W.ADD_ROARING, nested_add.marshal(True)),
        (W.REMOVE, (7 << 20) + (1 << 16) + 2),  # after the image that created it
        (W.ADD_N, [(7 << 20) + (1 << 16) + 2, (7 << 20) + (1 << 16) + 2]),  # duplicates in one batch
        (W.REMOVE_ROARING, nested_rm.marshal(False)),
        (W.ADD, (2 << 20) + (3 << 16) + 40000),  # into the container the image just emptied
        (W.ADD_ROARING, official),
        (W.REMOVE_N, pick(200) + [p0 + 1]),
        (W.ADD_N, []), (W.REMOVE_N, []),
    ]
    model = _replay_case(gpu_ctx, oracle, raw, ops)
    assert p0 in model and (7 << 20) + (1 << 16) + 2 in model and (2 << 20) + (3 << 16) + 40000 in model
    # prefixes of the log: every intermediate state
    for cut in (1, 3, 4, 5, 8, 9, 11, 12):
        _replay_case(gpu_ctx, oracle, raw, ops[:cut])


def test_ops_log_errors(gpu_ctx, oracle):
    from oracle import pywire_ops as W

    raw = oracle.bitmap_from_values([1, 2, 3, 1 << 21]).marshal(True)
    good = W.op_encode(W.ADD_N, [9, 10])
    for bad in (good[:-1],  # truncated batch
                good[:9] + bytes([good[9] ^ 1]) + good[10:],  # checksum
                b"\x07" + good[1:],  # unknown type
                good[:12],  # shorter than an op header
                W.op_encode(W.ADD, 5) + good[:5],  # a valid op, then garbage
                W.op_encode(W.ADD_ROARING, b"\x3c\x30\x01\x00\x00\x00\x00\x00")):  # nested image announces a container it lacks
        with pytest.raises(L.FbkError):
            gpu_ctx.upload_roaring(raw + bad)
    # the same ops behind a row id buffer that is too small
    import ctypes as C
    h, n = C.c_void_p(), C.c_uint32()
    ids = np.zeros(1, dtype=np.uint64)
    data = raw + W.op_encode(W.ADD, 40 << 20)
    assert gpu_ctx.lib.fbk_batch_upload_roaring(gpu_ctx.h, data, len(data), C.byref(h), ids.ctypes.data, 1, C.byref(n)) == L.FBK_E_CAPACITY
    assert n.value == 3 and not h.value
