"""Pin the oracle's roaring wire codec (oracle/wire_oracle.c) to the reference's fixtures:
TestUnmarshalRoaringWithNoErrors / WithErrors (roaring_internal_test.go:3793-3880, recorded in
tests/golden/wire_fixtures.json by extract_wire_fixtures.py) and check the Pilosa writer
against the documented layout (roaring.go:1730-1817) and by round trips.  CPU only."""
import json
import os
import struct

import numpy as np
import pytest

import datagen as D

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "wire_fixtures.json")))


@pytest.mark.parametrize("fx", FIX["ok"], ids=lambda f: f"{len(f['hex']) // 2}B")
def test_unmarshal_official_fixtures(oracle, fx):
    bm = oracle.OBitmap.unmarshal(bytes.fromhex(fx["hex"]))
    assert bm.count() == fx["count"]
    if fx["bits"] is not None:
        assert bm.slice() == fx["bits"]
    else:  # the 8218-byte file: a bitmap container (N = 9999 >= 4096) at key 0 + a 1-element array
        items = bm.items()
        assert [(k, c.typ, c.n) for k, c in items] == [(0, 2, 9999), (1, 1, 1)]
        assert int(np.bitwise_count(items[0][1].words()).sum()) == 9999  # header N matches the payload


def test_unmarshal_errors_and_empty(oracle):
    for e in FIX["errors"]:
        with pytest.raises(ValueError):
            oracle.OBitmap.unmarshal(bytes.fromhex(e["hex"]))
    for h in FIX["pilosa_empty_ok"]:
        assert len(oracle.OBitmap.unmarshal(bytes.fromhex(h))) == 0
    with pytest.raises(ValueError):
        oracle.OBitmap.unmarshal(b"\x3c\x30")  # shorter than headerBaseSize
    with pytest.raises(ValueError):
        oracle.OBitmap.unmarshal(bytes.fromhex("3C30010000000000"))  # storage version 1
    with pytest.raises(ValueError):
        oracle.OBitmap.unmarshal(bytes.fromhex("FFFF000000000000"))  # unknown magic


def test_pilosa_writer_layout_and_round_trip(oracle):
    O = oracle
    rng = D.rng_for(91)
    bm = O.OBitmap.from_containers(
        [
            (3, O.OContainer.array([1, 5, 9])),
            (7, O.OContainer.run([(10, 20), (40, 65535)])),
            (1 << 40, O.OContainer.bitmap(rng.integers(0, 1 << 63, size=1024, dtype=np.uint64))),
            (9, O.OContainer.array([])),  # empty containers are not written (roaring.go:1768)
        ]
    )
    raw = bm.marshal(optimize_first=False)
    cookie, n = struct.unpack_from("<II", raw, 0)
    assert cookie == 12348 and n == 3  # MagicNumber, version 0, flags 0 (roaring.go:20-30)
    hdr = [struct.unpack_from("<QHH", raw, 8 + 12 * i) for i in range(n)]
    offs = [struct.unpack_from("<I", raw, 8 + 12 * n + 4 * i)[0] for i in range(n)]
    assert [h[0] for h in hdr] == [3, 7, 1 << 40]
    assert [h[1] for h in hdr] == [1, 3, 2]  # array, run, bitmap (roaring.go:53-58)
    assert hdr[0][2] == 2 and hdr[1][2] == (11 + 65496) - 1  # N-1
    assert offs[0] == 8 + 16 * n and offs[1] == offs[0] + 6 and offs[2] == offs[1] + 2 + 8  # sizes 2N / 2+4r / 8192
    assert struct.unpack_from("<H", raw, offs[1])[0] == 2  # run count prefix (roaring.go:4100)
    assert len(raw) == offs[2] + 8192
    back = O.OBitmap.unmarshal(raw)
    assert back.slice() == bm.slice()
    assert [(k, c.typ, c.n) for k, c in back.items()] == [(k, c.typ, c.n) for k, c in bm.items() if c.n]
    # WriteTo optimizes first (roaring.go:1731): the run [40, 65535] + [10, 20] stays a run,
    # the random bitmap stays a bitmap, the array stays an array
    assert bm.marshal(True) == raw
    # random bitmaps: marshal -> unmarshal is the identity on bit content and Optimize()d encodings
    for t in range(20):
        rows = D.random_row(rng, t)
        b = O.OBitmap.from_containers(list(rows.items()))
        r1 = b.marshal(True)
        u = O.OBitmap.unmarshal(r1)
        assert u.slice() == b.slice()
        assert u.marshal(False) == r1  # already optimal: byte-identical
        for (k, c) in u.items():
            oc = O.optimize(c)
            assert oc.typ == c.typ


def test_official_with_runs_converts_start_length(oracle):
    """3B30 fixture: one run stored as {start=1, length-1=9} must become [1, 10] (:2243-2246)."""
    fx = FIX["ok"][1]
    bm = oracle.OBitmap.unmarshal(bytes.fromhex(fx["hex"]))
    items = bm.items()
    assert items[0][1].typ == 3 and items[0][1].data().reshape(-1).tolist() == [1, 10]
    assert items[1][1].typ == 1 and items[1][0] == 1
