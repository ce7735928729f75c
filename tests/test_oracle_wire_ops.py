"""The ops-log restatement (oracle/pywire_ops.py) against the reference's own test material: the twelve ops of
TestOpLogWriteUnmarshal (roaring_internal_test.go:4007-4090, extracted mechanically into golden/literal_vectors.json)
written, parsed back and applied; the byte layout op.WriteTo documents; the errors op.UnmarshalBinary returns."""
import json
import os
import struct

import pytest

from oracle import pywire_ops as W

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = json.load(open(os.path.join(HERE, "golden", "literal_vectors.json")))["op_log_ops"]


def as_op(o):
    return (o["type"], o["value"] if "value" in o else o["values"])


def test_fnv1a32_known_answers():
    # FNV-1a 32 test vectors (hash/fnv's own: "", "a", "ab", "abc")
    assert W.fnv1a32(b"") == 0x811C9DC5
    assert W.fnv1a32(b"a") == 0xE40C292C
    assert W.fnv1a32(b"ab") == 0x4D2505CA
    assert W.fnv1a32(b"abc") == 0x1A47E90B
    assert W.fnv1a32(b"a", b"bc") == W.fnv1a32(b"abc")


def test_reference_ops_write_then_unmarshal():
    """"test them all in sequence" and "test each one separately" of TestOpLogWriteUnmarshal."""
    assert len(OPS) == 12
    ops = [as_op(o) for o in OPS]
    log = b"".join(W.op_encode(t, p) for t, p in ops)
    assert W.ops_parse(log) == ops
    for t, p in ops:
        assert W.ops_parse(W.op_encode(t, p)) == [(t, p)]
    # sizes: op.encodeSize (roaring.go:6447): 13, 13 + 8 n
    assert [len(W.op_encode(t, p)) for t, p in ops] == [13 if t < 2 else 13 + 8 * len(p) for t, p in ops]
    # applied in order to an empty bitmap
    assert W.apply_ops(set(), ops, None) == {27}


def test_layout_and_errors():
    e = W.op_encode(W.ADD, 0x0102030405060708)
    assert e[0] == 0 and e[1:9] == bytes([8, 7, 6, 5, 4, 3, 2, 1]) and len(e) == 13
    assert struct.unpack_from("<I", e, 9)[0] == W.fnv1a32(e[:9])
    r = W.op_encode(W.ADD_ROARING, b"\x3c\x30\0\0\0\0\0\0", op_n=5)
    assert len(r) == 13 + 4 + 8 and struct.unpack_from("<Q", r, 1)[0] == 8 and struct.unpack_from("<I", r, 13)[0] == 5
    assert W.ops_parse(r) == [(W.ADD_ROARING, b"\x3c\x30\0\0\0\0\0\0")]
    with pytest.raises(ValueError, match="out of bounds"):
        W.ops_parse(e[:12])
    with pytest.raises(ValueError, match="checksum"):
        W.ops_parse(e[:3] + b"\xff" + e[4:])
    with pytest.raises(ValueError, match="unknown op type"):
        W.ops_parse(b"\x09" + e[1:])
    b = W.op_encode(W.ADD_N, [1, 2, 3])
    with pytest.raises(ValueError, match="truncated"):
        W.ops_parse(b[:-1])
    with pytest.raises(ValueError, match="maximum operation size"):
        W.ops_parse(struct.pack("<BQI", 2, (1 << 59) + 1, 0))
    with pytest.raises(ValueError, match="truncated"):
        W.ops_parse(r[:-1])


def test_apply_order_matters():
    ops = [(W.ADD_N, [5, 6, 7]), (W.REMOVE, 6), (W.ADD, 6), (W.REMOVE_N, [5, 6]), (W.ADD, 5)]
    assert W.apply_ops({1}, ops, None) == {1, 5, 7}
    assert W.apply_ops({1, 2, 3}, [(W.REMOVE_ROARING, b"x"), (W.ADD_ROARING, b"y")], {b"x": {2, 9}, b"y": {4}}.__getitem__) == {1, 3, 4}
