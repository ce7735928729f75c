"""TEST INFRASTRUCTURE ONLY (oracle): the ops log that can follow the containers of a Pilosa-format roaring file —
op.WriteTo / op.UnmarshalBinary / op.apply (roaring/roaring.go:6254-6431) and the replay loop of
Bitmap.UnmarshalBinary (roaring/unmarshal_binary.go:66-95) — restated on Python sets.

An op is (type, payload): 0 add / 1 remove (one position), 2 addN / 3 removeN (list of positions), 4 addRoaring /
5 removeRoaring (a serialised bitmap, bytes).  Encoding: type u8, value u64, checksum u32 = FNV-1a 32 over bytes [0, 9)
and everything after the checksum; batches append their positions (u64 each); roaring ops append opN (u32) and the
image.  The reference's tests (TestOpLogWriteUnmarshal, roaring_internal_test.go:4007-4090) write ops and read them
back with the library under test — there are no byte fixtures to extract; the 12 ops of that test are extracted
mechanically (tests/golden/literal_vectors.json) and go through this writer / parser pair and through fbk's parser.
"""
import struct
from typing import Iterable, List, Sequence, Set, Tuple, Union

ADD, REMOVE, ADD_N, REMOVE_N, ADD_ROARING, REMOVE_ROARING = range(6)


def fnv1a32(*chunks: bytes) -> int:
    h = 2166136261
    for c in chunks:
        for b in c:
            h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h


def op_encode(typ: int, payload: Union[int, Sequence[int], bytes], op_n: int = 0) -> bytes:
    """op.WriteTo (roaring.go:6325-6361)."""
    if typ in (ADD, REMOVE):
        head, tail, extra = struct.pack("<BQ", typ, int(payload)), b"", b""
    elif typ in (ADD_N, REMOVE_N):
        vals = list(payload)
        head, tail, extra = struct.pack("<BQ", typ, len(vals)), b"".join(struct.pack("<Q", v) for v in vals), b""
    elif typ in (ADD_ROARING, REMOVE_ROARING):
        head, tail, extra = struct.pack("<BQ", typ, len(payload)), struct.pack("<I", op_n), bytes(payload)
    else:
        raise ValueError(f"can't marshal unknown op type {typ}")
    return head + struct.pack("<I", fnv1a32(head, tail, extra)) + tail + extra


def ops_parse(data: bytes) -> List[Tuple[int, object]]:
    """The loop of Bitmap.UnmarshalBinary over op.UnmarshalBinary (roaring.go:6364-6431); ValueError where the
    reference returns an error."""
    out, off = [], 0
    while off < len(data):
        d = data[off:]
        if len(d) < 13:
            raise ValueError(f"op data out of bounds: len={len(d)}")
        typ, value, chk = d[0], struct.unpack_from("<Q", d, 1)[0], struct.unpack_from("<I", d, 9)[0]
        if typ in (ADD, REMOVE):
            size, body, payload = 13, b"", value
        elif typ in (ADD_N, REMOVE_N):
            if value > (1 << 59):
                raise ValueError("maximum operation size exceeded")
            size = 13 + value * 8
            if len(d) < size:
                raise ValueError("op data truncated")
            body = d[13:size]
            payload = list(struct.unpack_from(f"<{value}Q", d, 13))
        elif typ in (ADD_ROARING, REMOVE_ROARING):
            size = 17 + value
            if len(d) < size:
                raise ValueError("op data truncated")
            body = d[13:size]
            payload = bytes(d[17:size])
        else:
            raise ValueError(f"unknown op type: {typ}")
        if chk != fnv1a32(d[0:9], body):
            raise ValueError(f"checksum mismatch: type {typ}")
        out.append((typ, payload))
        off += size
    return out


def apply_ops(bits: Set[int], ops: Iterable[Tuple[int, object]], image_to_set) -> Set[int]:
    """op.apply one by one (roaring.go:6296-6322).  image_to_set: bytes -> set of positions (the containers of a
    nested image; ImportRoaringBits does not replay an ops log inside it)."""
    for typ, payload in ops:
        if typ == ADD:
            bits.add(payload)
        elif typ == REMOVE:
            bits.discard(payload)
        elif typ == ADD_N:
            bits.update(payload)
        elif typ == REMOVE_N:
            bits.difference_update(payload)
        elif typ == ADD_ROARING:
            bits |= image_to_set(payload)
        elif typ == REMOVE_ROARING:
            bits -= image_to_set(payload)
        else:
            raise ValueError(f"invalid op type: {typ}")
    return bits
