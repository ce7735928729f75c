"""Executor-level known answers of the reference (executor_test.go:1236-1373, extracted into
tests/golden/executor_vectors.json): set-ops over rows whose columns straddle ShardWidth, i.e.
the per-shard map (executeIntersectShard executor.go:5357 etc.) + concatenation by shard.
Checked on the oracle (CPU) and on the HIP path (GPU)."""
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "executor_vectors.json")))
SW = VEC["shard_width"]


def parse(q):
    m = re.match(r"(\w+)\((.*)\)$", q)
    op, args = m.group(1), m.group(2)
    rows = [int(r) for r in re.findall(r"Row\(\w+=(\d+)\)", args)]
    return op, rows


def per_shard_columns(case):
    """{shard: {rowid: [columns within the shard]}}"""
    out = {}
    for _f, r, c in case["bits"]:
        out.setdefault(c // SW, {}).setdefault(r, []).append(c % SW)
    return out


@pytest.mark.parametrize("case", VEC["cases"], ids=lambda c: c["test"])
def test_executor_vectors_oracle(oracle, case):
    O = oracle
    op, rows = parse(case["query"])
    cols, total = [], 0
    for shard, byrow in sorted(per_shard_columns(case).items()):
        bms = [O.bitmap_from_values(byrow.get(r, [])) for r in rows]
        if op == "Count":
            total += bms[0].count()
            continue
        res = {"Intersect": lambda: bms[0].intersect(bms[1]), "Union": lambda: bms[0].union(bms[1]), "Difference": lambda: bms[0].difference(bms[1]), "Xor": lambda: bms[0].xor(bms[1])}[op]()
        cols += [shard * SW + v for v in res.slice()]
    if op == "Count":
        assert total == case["count"]
    else:
        assert cols == case["columns"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", VEC["cases"], ids=lambda c: c["test"])
def test_executor_vectors_gpu(gpu_ctx, oracle, case):
    import datagen as D
    from featurebase_amd import lib as L

    O = oracle
    op, rows = parse(case["query"])
    shards = sorted(per_shard_columns(case).items())
    batch_rows = []
    for shard, byrow in shards:
        for r in rows:
            bm = O.bitmap_from_values(byrow.get(r, []))
            batch_rows.append({shard * 16 + k: D.to_fbk(c) for k, c in bm.items() if c.n})
    batch = gpu_ctx.upload(batch_rows)
    n = len(shards)
    if op == "Count":
        assert int(batch.count(np.arange(n)).sum()) == case["count"]
        batch.free()
        return
    ra = np.arange(n) * 2
    code = {"Intersect": L.OP_AND, "Union": L.OP_OR, "Difference": L.OP_ANDNOT, "Xor": L.OP_XOR}[op]
    out, cnt = gpu_ctx.setop(code, batch, ra, batch, ra + 1, flags=L.SETOP_OPTIMIZE)
    cols = []
    for row in out.download():
        for key, c in sorted(row.items()):
            bits = np.unpackbits(c.words().view(np.uint8), bitorder="little")
            cols += [(key << 16) + int(v) for v in np.nonzero(bits)[0]]  # key = shard*16 + slot
    assert cols == case["columns"]
    assert int(cnt.sum()) == len(case["columns"])
    out.free()
    batch.free()
