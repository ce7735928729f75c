"""fbk_setop(..., FBK_SETOP_OPTIMIZE) on config 3's rows (rows 0..31 against rows 32..63 of every shard, 16 containers
each): the set-op kernel writes right-sized cells (arrays for results of <= 1024 values), the re-encode pass compacts
them.  Prints the encoded size of each result; run under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE / --kernel-trace --stats
(scripts/pmc_write_misc.sh) for the HBM bytes and times of k_setop<OP>, k_encode_*.
    python scripts/profile_setop_optimize.py [shards=64] [op = and | or | xor | andnot | all]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

N3 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "all"
rows, groups, filt = D.config3_flat(N3, mp="fork")
from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
out = {"shards": N3, "pairs": int(pa.size), "operand_bytes": int(rows.bytes)}
for name, op in (("and", L.OP_AND), ("or", L.OP_OR), ("xor", L.OP_XOR), ("andnot", L.OP_ANDNOT)):
    if which not in ("all", name):
        continue
    for _ in range(5):
        o, cnt = ctx.setop(op, batch, pa, batch, pb, L.SETOP_OPTIMIZE)
        size = o.info()[2]  # payload bytes of the encoded result
        o.free()
    out[name] = {"result_bits": int(cnt.sum()), "encoded_result_payload_bytes": int(size)}
print(json.dumps(out))
