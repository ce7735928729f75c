"""fbk_bsi_add (k_bsi_add) of the library FBK_LIB_PATH names: two 16-plane operands over 96 shards, call time between events on the
context's stream, median of 30 (library builds compared by running this once per build, alternating: scripts/r6_call.sh addab)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

n5, depth = 96, 64
ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
g = torch.Generator(device="cuda")
g.manual_seed(5)
w = torch.randint(-(2**63), 2**63 - 1, (n5 * (depth + 2), 16, 1024), dtype=torch.int64, device="cuda", generator=g).cpu().numpy().view(np.uint64)
bsi = ctx.upload_dense(w.reshape(-1))
base = np.arange(n5, dtype=np.uint32) * (depth + 2)
px = (base[:, None] + 2 + np.arange(16)[None, :]).astype(np.uint32)
py = (base[:, None] + 18 + np.arange(16)[None, :]).astype(np.uint32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
chk = None
for i in range(40):
    with torch.cuda.stream(st):
        e0.record(st)
        r = ctx.bsi_add(bsi, px, bsi, py)
        e1.record(st)
    torch.cuda.synchronize()
    if i >= 10:
        ts.append(e0.elapsed_time(e1) * 1e3)
    out = r[0] if isinstance(r, tuple) else r
    if chk is None and hasattr(out, "info"):
        chk = [int(x) for x in out.info()]
    out.free()
ts.sort()
pl = n5 * 16 * 8192
print(json.dumps({"lib": os.environ.get("FBK_LIB_PATH", "product"), "call_us_median": round(ts[len(ts) // 2], 1), "min": round(ts[0], 1), "bytes": pl * (32 + 17),
                  "frac": round(pl * 49 / (ts[len(ts) // 2] * 1e-6) / 8e12, 4), "out_info": chk}))
