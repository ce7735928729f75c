"""The BSI kernels on config 5's shape (96 shards x (64 planes + exists + sign), dense): kernel time by the
library's own HIP events (option time_kernels), both forms of Sum / Range / Min-Max side by side.

    python scripts/bsi_bench.py [shards=96]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

n5, depth = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 64
ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
g = torch.Generator(device="cuda:0").manual_seed(51)
w = torch.randint(-(1 << 63), (1 << 63) - 1, (n5, depth + 2, 16, 1024), dtype=torch.int64, device="cuda:0", generator=g).cpu().numpy().view(np.uint64)
w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
w[-1, 0, 6:] = 0
batch = ctx.upload_dense(w.reshape(-1))
base = np.arange(n5, dtype=np.uint32) * (depth + 2)
plane_bytes = n5 * 16 * 8192
filt, _ = ctx.bsi_range(batch, base, L.BSI_GT, depth, 1 << 62)
fidx = np.arange(n5)


def kernel_us(fn, iters=15):
    for _ in range(3):
        fn()
    ctx.set_option("time_kernels", 1)
    ts = []
    for _ in range(iters):
        fn()
        ts.append(ctx.get_option("last_kernel_ns") / 1e3)
    ctx.set_option("time_kernels", 0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def line(what, nbytes, fn):
    med, lo = kernel_us(fn)
    print(f"{what:58s} {med:8.1f} us (min {lo:6.1f})  {nbytes / med / 1e6:6.2f} TB/s = {nbytes / med / 1e6 / 8:.2f}")


line("Sum(filter)  k_bsi_sum_slot (wavefront per (shard, slot))", plane_bytes * (depth + 3), lambda: ctx.bsi_sum(batch, base, depth, filt, fidx))
line("Sum()        k_bsi_sum_slot (wavefront per (shard, slot))", plane_bytes * (depth + 2), lambda: ctx.bsi_sum(batch, base, depth))
for op, pred, what in ((L.BSI_GT, 1 << 62, "Range(> 2^62)"), (L.BSI_LT, -5, "Range(< -5)"), (L.BSI_EQ, 12345, "Range(== 12345)")):
    line(f"{what:16s} k_bsi_range_slot (wavefront)", plane_bytes * (depth + 3), lambda: ctx.bsi_range(batch, base, op, depth, pred)[0].free())
def range_then_sum(op, pred):
    rows, _ = ctx.bsi_range(batch, base, op, depth, pred)
    r = ctx.bsi_sum(batch, base, depth, rows, fidx)
    rows.free()
    return r


ref_rs = range_then_sum(L.BSI_GT, 1 << 62)
r = ctx.bsi_range_sum(batch, base, L.BSI_GT, depth, 1 << 62)
assert (ref_rs[0] == r[0]).all() and (ref_rs[1] == r[1]).all(), "Range+Sum: one pass and two calls disagree"
line("Sum(Range(> 2^62)) one pass  k_bsi_range_sum_slot / _half", plane_bytes * (depth + 2), lambda: ctx.bsi_range_sum(batch, base, L.BSI_GT, depth, 1 << 62))
line("Sum(Range(< -5)) one pass    k_bsi_range_sum_slot / _half", plane_bytes * (depth + 2), lambda: ctx.bsi_range_sum(batch, base, L.BSI_LT, depth, -5))
line("Sum(Range(> -5)) one pass    ..<other class>", plane_bytes * (depth + 2), lambda: ctx.bsi_range_sum(batch, base, L.BSI_GT, depth, -5))
print("  (two passes: the Range and the Sum(filter) lines above, one after the other)")
for (lo, hi), what in (((-(1 << 61), 1 << 62), "both signs"), ((1 << 60, 1 << 62), "one sign, split lanes")):
    rows, _ = ctx.bsi_range_between(batch, base, depth, lo, hi)
    ref_x = ctx.bsi_sum(batch, base, depth, rows, fidx)
    rows.free()
    r = ctx.bsi_range_between_sum(batch, base, depth, lo, hi)
    assert (ref_x[0] == r[0]).all() and (ref_x[1] == r[1]).all(), "Between+Sum: one pass and two calls disagree"
    line(f"Sum(Between) one pass, {what:22s} k_bsi_between_sum_part", plane_bytes * (depth + 2), lambda: ctx.bsi_range_between_sum(batch, base, depth, lo, hi))
line("Between (row output)                     k_bsi_range_slot", plane_bytes * (depth + 2), lambda: ctx.bsi_range_between(batch, base, depth, 1 << 60, 1 << 62)[0].free())
line("Min          k_bsi_minmax_slot (wavefront per (shard, slot))", plane_bytes * (depth + 2), lambda: ctx.bsi_min(batch, base, depth))
line("Max(filter)  k_bsi_minmax_slot (wavefront per (shard, slot))", plane_bytes * (depth + 3), lambda: ctx.bsi_max(batch, base, depth, filt, fidx))

