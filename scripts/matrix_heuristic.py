"""Measures where fbk_count_matrix should densify encoded rows (k_densify_rows + the matrix-core
kernel) instead of running the generic pair kernel (k_count_matrix<4>): rows of one uniform
density per run, nA x nB matrix per shard, both paths forced with FBK_MATRIX_DENSIFY=0/1 and
checked against each other.  One line per (density, shape).

    python scripts/matrix_heuristic.py [--shards 64]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen as D  # noqa: E402
from bench_configs import encoded_bytes, timed  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ctx = Context(0)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    n_rows = 64
    with torch.cuda.stream(stream):
        for d in (0.001, 0.004, 0.01, 0.02, 0.04, 0.2):
            rows, nbytes, ncont = [], 0, 0
            for s in range(a.shards):
                rng = D.rng_for(9000 + s, int(d * 1e4))
                for r in range(n_rows):
                    row = {}
                    for slot in range(16):
                        rs = rng.random() < 0.25
                        c = D.fbk_container_of_vals(D.mixed_vals_for_density(rng, d, rs))
                        if c is not None and c.n:
                            row[s * 16 + slot] = c
                            nbytes += encoded_bytes(c)
                            ncont += 1
                    rows.append(row)
            batch = ctx.upload(rows)
            ids = np.arange(a.shards * n_rows, dtype=np.uint32).reshape(a.shards, n_rows)
            for n_a, n_b in ((4, 4), (8, 8), (16, 16), (32, 32), (8, 32), (2, 62)):
                ra, rb = ids[:, :n_a], ids[:, n_a:n_a + n_b]
                res = {}
                for mode in ("0", "1"):
                    os.environ["FBK_MATRIX_DENSIFY"] = mode
                    tot = ctx.count_matrix(batch, ra, batch, rb)
                    res[mode] = (timed(stream, lambda: ctx.count_matrix(batch, ra, batch, rb), a.iters), tot)
                os.environ.pop("FBK_MATRIX_DENSIFY")
                assert (res["0"][1] == res["1"][1]).all()
                avg = nbytes / (a.shards * n_rows * 16)
                print(json.dumps({"density": d, "avg_bytes_per_slot": round(avg), "n_a": n_a, "n_b": n_b,
                                  "metric": round(avg * n_a * n_b / (n_a + n_b)),
                                  "generic_us": round(res["0"][0] * 1e6, 1), "densify_us": round(res["1"][0] * 1e6, 1)}), flush=True)
            batch.free()


if __name__ == "__main__":
    main()
