#!/usr/bin/env python3
"""The reference's container-operation benchmark matrix on the GPU.

BenchmarkCtOps (roaring/roaring_container_test.go:33-59) runs intersect / union / difference / xor /
intersectionCount over every ordered pair of the 20 container archetypes of
roaring/container_archetypes.go:19-40 (Empty, Ary1 .. Ary4096, RunFull, RunSplit, Run16 .. Run1024,
BM512 .. BM65000; 8 instances each, the 8 x 8 instance matrix per cell).  Here one cell = one launch
over ROWS x 16 container pairs of the two archetypes (64 instances each, every container stored
separately in HBM), timed with HIP events over back-to-back launches of a plan; reported per cell:
container operations per second and the encoded bytes the operands occupy.  Every cell's result is
checked bit-exactly against the oracle on its first row (intersectionCount: all rows).

    python scripts/bench_ctops.py [--rows 1024] [--iters 20] [--out profiles/ctops_rNN.json] [--opt name=value ...]

The archetype generators follow the reference's (sizes, run geometry); its RNG streams (math/rand,
apophenia) are not reproducible without Go, seeds here are numpy's.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

NAMES = ["Empty", "Ary1", "Ary16", "Ary256", "Ary512", "Ary1024", "Ary4096", "RunFull", "RunSplit", "Run16", "Run16Small", "Run256",
         "Run256Small", "Run1024", "BM512", "BM1024", "BM4096", "BM4097", "BM32768", "BM65000"]
OPS = ["intersectionCount", "intersect", "union", "difference", "xor"]


def archetype(rng, name):
    """(type, data, n) of one container: container_archetypes.go:47-160."""
    if name == "Empty":
        return None
    if name.startswith("Ary"):
        size = int(name[3:])
        v = np.sort(rng.choice(65536, size=size, replace=False)).astype(np.uint16)
        return 1, v, size
    if name == "RunFull":
        return 3, np.array([[0, 65535]], dtype=np.uint16), 65536
    if name == "RunSplit":
        a, b = 32700 + int(rng.integers(30)), 32768 + int(rng.integers(30))
        return 3, np.array([[0, a], [b, 65535]], dtype=np.uint16), a + 1 + 65536 - b
    if name.startswith("Run"):
        small = name.endswith("Small")
        count = int(name[3:-5] if small else name[3:])
        stride = 65535 // (count + 1)
        upper, lower = stride - 10, stride // 10
        if small:
            lower = 3
            upper = lower + stride // 20
        variance = upper - lower
        runs = np.zeros((count, 2), dtype=np.int64)
        nxt = prev = 0
        for i in range(count):
            nxt += stride
            middle = (prev + nxt) // 2
            size = int(rng.integers(variance)) + lower
            off = int(rng.integers(variance))
            s = (middle + off - size // 2) & 0xFFFF
            runs[i] = (s, (s + size) & 0xFFFF)
            prev = nxt
            if i > 0 and runs[i, 0] <= runs[i - 1, 1]:
                runs[i, 0] = runs[i - 1, 1] + 2
                if runs[i, 1] < runs[i, 0]:
                    runs[i, 1] = runs[i, 0]
        return 3, runs.astype(np.uint16), int((runs[:, 1] - runs[:, 0] + 1).sum())
    size = int(name[2:])  # BM<n>: a bitmap container with exactly n bits
    v = rng.choice(65536, size=size, replace=False)
    return 2, D.words_of(np.sort(v)), size


def build_batch(ctx, name, rows, seed, instances=64):
    """rows x 16 containers of one archetype, cycling over `instances` distinct ones (each stored separately)."""
    rng = D.rng_for(9000 + seed)
    inst = [archetype(rng, name) for _ in range(instances)]
    fr = D.FlatRows()
    for r in range(rows):
        for s in range(16):
            c = inst[(r * 16 + s) % instances]
            if c is not None:
                fr.add(r, r * 16 + s, c[0], c[1], c[2])
    fr.n_rows = rows
    return ctx.upload_flat(fr.descs(), fr.payload(), rows), inst, fr.bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--opt", action="append", default=[], help="context option name=value (A/B runs)")
    ap.add_argument("--only", default="", help="comma list of archetype names (rows and columns)")
    args = ap.parse_args()
    import torch

    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Container, Context
    from oracle import pyoracle as O

    names = [n for n in NAMES if not args.only or n in args.only.split(",")]
    ctx = Context(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    rows = args.rows
    t0 = time.time()
    A = {n: build_batch(ctx, n, rows, 2 * i) for i, n in enumerate(names)}
    B = {n: build_batch(ctx, n, rows, 2 * i + 1) for i, n in enumerate(names)}
    print(f"# {len(names)} archetypes x 2 roles uploaded in {time.time() - t0:.1f} s, {rows} rows x 16 containers each", file=sys.stderr)

    def ocont(c):
        if c is None:
            return O.OContainer(None)
        t, d, n = c
        return O.OContainer.array(d) if t == 1 else O.OContainer.bitmap(d, n) if t == 2 else O.OContainer.run(d.tolist())

    idx = np.arange(rows)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    opcode = {"intersect": L.OP_AND, "union": L.OP_OR, "difference": L.OP_ANDNOT, "xor": L.OP_XOR}
    table = {}
    pairs = rows * 16
    for n1 in names:
        for n2 in names:
            ba, ia, bytes_a = A[n1]
            bb, ib, bytes_b = B[n2]
            plan = ctx.plan(ba, idx, bb, idx)
            cell = {"bytes_per_pair": (bytes_a + bytes_b) / pairs}
            # parity: intersectionCount of every row; the four set-ops on row 0
            plan.intersection_count()
            got = plan.read()
            per_inst = {}
            exp_row = np.zeros(rows, dtype=np.uint64)
            for r in range(min(rows, 8)):
                tot = 0
                for s in range(16):
                    k = (r * 16 + s) % 64
                    if k not in per_inst:
                        per_inst[k] = O.intersection_count(ocont(ia[k]), ocont(ib[k])) if ia[k] is not None and ib[k] is not None else 0
                    tot += per_inst[k]
                assert int(got[r]) == tot, (n1, n2, r, int(got[r]), tot)
            for op in OPS:
                if op == "intersectionCount":
                    fn = plan.intersection_count
                else:
                    fn = (lambda o=opcode[op]: plan.setop(o))
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0.record(st)
                for _ in range(args.iters):
                    fn()
                e1.record(st)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                cell[op] = {"us": us, "ops_per_s": pairs / (us * 1e-6)}
                if op != "intersectionCount":
                    out = plan.output().download()
                    for s in range(16):
                        a, b = ia[s % 64], ib[s % 64]
                        oa, ob = ocont(a), ocont(b)
                        e = O.OPS[op](oa, ob) if a is not None and b is not None else (O.OContainer(None) if op == "intersect" or (op == "difference" and a is None) else (oa if a is not None else (ob if op != "difference" else O.OContainer(None))))
                        g = out[0].get(s)
                        gw = g.words() if g is not None else np.zeros(1024, dtype=np.uint64)
                        assert (gw == e.words()).all(), (n1, n2, op, s)
            plan.free()
            table[f"{n1}/{n2}"] = cell
    # Intersect + optimize() (fbk_setop with FBK_SETOP_OPTIMIZE, one C-ABI call end to end) with and without
    # right-sized outputs: small-array intersections written as arrays by k_setop itself
    # (option setop_direct_encode) instead of as 8 KiB bitmap cells that k_encode_* then shrink
    opt_path = {}
    for n1, n2 in (("Ary16", "Ary16"), ("Ary16", "BM4096"), ("BM32768", "Ary16"), ("Ary256", "Ary256")):
        if n1 not in A or n2 not in B:
            continue
        for direct in (0, 1):
            ctx.set_option("setop_direct_encode", direct)
            fn = lambda: ctx.setop(L.OP_AND, A[n1][0], idx, B[n2][0], idx, L.SETOP_OPTIMIZE)[0].free()  # noqa: E731
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
            opt_path[f"{n1}/{n2}/direct={direct}"] = (time.perf_counter() - t0) / args.iters * 1e6
        ctx.set_option("setop_direct_encode", 1)
    print("\nintersect + optimize(), one fbk_setop call, wall us:", json.dumps(opt_path))
    res = {"rows": rows, "optimize_path_wall_us": opt_path, "pairs_per_launch": pairs, "iters": args.iters, "options": args.opt, "names": names, "cells": table}
    if args.out:
        json.dump(res, open(args.out, "w"))
    # compact text table: Mops/s of intersectionCount and of intersect per archetype pair
    for op in ("intersectionCount", "intersect", "union"):
        print(f"\n{op}: container pairs per second (x 1e6), rows = first operand")
        print("            " + " ".join(f"{n[:7]:>7s}" for n in names))
        for n1 in names:
            print(f"{n1:11s} " + " ".join(f"{table[f'{n1}/{n2}'][op]['ops_per_s'] / 1e6:7.0f}" for n2 in names))


if __name__ == "__main__":
    main()
