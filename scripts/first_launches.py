#!/usr/bin/env python3
"""Why are the first ~10 launches of the encoded-row count matrix after an upload 25 % slower than the sustained rate?
(bench.py `c4.loguniform_slice.kernel_us_first_launches` 1 246 us against 990 sustained, round 5.)

In bench.py those "first launches" come after the upload AND after seconds of host-only work (the oracle checks every shard)
during which the device is idle.  Three experiments on ONE resident prepared query (SURVEY 8d's log-uniform configs[3] slice),
every launch timed by itself (library option time_kernels: HIP events right around k_count_matrix_fusedq):

  A  idle sensitivity, NO upload:  sleep t in {0, 1 ms, 10 ms, 50 ms, 200 ms, 1 s, 3 s}, then 24 launches
  B  upload WITHOUT idling:        re-upload the batch + prepare a new query, launch at once (24 launches)
  C  upload, then the device kept busy for 30 ms by another kernel (the dense count), then 24 launches
  D  idle 1 s, then the device kept busy for 30 ms by another kernel, then 24 launches

The shader clock is sampled from sysfs (pp_dpm_sclk: the active level) before each group where the box exposes it.

    python scripts/first_launches.py [--shards 512] [--out profiles/r06_first_launches.json]
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402


def sclk():
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(p):
                if "*" in line:
                    return line.strip()
        except OSError:
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=512)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--out", default="")
    ap.add_argument("--fresh-sleep", type=float, default=2.0, help="seconds the device idles before each group-H upload (bench.py: ~9 s of host-side data generation)")
    ap.add_argument("--only-fresh", action="store_true", help="group H only (for a run under rocprofv3 --kernel-trace: the same launches on the profiler's clock)")
    args = ap.parse_args()
    import torch

    from featurebase_amd.roaring import Context

    n, n_a, n_b = args.shards, 32, 32
    d4, p4, nr4, g4, fd4, fp4, nbytes = D.config3_flat_subprocess(n, n_a + n_b, 4000, config4=True)
    ctx = Context(0)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    ga, gb, fidx = np.ascontiguousarray(g4[:, :n_a]), np.ascontiguousarray(g4[:, n_a:]), np.arange(n)

    def resident():
        b, f = ctx.upload_flat(d4, p4, nr4), ctx.upload_flat(fd4, fp4, n)
        q = ctx.prepare_count_matrix(b, ga, b, gb, f, fidx, keep_per_shard=True)
        return b, f, q

    # the "other kernel": the dense count over 256 row pairs (10 us a launch)
    w = D.dense_rows(2 * 256, 0.5, 7)
    Aden = ctx.upload_dense(w)
    plan = ctx.plan(Aden, np.arange(256) * 2, Aden, np.arange(256) * 2 + 1)

    def busy(seconds):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                plan.intersection_count()
            torch.cuda.synchronize()

    # the shader clock right behind each launch: torch.cuda._sleep spins for a given number of SHADER cycles (clock64 = s_memtime on
    # gfx9), so its duration between two events is cycles / clock
    PROBE = 100000
    pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mhz_log = []

    def launches(q, k):
        ctx.set_option("time_kernels", 1)
        out, clk = [], []
        try:
            for _ in range(k):
                q.run()
                with torch.cuda.stream(st):
                    pe0.record(st)
                    torch.cuda._sleep(PROBE)
                    pe1.record(st)
                torch.cuda.synchronize()
                out.append(ctx.get_option("last_kernel_ns") / 1e3)
                clk.append(round(PROBE / (pe0.elapsed_time(pe1) * 1e3)))
        finally:
            ctx.set_option("time_kernels", 0)
        mhz_log.append(clk)
        return [round(x, 1) for x in out]

    b, f, q = resident()
    q.run()
    torch.cuda.synchronize()
    ref = q.read()
    res = {"shards": n, "encoded_bytes": int(nbytes), "launches_per_group": args.launches, "groups": []}

    def group(name, us, clk):
        s = sorted(us)
        g = {"name": name, "sclk_before": clk, "first": us[0], "mean_first_5": round(float(np.mean(us[:5])), 1), "median_last_10": round(float(np.median(us[-10:])), 1),
             "ratio_first5_to_sustained": round(float(np.mean(us[:5]) / np.median(us[-10:])), 3), "min": s[0], "us": us, "shader_MHz_behind_each_launch": mhz_log[-1]}
        res["groups"].append(g)
        print(f"    us  {[int(x) for x in us]}\n    MHz {mhz_log[-1]}", flush=True)
        print(f"{name:58s} sclk {clk}  first {us[0]:8.1f}  first5 {g['mean_first_5']:8.1f}  sustained {g['median_last_10']:8.1f}  ratio {g['ratio_first5_to_sustained']:.3f}", flush=True)

    launches(q, 40)  # settle
    for idle in (() if args.only_fresh else (0.0, 0.001, 0.01, 0.05, 0.2, 1.0, 3.0)):
        torch.cuda.synchronize()
        time.sleep(idle)
        clk = sclk()
        group(f"A idle {idle * 1e3:g} ms, no upload", launches(q, args.launches), clk)
    if not args.only_fresh:
        # B: upload without idling
        busy(0.05)
        q.free()
        b.free()
        f.free()
        b, f, q = resident()
        clk = sclk()
        group("B re-upload + prepare, launched at once", launches(q, args.launches), clk)
        assert (q.read() == ref).all()
        # C: upload, busy 30 ms on another kernel, then launch
        q.free()
        b.free()
        f.free()
        b, f, q = resident()
        busy(0.03)
        clk = sclk()
        group("C re-upload + prepare, 30 ms of another kernel, then", launches(q, args.launches), clk)
        # D: idle 1 s, busy 30 ms, launch
        time.sleep(1.0)
        busy(0.03)
        clk = sclk()
        group("D idle 1 s, 30 ms of another kernel, then", launches(q, args.launches), clk)
        # F: what bench.py does between the upload and its "first launches": seconds of host work on every core, device idle
        import threading

        def burn(sec):
            a = np.random.default_rng(1).random((384, 384))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < sec:
                a = a @ a
                a /= np.abs(a).max()

        th = [threading.Thread(target=burn, args=(3.0,)) for _ in range(min(64, os.cpu_count() or 8))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        clk = sclk()
        group("F 3 s of host work on up to 64 threads (device idle), then", launches(q, args.launches), clk)
        # G: 5 launches only, after 3 s idle (bench.py's kernel_us_first_launches protocol), three times
        for rep in range(3):
            time.sleep(3.0)
            us = launches(q, 5)
            res["groups"].append({"name": f"G 3 s idle, then 5 launches (#{rep})", "us": us, "shader_MHz_behind_each_launch": mhz_log[-1]})
            print(f"G 3 s idle, then 5 launches (#{rep}): {us}  MHz {mhz_log[-1]}", flush=True)
    # H: what the strong-scaling loop of bench.py does per slice — rows uploaded into FRESH device memory (the earlier batches stay
    # resident, so nothing comes back from the pool), a new prepared query, its FIRST run (window index, shadows, program are built
    # by their own kernels in front of it) — the launch that took 29-31 ms in profiles/r06_kernel_trace_by_grid.csv
    held = []
    for rep in range(3):
        time.sleep(args.fresh_sleep)
        t_up = time.perf_counter()
        b2, f2 = ctx.upload_flat(d4, p4, nr4), ctx.upload_flat(fd4, fp4, n)
        q2 = ctx.prepare_count_matrix(b2, ga, b2, gb, f2, fidx, keep_per_shard=True)
        t_up = time.perf_counter() - t_up
        ctx.set_option("time_kernels", 1)
        t_run = time.perf_counter()
        q2.run()
        torch.cuda.synchronize()
        t_run = time.perf_counter() - t_run
        first_ns = ctx.get_option("last_kernel_ns")
        ctx.set_option("time_kernels", 0)
        us = launches(q2, 6)
        held += [b2, f2, q2]
        res["groups"].append({"name": f"H {args.fresh_sleep:g} s idle, fresh memory, first run of a new query (#{rep})", "upload_prepare_s": round(t_up, 4), "first_run_wall_ms": round(t_run * 1e3, 2),
                              "first_run_kernel_us": round(first_ns / 1e3, 1), "next_us": us})
        print(f"H {args.fresh_sleep:g} s idle, fresh memory (#{rep}): upload + prepare {t_up * 1e3:.1f} ms, first run wall {t_run * 1e3:.2f} ms, its k_count_matrix_fusedq {first_ns / 1e3:.1f} us, next {us}", flush=True)
    # E: back-to-back enqueue (no host round trip between launches): 24 launches between two events, after 1 s idle and sustained
    for name, idle in (() if args.only_fresh else (("E 24 launches enqueued back to back after 1 s idle", 1.0), ("E 24 launches enqueued back to back, sustained", 0.0))):
        time.sleep(idle)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(args.launches):
            q.run()
        e1.record(st)
        torch.cuda.synchronize()
        res["groups"].append({"name": name, "us_per_launch_incl_memset_and_reduce": round(e0.elapsed_time(e1) * 1e3 / args.launches, 1)})
        print(f"{name:58s} {res['groups'][-1]['us_per_launch_incl_memset_and_reduce']:8.1f} us per launch (memset + kernel + reduce)", flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
