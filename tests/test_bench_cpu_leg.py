"""bench.py's cpu_baseline leg on a small sample (no GPU): the C restatement built for this host
(vectorised and with auto-vectorisation off), timed on 1 and all threads, totals checked against
numpy popcounts — and the keys the bench contract asks for in the cpu_baseline object."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_leg():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(5)
    wa = rng.integers(0, 2**64, (8, 16, 1024), dtype=np.uint64)
    wb = rng.integers(0, 2**64, (8, 16, 1024), dtype=np.uint64)
    r = bench.cpu_baseline(wa, wb, budget_s=0.6)
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in r
    assert r["kind"] == "port" and r["unit"] == "set-ops/s" and r["cores"] >= 1 and r["value"] > 0
    assert r["single_thread_set_ops_per_s"] > 0
    assert r["single_thread_no_autovectorize_set_ops_per_s"] is None or r["single_thread_no_autovectorize_set_ops_per_s"] > 0
    # the last timed call made `passes` passes over the workload: its total is a multiple of one pass
    one_pass = int(np.bitwise_count(wa & wb).sum())
    assert r["total_count"] % one_pass == 0 and r["total_count"] > 0
