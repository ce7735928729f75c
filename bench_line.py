"""The ONE line bench.py prints on stdout, built from the verbose result object.

The driver parses the last stdout line of `python bench.py ...` as JSON; round 4's line had grown to 21.6 KB and was
not parsed (BENCH_r04.json: "parsed": null), so the line is now a COMPACT view (target <= 6 KB, hard limit 8 KB, see
tests/test_bench_line_size.py) and everything else goes to `bench_detail.json` next to bench.py (and, when the
directory exists, `gpurun_out/bench_detail.json`).  The contract keys are copied unchanged; secondary entries keep
{id, kernel, kernel_us, frac, parity}: `frac` there is the named kernel alone against 8 TB/s when the library timed
it (option time_kernels), otherwise the whole call.
"""
import json
import os

LINE_TARGET = 6 * 1024
LINE_LIMIT = 8 * 1024

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")

TIMING_NOTE = ("secondary[].kernel_us/frac: the named kernel alone (HIP events recorded by the library around the launch, median), frac = "
               "algorithmic bytes / kernel_us / 8 TB/s; call-level times, distributions, CPU legs and notes are in bench_detail.json")


def _r(x, nd=4):
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        ax = abs(x)
        if ax >= 1e6:
            return float(f"{x:.5g}")
        return round(x, nd)
    return x


def _short(s, n):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _head_tail(s, head, tail):
    """a long note cut in the middle (the CPU leg's sample says WHAT was timed first and WHERE last)"""
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= head + tail + 5 else s[:head] + " ... " + s[-tail:]


def _median(d):
    if isinstance(d, dict):
        return d.get("median")
    return d


def _parity_word(p):
    """'every one of the 256 shards bit-exact against the oracle' -> 'exact:256 shards'"""
    if not p:
        return None
    p = str(p)
    if "bit-exact" in p or "equal" in p:
        import re

        m = re.search(r"(\d+) (?:per-shard matrices|shards)", p)
        return f"exact:{m.group(1)} shards" if m else "exact"
    return _short(p, 40)


def compact_secondary(entries):
    out = []
    for e in entries or []:
        if "error" in e:
            out.append({"error": _short(e["error"], 160)})
            continue
        k_us = _median(e.get("kernel_us"))
        c = {
            "id": e.get("id") or _short(e.get("name"), 48),
            "kernel": _short(e.get("kernel"), 40),
            "kernel_us": _r(k_us if k_us is not None else _median(e.get("gpu_us")), 2),
            "frac": _r(e.get("kernel_frac", e.get("frac")), 4),
            "parity": _parity_word(e.get("parity")),
        }
        if k_us is None:
            c["timed"] = "call"
        out.append(c)
    return out


def compact_strong(s):
    if not s:
        return None
    if "error" in s:
        return {"error": _short(s["error"], 200)}
    out = {
        "workload": _short(s.get("workload"), 150),
        "scaling": "strong",
        "n_gpus": s.get("n_gpus"),
        "backend": s.get("backend"),
        "ms_per_query": _r(s.get("ms_per_query_pipelined")),
        "set_ops_per_s": _r(s.get("set_ops_per_s")),
        "kernel_us_max_over_ranks": _r(s.get("kernel_us_max_over_ranks"), 1),
    }
    r0 = s.get("rank0") or {}
    out["rank0"] = {
        "shards_total": r0.get("shards_total"),
        "shards": r0.get("shards_this_rank"),
        "collectives": r0.get("collectives"),
        "kernel_frac": _r(r0.get("kernel_frac_of_8TBps")),
        "parity": _parity_word(r0.get("parity")),
    }
    for v in s.get("variants") or []:
        out.setdefault("variants", []).append({
            "id": v.get("id"),
            "ms_per_query": _r(v.get("ms_per_query_pipelined")),
            "set_ops_per_s": _r(v.get("set_ops_per_s")),
            "kernel": _short(v.get("kernel"), 40),
            "kernel_us_max_over_ranks": _r(v.get("kernel_us_max_over_ranks"), 1),
            "kernel_frac": _r(v.get("kernel_frac")),
            "parity": _parity_word(v.get("parity")),
        })
    return out


def compact_group(g):
    if not g:
        return None
    if "error" in g and len(g) == 1:
        return {"error": _short(g["error"], 200)}
    out = {"members": g.get("members"), "devices": g.get("devices")}
    modes = {}
    for name, m in (g.get("modes") or {}).items():
        if isinstance(m, dict):
            modes[name] = {k: _r(m[k]) for k in ("ms_per_step", "latency_ms_median", "set_ops_per_s", "matrix_ms") if isinstance(m.get(k), (int, float))}
            if isinstance(m.get("latency_ms"), dict):  # (the in-process form of the N = 1 line)
                modes[name]["latency_ms_median"] = _r(m["latency_ms"].get("median"))
            if "error" in m:
                modes[name] = {"error": _short(m["error"], 120)}
    out["modes"] = modes
    cm = g.get("count_matrix")
    if cm:
        out["count_matrix"] = {"scaling": cm.get("scaling"), "shards_per_member": cm.get("shards_per_member") if len(cm.get("shards_per_member") or []) <= 8 else None,
                               "ms_per_call": {k: _r(v.get("ms_per_call")) for k, v in (cm.get("modes") or {}).items() if isinstance(v, dict)}}
    if "error" in g:
        out["error"] = _short(g["error"], 160)
    return out


def compact_line(full: dict, detail_name: str = "bench_detail.json") -> dict:
    """The compact view of bench.py's result object (pure: no I/O)."""
    out = {k: _r(full.get(k), 6) if k != "value" else full.get(k) for k in CONTRACT_KEYS if k in full}
    cfg = dict(full.get("config") or {})
    if "op" in cfg:
        cfg["op"] = _short(cfg["op"], 200)
    out["config"] = cfg
    rf = full.get("roofline") or {}
    out["roofline"] = {
        "kernel": rf.get("kernel"),
        "bound": rf.get("bound"),
        "achieved": _r(rf.get("achieved"), 1),
        "peak": rf.get("peak"),
        "unit": rf.get("unit"),
        "frac": _r(rf.get("frac")),
        "traffic": rf.get("traffic"),
        "traffic_source": _short(rf.get("traffic_source"), 120),
        "kernel_us": _r(rf.get("kernel_us"), 2),
        "algorithmic_bytes": rf.get("algorithmic_bytes"),
    }
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {
            "value": _r(cb.get("value")),
            "unit": cb.get("unit"),
            "cores": cb.get("cores"),
            "kind": cb.get("kind"),
            "sample": _head_tail(cb.get("sample"), 170, 150),
            "single_thread": _r(cb.get("single_thread_set_ops_per_s")),
            "streaming_pass": _r((cb.get("streaming_pass") or {}).get("value")),
        }
    out["bits_scanned_GBps"] = _r(full.get("bits_scanned_GBps"), 1)
    for k, nd in (("ms_per_step_first_region", 6), ("host_enqueue_us_per_step", 2)):
        if isinstance(full.get(k), (int, float)):
            out[k] = _r(full[k], nd)
    if full.get("timed_regions") is not None:
        out["timed_regions"] = full["timed_regions"]
    d = full.get("ms_per_step_distribution")
    if d:
        out["ms_per_step_distribution"] = {k: _r(d.get(k), 5) for k in ("median", "p10", "p90", "n")}
    m = full.get("materialized")
    if m:
        out["materialized"] = {"kernel": m.get("kernel"), "kernel_us": _r(m.get("kernel_us"), 2), "frac": _r(m.get("frac"))}
    c = full.get("roofline_l3_cold")
    if c:
        out["roofline_l3_cold"] = {"working_set_MiB": c.get("working_set_MiB"), "kernel_us": _r(c.get("kernel_us"), 2), "frac": _r(c.get("frac"))}
    t2 = full.get("two_contexts")
    if t2 and "error" not in t2:
        out["two_contexts"] = {k: _r(t2.get(k), 5) for k in ("ms_per_step", "set_ops_per_s", "frac_of_8TBps")}
    tb = full.get("throughput_mode_bucketed")
    if tb:
        out["throughput_mode_bucketed"] = {k: _r(tb.get(k)) for k in ("ms_per_step", "set_ops_per_s", "steps_per_collective")}
    pq = full.get("per_query")
    if pq:
        out["per_query"] = {"one_cell_ms_per_step": _r(pq.get("collective_per_step_pipelined_ms_per_step")),
                            "host_readback_latency_ms": _r(_median(pq.get("collective_per_step_host_readback_latency_ms"))),
                            "host_add_latency_ms": _r(_median(pq.get("host_add_latency_ms"))) if "error" not in (pq.get("host_add_latency_ms") or {}) else None}
    if full.get("h2d_upload_GBps") is not None:
        out["h2d_upload_GBps"] = _r(full.get("h2d_upload_GBps"), 2)
    if full.get("secondary") is not None:
        out["secondary"] = compact_secondary(full["secondary"])
    if full.get("strong_scaling") is not None:
        out["strong_scaling"] = compact_strong(full["strong_scaling"])
    if full.get("group_api") is not None:
        out["group_api"] = compact_group(full["group_api"])
    out["note"] = TIMING_NOTE
    out["detail"] = detail_name
    return out


def dumps_line(full: dict, detail_name: str = "bench_detail.json") -> str:
    """The line itself; shrinks in steps (never past the contract keys) if a future entry pushes it over the limit."""
    c = compact_line(full, detail_name)
    s = json.dumps(c, separators=(",", ":"))
    for drop in ("group_api", "two_contexts", "materialized", "roofline_l3_cold", "ms_per_step_distribution", "note"):
        if len(s.encode()) <= LINE_TARGET:
            break
        c.pop(drop, None)
        s = json.dumps(c, separators=(",", ":"))
    if len(s.encode()) > LINE_TARGET and "secondary" in c:
        for e in c["secondary"]:
            e.pop("parity", None)
        s = json.dumps(c, separators=(",", ":"))
    if len(s.encode()) > LINE_LIMIT:
        c.pop("secondary", None)
        c.pop("strong_scaling", None)
        s = json.dumps(c, separators=(",", ":"))
    return s


def write_detail(full: dict, root: str, name: str = "bench_detail.json"):
    """The verbose object: beside bench.py and, on the GPU box, under gpurun_out/ (merged back by gpurun)."""
    paths = [os.path.join(root, name)]
    go = os.path.join(root, "gpurun_out")
    if os.path.isdir(go):
        paths.append(os.path.join(go, name))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written.append(p)
        except OSError:
            pass
    return written
