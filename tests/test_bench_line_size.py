"""The ONE stdout line of bench.py must stay parseable by the driver: round 4's line had grown to 21.6 KB and
BENCH_r04.json recorded "parsed": null.  Builds the line from canned verbose result objects (the round-4 line as the
builder kept it under profiles/, with and without the strong-scaling variants added in round 5) — no GPU needed."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r04_bench_n1.json")


def _full():
    return json.load(open(CANNED))


def test_canned_round4_line_was_too_long():
    assert len(json.dumps(_full())) > 16 * 1024


def test_compact_line_size_and_contract_keys():
    full = _full()
    s = bench_line.dumps_line(full)
    assert "\n" not in s
    assert len(s.encode()) <= bench_line.LINE_TARGET, len(s)
    c = json.loads(s)
    for k in bench_line.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == round(full["ms_per_step"], 6)
    assert c["n_gpus"] == 1 and c["scaling"] == "weak" and c["vs_baseline"] is None
    assert c["config"]["workload"].startswith("configs[1]")
    rf = c["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-4 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert rf["traffic"] == full["roofline"]["traffic"]
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 256 and cb["unit"] == "set-ops/s" and cb["value"] > 0 and cb["sample"]
    # the sample note keeps its head (what was timed) AND its tail (where: "... on rank 0 ..." at N > 1)
    full2 = _full()
    full2["cpu_baseline"]["sample"] += "; timed on rank 0 (its 1024 shards = the N = 1 workload) after the timed GPU regions, the other 7 ranks asleep in a host barrier"
    c2 = json.loads(bench_line.dumps_line(full2))
    assert c2["cpu_baseline"]["sample"].startswith("full workload") and "rank 0" in c2["cpu_baseline"]["sample"] and len(c2["cpu_baseline"]["sample"]) < 340
    # every secondary entry survives as {id, kernel, kernel_us, frac, parity}
    assert len(c["secondary"]) == len(full["secondary"])
    for e, f in zip(c["secondary"], full["secondary"]):
        assert set(e) >= {"id", "kernel", "kernel_us", "frac", "parity"}
        assert e["kernel"] == f["kernel"][:40] and e["parity"].startswith("exact")
        assert abs(e["frac"] - f.get("kernel_frac", f["frac"])) < 1e-3
    assert c["strong_scaling"]["rank0"]["shards"] == 8192


def test_compact_line_with_round5_additions_and_eight_ranks():
    full = _full()
    for e in full["secondary"]:
        e["id"] = "c9.some_identifier_of_usual_size"
    # four more secondary entries and two strong-scaling variants, as bench.py emits them from round 5 on
    full["secondary"] += [copy.deepcopy(full["secondary"][2]) for _ in range(4)]
    v = {"id": "mixed", "ms_per_query_pipelined": 12.3456789, "set_ops_per_s": 1.2345678e10, "kernel": "k_count_matrix_fused<...>", "kernel_us_max_over_ranks": 1234.5,
         "kernel_frac": 0.3456789, "parity": "every one of this rank's 1024 shards bit-exact against the oracle"}
    full["strong_scaling"]["variants"] = [dict(v, id="dense"), v]
    full["n_gpus"] = 8
    full["group_api"] = {"members": 8, "devices": list(range(8)), "modes": {m: {"ms_per_step": 0.0412345, "set_ops_per_s": 3.1e9, "matrix_ms": 1.5} for m in ("host", "peer", "rccl")}}
    s = bench_line.dumps_line(full)
    assert len(s.encode()) <= bench_line.LINE_LIMIT, len(s)
    c = json.loads(s)
    assert c["n_gpus"] == 8 and "roofline" in c and "cpu_baseline" in c
    assert [x["id"] for x in c["strong_scaling"]["variants"]] == ["dense", "mixed"]


def test_error_entries_stay_short():
    full = _full()
    full["secondary"] = [{"error": "RuntimeError: " + "x" * 5000, "traceback": ["y" * 500] * 6}]
    full["strong_scaling"] = {"error": "z" * 5000, "traceback": ["y" * 500] * 6}
    s = bench_line.dumps_line(full)
    assert len(s.encode()) <= bench_line.LINE_TARGET
    assert "roofline" in json.loads(s)
