"""`python bench.py --gpus N` must start N ranks by itself (the driver calls it exactly like the
N = 1 run) and must never print a line whose n_gpus differs from what was asked for."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the -m gpu run of bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=280)
    # no GPU here: both ranks must fail loudly (no CPU fallback), after having been launched
    assert p.returncode != 0
    assert "--nproc-per-node=2" in p.stderr
    assert "No HIP GPUs" in p.stderr or "no HIP device" in p.stderr or "NODEVICE" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")], "a result line was printed without a GPU"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpus_2_line_schema_on_the_gpu_box():
    """`bench.py --gpus 2` end to end on whatever the box has (one GPU: the two ranks share it and reduce over gloo —
    the complete N > 1 code path, flagged "oversubscribed"): the line must carry the per-query-collective headline,
    the bucketed throughput mode beside it, BASELINE configs[3] strong-scaled (`strong_scaling`) and the in-library
    group path (`group_api`), every reduced result having been checked inside the run."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    detail = "bench_detail_test_n2.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--shards", "128", "--repeats", "2", "--cold-sets", "1",
           "--shards4-total", "48", "--shards4-mixed-total", "20", "--queries4", "3", "--detail", detail]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=850)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-2000:])
    # the stdout line is the COMPACT one (the driver could not parse round 4's 21.6 KB line); the verbose object is in the detail file
    assert len(lines[0].encode()) < 8192, len(lines[0])
    c = json.loads(lines[0])
    _check_compact(c, 2, 20, 48)
    r = json.load(open(os.path.join(ROOT, detail)))
    os.remove(os.path.join(ROOT, detail))
    assert r["value"] == c["value"] and r["n_gpus"] == 2 and r["config"]["ranks"] == 2 and r["steps"] == 20 and r["scaling"] == "weak" and r["unit"] == "set-ops/s"
    assert r["config"]["backend"] in ("rccl", "gloo") and r["config"]["collectives_per_step"] == 1
    assert r["throughput_mode_bucketed"]["steps_per_collective"] == 16 and r["per_query"]["collective_per_step_pipelined_ms_per_step"] > 0
    s = r["strong_scaling"]
    assert "error" not in s, s
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["backend"] == r["config"]["backend"] and s["rank0"]["shards_total"] == 48 and s["rank0"]["shards_this_rank"] == 24
    assert s["ms_per_query_pipelined"] > 0 and s["ms_per_query_host_add"] > 0 and s["rank0"]["collectives"] == 2 + 3 + 3
    assert [v["id"] for v in s["variants"]] == ["dense", "loguniform"] and s["variants"][1]["shards_total"] == 20 and s["variants"][1]["shards_this_rank"] == 10
    assert s["variants"][1]["parity"].startswith("every one of this rank's 10 shards bit-exact") and s["variants"][1]["kernel_us"] > 0
    g = r["group_api"]
    assert g["members"] == 2 and "host" in g["modes"] and g["count_matrix"]["scaling"] == "strong" and sum(g["count_matrix"]["shards_per_member"]) == 48
    # the CPU leg is part of EVERY line (same keys as at N = 1): rank 0 times it while the other ranks sleep
    cb = r["cpu_baseline"]
    assert r["roofline"]["frac"] > 0 and cb["kind"] == "port" and cb["unit"] == "set-ops/s" and cb["value"] > 0 and cb["cores"] >= 1
    assert "rank 0" in cb["sample"] and cb["single_thread_set_ops_per_s"] > 0


def _check_compact(c, n, steps, shards4_total):
    """what the driver reads: the contract keys + roofline + cpu_baseline, and the N > 1 sections in short form"""
    assert c["n_gpus"] == n and c["config"]["ranks"] == n and c["steps"] == steps and c["scaling"] == "weak" and c["unit"] == "set-ops/s"
    assert c["value"] > 0 and c["ms_per_step"] > 0 and c["higher_is_better"] is True and c["dtype"] == "u64" and c["data"] == "synthetic"
    assert c["config"]["workload"].startswith("configs[1]") and c["config"]["collectives_per_step"] == 1
    # who issues the per-step collective (the library's own RCCL communicator when every rank has a device of its own; ranks that
    # share a device reduce over gloo through torch) and what the loop costs the launching thread; the headline is the median of
    # the timed regions
    assert c["config"]["collective_path"] in ("torch.distributed", "library-rccl (fbk_comm_all_reduce_u64)")
    assert (c["config"]["backend"] == "rccl") or c["config"]["collective_path"] == "torch.distributed"
    assert c["host_enqueue_us_per_step"] > 0 and c["timed_regions"] >= 1 and c["ms_per_step_first_region"] > 0
    rf, cb = c["roofline"], c["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["frac"] > 0 and rf["peak"] == 8000.0 and rf["kernel"].startswith("k_icount_dense")
    assert cb["kind"] == "port" and cb["unit"] == "set-ops/s" and cb["value"] > 0 and cb["cores"] >= 1 and "rank 0" in cb["sample"]
    s = c["strong_scaling"]
    assert "error" not in s, s
    assert s["n_gpus"] == n and s["ms_per_query"] > 0 and s["rank0"]["shards_total"] == shards4_total
    assert [v["id"] for v in s["variants"]] == ["dense", "loguniform"] and all(v["ms_per_query"] > 0 and v["parity"].startswith("exact") for v in s["variants"])
    assert c["throughput_mode_bucketed"]["steps_per_collective"] == 16 and c["per_query"]["one_cell_ms_per_step"] > 0
    g = c["group_api"]
    assert g["members"] == n and "host" in g["modes"] and g["count_matrix"]["scaling"] == "strong"


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpus_8_oversubscribed_line_on_the_gpu_box():
    """`bench.py --gpus 8` as the driver will run it on an 8-GPU node, here with eight ranks on whatever the box has (gloo
    collectives when there are fewer devices than ranks): rank-count-dependent code — shards_for_rank remainders (60 dense and
    20 log-uniform shards over 8 ranks), the per-query collectives with 8 participants, the group child process with 8 members,
    the CPU leg on rank 0 with seven ranks asleep — has then run once, and the ONE stdout line stays under the size the driver
    parses."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    detail = "bench_detail_test_n8.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "3", "--shards", "32", "--repeats", "2", "--cold-sets", "1",
           "--shards4-total", "60", "--shards4-mixed-total", "20", "--queries4", "2", "--detail", detail]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=850)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 and os.path.isdir(os.path.join(ROOT, "gpurun_out")):  # (post-mortem: pytest abbreviates the message below)
        open(os.path.join(ROOT, "gpurun_out", "bench_n8_stderr.log"), "w").write(p.stderr)
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-3000:])
    assert len(lines[0].encode()) < 8192, len(lines[0])
    c = json.loads(lines[0])
    _check_compact(c, 8, 20, 60)
    r = json.load(open(os.path.join(ROOT, detail)))
    os.remove(os.path.join(ROOT, detail))
    s = r["strong_scaling"]
    assert s["rank0"]["shards_this_rank"] == 8 and s["variants"][1]["shards_this_rank"] == 3  # ranks 0..3 own 8 of the 60 / 3 of the 20
    assert r["group_api"]["members"] == 8 and sum(r["group_api"]["count_matrix"]["shards_per_member"]) == 60
    assert r["cpu_baseline"]["cores"] >= 1 and "the other 7 ranks" in r["cpu_baseline"]["sample"]
