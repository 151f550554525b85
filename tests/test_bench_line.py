"""bench.py's stdout contract: ONE compact JSON line (< 4 KB) that the driver can parse, built from the full record.
Round 4's line carried every `also` workload (34.8 KB) and came back `parsed: null` (Missing #1)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r04_bench_n1.json")    # a full record as run() + main() assemble it

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _full():
    return json.load(open(CANNED))


def test_headline_is_short_and_round_trips():
    full = _full()
    assert len(json.dumps(full)) > 30000                         # the record that broke the driver
    s = bench.headline_line(full, "bench_detail.json")
    assert "\n" not in s and len(s) < 4096, len(s)
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k
    assert d["metric"] == bench.METRIC and d["unit"] == "evals/s" and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    assert d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    # value = walkers * steps / time, on the line's own numbers
    assert d["value"] == pytest.approx(d["n_gpus"] * d["config"]["walkers_per_gpu"] / (d["ms_per_step"] * 1e-3), rel=1e-4)
    assert "model" not in d["config"] and d["config"]["workload"].startswith("c4")
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["parity_in_bench"]["within_tolerance"] is True
    assert set(d["also"]) == set(full["also"]) and all(isinstance(v, float) for v in d["also"].values())
    assert d["detail_file"] == "bench_detail.json"


def test_headline_stays_short_whatever_the_record_carries():
    full = _full()
    full["config"]["overrides"] = {f"option_{i}": float(i) for i in range(40)}
    full["also"] = {f"workload_{i}": {"value": 1.0e6 + i, "config": {"x": "y" * 5000}} for i in range(60)}
    full["cpu_baseline"]["sample"] = "s" * 10000
    full["cpu_baseline"].pop("sample_short", None)
    s = bench.headline_line(full, None)
    assert len(s) < 4096
    d = json.loads(s)
    assert all(k in d for k in CONTRACT)


def test_headline_of_a_multi_rank_record_keeps_the_swap_verdict():
    full = _full()
    full.update(n_gpus=8, swap_replay_ok=True, cross_rank_swaps=1234)
    full["config"]["rccl"] = {"ranks": 8, "version": "2.26.6", "transport": "rccl_allgather", "control_plane": "gloo",
                              "init_s": 3.25, "note": "n" * 3000}
    full.pop("cpu_baseline")
    full.pop("also")
    d = json.loads(bench.headline_line(full, "bench_detail.json"))
    assert d["swap_replay_ok"] is True and d["cross_rank_swaps"] == 1234
    assert d["config"]["rccl"] == {"ranks": 8, "version": "2.26.6", "transport": "rccl_allgather", "control_plane": "gloo",
                                   "init_s": 3.25}
    assert "cpu_baseline" not in d and "also" not in d


def test_nan_never_reaches_the_line():
    full = _full()
    full["roofline"]["frac_at_clock"] = float("nan")
    s = bench.headline_line(full, None)
    assert "NaN" not in s
    assert json.loads(s)["roofline"]["frac_at_clock"] is None


def test_headline_of_the_round_6_record_carries_the_cpu_only_reference_baseline():
    """The committed round-6 record (profiles/r06_bench_n1_detail.json -> profiles/r06_bench_n1.json): the compact line
    built from the full record equals the line the run printed, carries a `cpu_baseline` of kind "reference" -- the
    reference's own calc_likelihood on the HOST CORES ONLY (oracle/_ref/cpu_o2), the -O0 class and the C port beside it --
    and stays far below 4 KB."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_n1_detail.json")))
    printed = open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")).read().strip()
    assert "\n" not in printed and len(printed) < 4096
    d = json.loads(printed)
    again = json.loads(bench.headline_line(full, d["detail_file"]))
    assert again == d
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["build"] == "cpu_o2" and c["cores"] >= 1 and c["value"] > 0
    assert "host cores only" in c["sample"] and "GPU" not in c["sample"]
    # per core at all cores ~ per core alone (round 5's GPU-assisted build lost 40 % here), and -O0 is the slower class
    assert 0.85 * c["single_core"] <= c["per_core"] <= 1.05 * c["single_core"]
    assert c["o0"]["value"] < c["value"] and c["o0"]["cores"] == c["cores"]
    assert c["port"]["cores"] == c["cores"] and 0.5 < c["reference_over_port"] < 1.2
    assert d["roofline"]["counters_file"] == "profiles/r06_counters.json" and d["roofline"]["frac"] is not None
    p = d["parity_in_bench"]
    assert p["n"] == d["config"]["walkers_per_gpu"] == 8192 and p["within_tolerance"] and p["within_kappa_rule"]
    assert set(d["also"]) == set(full["also"]) and {"c4d", "c5d", "c4full", "c5full"} <= set(d["also"])
    # the whole 8-GPU jobs on one GPU run at the per-GPU shard's rate (the weak-scaling premise, measured)
    assert abs(d["also"]["c4full"] / d["value"] - 1.0) < 0.05 and abs(d["also"]["c5full"] / d["also"]["c5"] - 1.0) < 0.05


def test_a_failed_multi_rank_run_still_prints_one_json_line(tmp_path):
    """N > 1: whatever stops the run -- here: two ranks under a launcher's environment on a machine without a GPU -- rank 0
    prints ONE JSON line (value null, error, how far it got, what of config.rccl was known) before the non-zero exit; the
    bare `--gpus 2` form (bench.py as its own launcher) says so from the launcher."""
    import socket
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU (the failure it provokes is 'no GPU visible')")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--also", ""],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=str(tmp_path)))
    outs = [q.communicate(timeout=300)[0] for q in procs]
    assert all(q.returncode != 0 for q in procs)
    lines = [l for l in outs[0].splitlines() if l.strip()]
    assert len(lines) == 1 and outs[1].strip() == ""                 # rank 0 alone speaks, once
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "no GPU" in d["error"] and d["reached"] == "process_group_formed"
    assert d["config"]["rccl"]["control_plane"] == "gloo" and d["metric"] == bench.METRIC
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=120)
    d = json.loads(r.stdout.strip())
    assert r.returncode == 2 and d["value"] is None and d["reached"] == "launcher" and "visible GPUs" in d["error"]
