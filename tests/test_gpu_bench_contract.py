"""bench.py's driver-facing contract, end to end on the GPU (a short run): ONE JSON line on stdout with
the fields the task statement names - metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload - plus the `roofline` and
`cpu_baseline` objects, and the consistency relations between them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields_and_is_self_consistent():
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--host-steps", "10")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    B = cfg["streams_per_gpu"]
    assert sum(cfg["engine_sizes"]) == B and len(cfg["engine_sizes"]) == cfg["engines_per_gpu"]
    # value = frames of the timed region / its wall time
    assert abs(d["value"] - B * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["tracked_ok"] is True and d["min_iou_vs_truth"] > 0.5
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(rf["achieved"] - rf["flops_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e12) < 1e-6 * rf["achieved"]
    assert 0.2 < rf["frac"] < 0.6, rf                      # a broken build shows here first
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "frames/s" and cb["sample"]
    pi = d["pcie_inclusive"]
    assert pi["tracked_ok"] is True and 0.5 < pi["vs_hbm_resident"] < 1.1
    assert [k["kernel"] for k in d["byte_kernels"]["kernels"]][:2] == ["nv12_to_rgb8_kernel"] * 2
    assert d["reference_style"]["window"] == 100


def test_bench_single_stream_and_planned_engines():
    d = _run("--streams", "1", "--groups", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-host-leg",
             "--no-profile")
    assert d["config"]["engine_sizes"] == [1] and d["tracked_ok"] and "roofline" not in d and "cpu_baseline" not in d
    d = _run("--streams", "33", "--engines", "auto", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
             "--no-host-leg", "--no-profile")
    assert d["config"]["engine_sizes"] == [30, 3] and d["tracked_ok"]
