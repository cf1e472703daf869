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
    assert "source" in rf and "eager" in rf["source"]
    # `value` is the H2D-inclusive rate through the product's ingest; device_only and
    # full_frame are measured beside it in the same run
    assert "vt_group_enqueue_host" in cfg["ingest"]
    do, ff = d["device_only"], d["full_frame"]
    assert do["tracked_ok"] is True and 0.8 < do["headline_vs_device_only"] < 1.1
    assert ff["tracked_ok"] is True and 0.3 < ff["vs_headline"] < 1.1 and ff["h2d_copies_per_step"] <= 2 * cfg["engines_per_gpu"]
    assert abs(d["whole_frame_mfma_frac"] - d["value"] * d["gflop_per_frame"] / 1e3 / 2500.0) < 1e-9
    assert d["collective"]["world_size"] == 1 and d["per_rank_fps"] == [pytest.approx(d["value"])]
    assert cb["cpu_model"] and cb["tracked_ok"] is True
    xr = [k for k in d["byte_kernels"]["kernels"] if k["kernel"].startswith("gemm_bf16_xresid")]
    assert xr and all(k["flop_per_byte"] < 312 and 0.1 < k["frac_of_peak"] < 1.0 for k in xr)
    assert [k["kernel"] for k in d["byte_kernels"]["kernels"]][:2] == ["nv12_to_rgb8_kernel"] * 2
    assert d["reference_style"]["window"] == 100


def test_bench_gpus_2_on_one_gpu_fails_from_inside_the_ranks():
    """`python bench.py --gpus 2` starts its own ranks; on a one-GPU box each rank refuses with the reason"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    import gstreamer_vit_tracker_amd as vt
    if vt.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["collective"]["world_size"] == 2 and len(d["per_rank_fps"]) == 2
    else:
        assert r.returncode != 0 and "needs 2 devices" in r.stderr, r.stderr[-2000:]


def test_bench_single_stream_and_planned_engines():
    d = _run("--streams", "1", "--groups", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-host-leg",
             "--no-profile")
    assert d["config"]["engine_sizes"] == [1] and d["tracked_ok"] and "roofline" not in d and "cpu_baseline" not in d
    d = _run("--streams", "33", "--engines", "auto", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
             "--no-host-leg", "--no-profile")
    assert d["config"]["engine_sizes"] == [30, 3] and d["tracked_ok"]
