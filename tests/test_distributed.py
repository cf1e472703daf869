"""N>1 path on CPU: world_size-2 gloo. The only collective of the hot path is the start-up weight
broadcast; streams are sharded across ranks with no per-frame exchange."""
import hashlib
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, blob_path, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    import gstreamer_vit_tracker_amd.distributed as vd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = vd.broadcast_weights(blob_path if rank == 0 else None, device="cpu")
        digest = hashlib.sha256(t.numpy().tobytes()).hexdigest()
        shard = vd.shard_streams(5, rank, world)
        agg = vd.aggregate_max_time(0.5 + rank)
        plan = vd.plan_rank(rank, world, 6, 64)
        thr = vd.aggregate_throughput(6 * 10, 0.5 + rank)       # bench.py's own aggregation
        per_rank = vd.gather_per_rank(100.0 + rank)
        _, rec = vd.timed_broadcast_weights(blob_path if rank == 0 else None, device="cpu")
        q.put((rank, digest, t.numel(), shard, agg, plan, thr, per_rank, rec))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_stream_sharding_gloo(vt, weights_tiny):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, weights_tiny, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = hashlib.sha256(open(weights_tiny, "rb").read()).hexdigest()
    assert [g[1] for g in got] == [want, want]
    assert got[0][2] == os.path.getsize(weights_tiny)
    # 5 streams over 2 ranks: disjoint, complete, balanced within one
    s0, s1 = got[0][3], got[1][3]
    assert sorted(s0 + s1) == list(range(5)) and abs(len(s0) - len(s1)) <= 1
    # timing is the max over ranks
    assert got[0][4] == got[1][4] == pytest.approx(1.5)
    # bench.py's per-rank plan: own clip per rank, disjoint global stream ids, same phases
    p0, p1 = got[0][5], got[1][5]
    assert p0["clip_seed"] != p1["clip_seed"] and p0["phase"] == p1["phase"] == [0, 1, 2, 3, 4, 5]
    assert sorted(p0["global_stream_ids"] + p1["global_stream_ids"]) == list(range(12))
    # whole-job throughput = frames of all ranks / slowest rank's time, identical on every rank
    for g in got:
        assert g[6]["frames"] == 120 and g[6]["seconds"] == pytest.approx(1.5)
        assert g[6]["frames_per_s"] == pytest.approx(80.0)
        # per-rank rates in rank order on every rank; the collective record proves the group's size
        assert g[7] == [100.0, 101.0]
        assert g[8]["backend"] == "gloo" and g[8]["world_size"] == 2 and g[8]["collectives_per_frame"] == 0
        assert g[8]["broadcast_bytes"] == os.path.getsize(weights_tiny) and g[8]["broadcast_ms"] > 0
    assert got[0][8]["broadcast_ms"] == got[1][8]["broadcast_ms"]      # max over ranks, same on both


def test_bench_self_launches_its_ranks_gloo_dry_run(vt, weights_tiny):
    """`python bench.py --gpus 2` with no RANK in the environment starts its two ranks itself
    (torch.distributed.run as a child; the parent imports neither torch nor HIP), relays rank 0's
    one JSON line and returns the children's status. --dry-run keeps the ranks on the CPU (gloo):
    launcher, collective record, per-rank plan and aggregation are what is exercised."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--workload", "tiny", "--streams", "3", "--dry-run"], capture_output=True, text=True,
                       cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["broadcast_bytes"] == os.path.getsize(weights_tiny)
    # rank r "took" 0.5 + r/4 s for 3 streams x 10 steps: whole job = 60 frames / 0.75 s
    assert d["per_rank_fps"] == pytest.approx([60.0, 40.0]) and d["value"] == pytest.approx(80.0)
    assert d["global_stream_ids_rank0"] == [0, 1, 2]
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    _check_masks(d, 2)


def _check_masks(d, world):
    """every rank pinned itself before torch / HIP loaded: the masks the ranks REALLY run under (sched_getaffinity
    inside each rank, gathered) are disjoint, non-empty, and what the rank's own record says"""
    from gstreamer_vit_tracker_amd import placement
    masks = d["cpu_affinity_by_rank"]
    assert [m["rank"] for m in masks] == list(range(world))
    allowed = len(os.sched_getaffinity(0))
    seen = set()
    for m in masks:
        cpus = set(m["running_on"])
        assert cpus and m["n"] == len(cpus) and placement.parse_cpulist(m["cpus"]) == sorted(cpus)
        if allowed >= world:
            assert not (cpus & seen), masks
        seen |= cpus
    assert d["config"]["cpu_affinity"] == {k: masks[0][k] for k in ("cpus", "n", "source")}


def test_externally_launched_ranks_set_the_ipc_mode_themselves(vt, weights_tiny):
    """the driver's route: `python -m torch.distributed.run ... bench.py --gpus N` with the variable absent from
    the environment - run_rank sets HSA_ENABLE_IPC_MODE_LEGACY=0 before torch / HIP load (dmabuf IPC is the only
    mode this pool's driver supports; RCCL fails with hipIpcGetMemHandle otherwise), and the line records it"""
    import json
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "tiny", "--streams", "1",
                        "--groups", "1", "--dry-run"], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["collective"]["world_size"] == 2
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # BASELINE.json configs[3] (one stream per GPU): the N-rank line shape of `--streams 1 --groups 1`; literal only on 8 GPUs
    assert d["config"]["streams_per_gpu"] == 1 and d["config"]["engines_per_gpu"] == 1
    assert d["config"]["cfg4_shape"] is True and d["config"]["cfg4_literal"] is False
    assert d["per_rank_fps"] == pytest.approx([8.0, 4 / 0.75]) and len(d["per_rank_fps"]) == 2
    _check_masks(d, 2)


def test_eight_ranks_one_stream_each_is_the_literal_cfg4_line(vt, weights_tiny):
    """BASELINE.json configs[3] taken literally - 8 independent streams, ONE per GPU, on 8 GPUs - as far as a CPU box can
    rehearse it: `bench.py --gpus 8 --streams 1 --groups 1 --dry-run` self-launches eight gloo ranks; the line says
    cfg4_literal, carries eight per-rank rates and a collective record of world size 8, the eight ranks hold disjoint global
    stream ids and (where the box has at least eight CPUs) disjoint CPU masks"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "1",
                        "--workload", "tiny", "--streams", "1", "--groups", "1", "--dry-run"], capture_output=True, text=True,
                       cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["collective"]["world_size"] == 8
    assert d["collective"]["backend"] == "gloo" and d["collective"]["broadcast_bytes"] == os.path.getsize(weights_tiny)
    assert d["config"]["cfg4_literal"] is True and d["config"]["cfg4_shape"] is True and d["config"]["streams_per_gpu"] == 1
    # rank r "took" 0.5 + r/4 s for 1 stream x 6 steps: whole job = 48 frames / 2.25 s (the slowest rank)
    assert len(d["per_rank_fps"]) == 8 and d["per_rank_fps"][0] == pytest.approx(12.0) and d["per_rank_fps"][7] == pytest.approx(6 / 2.25)
    assert d["value"] == pytest.approx(48 / 2.25)
    _check_masks(d, 8)


def test_bench_launcher_does_not_touch_torch_in_the_parent_and_propagates_failure(vt):
    """the parent of a self-launch must not initialise the GPU (a HIP-initialised process may not be
    replaced or forked on this pool): bench.self_launch imports nothing heavy; a failing child
    (ranks that see fewer devices than --gpus) makes the parent exit non-zero with the reason on stderr"""
    import subprocess
    code = ("import sys; sys.argv=['bench.py']; import bench; "
            "assert 'torch' not in sys.modules and 'gstreamer_vit_tracker_amd' not in sys.modules; print('clean')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-2000:]
    if vt.device_count() >= 2:
        pytest.skip("two GPUs present: the 'needs 2 devices' refusal cannot be provoked")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny"], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr, r.stderr[-3000:]


def test_rank_cpu_masks_are_numa_local_and_disjoint(vt, tmp_path):
    """placement.rank_cpu_mask on a made-up 8-GPU, 2-socket topology (sysfs tree under tmp_path): ranks 0-3 share
    socket 0's CPUs, 4-7 socket 1's, every rank gets its own slice; restricted affinity and missing topology fall back
    to slicing what is allowed"""
    from gstreamer_vit_tracker_amd import placement as pl
    assert pl.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and pl.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    sysfs = tmp_path / "sys"
    # two CPU nodes (no SIMDs) then eight GPU nodes, as KFD lists them
    for i in range(10):
        nd = sysfs / "class/kfd/kfd/topology/nodes" / str(i)
        nd.mkdir(parents=True)
        gpu = i - 2
        (nd / "properties").write_text(f"cpu_cores_count {64 if i < 2 else 0}\nsimd_count {0 if i < 2 else 1024}\n"
                                       f"drm_render_minor {128 + gpu if i >= 2 else 0}\n")
        if i >= 2:
            dv = sysfs / f"class/drm/renderD{128 + gpu}/device"
            dv.mkdir(parents=True)
            (dv / "local_cpulist").write_text("0-63,128-191\n" if gpu < 4 else "64-127,192-255\n")
    gl = pl.gpu_local_cpulists(str(sysfs))
    assert len(gl) == 8 and gl[0] == pl.parse_cpulist("0-63,128-191") and gl[7] == pl.parse_cpulist("64-127,192-255")
    allowed = list(range(256))
    masks = [pl.rank_cpu_mask(r, 8, allowed, gl)[0] for r in range(8)]
    assert all(len(m) == 32 for m in masks)
    assert sorted(c for m in masks for c in m) == allowed                     # disjoint and complete
    assert all(set(masks[r]) <= set(gl[r]) for r in range(8))                 # NUMA-local
    # HIP_VISIBLE_DEVICES remaps local ranks to GPUs: rank 0 -> GPU 5 lives on socket 1
    m0, src = pl.rank_cpu_mask(0, 2, allowed, gl, devmap=[5, 1])
    assert set(m0) <= set(gl[5]) and "NUMA-local" in src
    assert pl.visible_device_map({"HIP_VISIBLE_DEVICES": "5,1"}) == [5, 1] and pl.visible_device_map({"ROCR_VISIBLE_DEVICES": "GPU-abc"}) is None
    # a cgroup that allows 16 CPUs of socket 0 only: rank on socket 1 has no local CPU -> everybody slices what is allowed
    small = list(range(16))
    ms = [pl.rank_cpu_mask(r, 8, small, gl)[0] for r in range(8)]
    assert sorted(c for m in ms for c in m) == small and all(len(m) == 2 for m in ms)
    # no topology at all
    ms = [pl.rank_cpu_mask(r, 2, list(range(8)), [])[0] for r in range(2)]
    assert ms == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert pl.rank_cpu_mask(0, 4, [3])[0] == [3]                               # fewer CPUs than ranks: shared
    with pytest.raises(ValueError):
        pl.rank_cpu_mask(2, 2, [0, 1])
    assert pl.apply(0, 1)["source"].startswith("single rank")                 # N = 1: mask untouched


def test_shard_streams_properties(vt):
    import gstreamer_vit_tracker_amd.distributed as vd
    for n in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            all_ = []
            for r in range(world):
                s = vd.shard_streams(n, r, world)
                all_ += s
                assert len(s) in (n // world, n // world + 1)
            assert sorted(all_) == list(range(n))


def test_plan_and_aggregate_single_process(vt):
    import gstreamer_vit_tracker_amd.distributed as vd
    p = vd.plan_rank(3, 8, 60, 64)
    assert p["clip_seed"] == 3 and len(p["phase"]) == 60 and p["phase"][0] == 0 and max(p["phase"]) < 64
    assert p["global_stream_ids"][0] == 180
    with pytest.raises(ValueError):
        vd.plan_rank(8, 8, 60, 64)
    a = vd.aggregate_throughput(600, 0.1)            # no process group: the local numbers
    assert a == {"frames": 600.0, "seconds": 0.1, "frames_per_s": 6000.0}


def test_rccl_entry_points_fail_cleanly_without_a_gpu(vt):
    """vt_broadcast_weights_rccl / vt_rccl_unique_id exist in the C ABI (a non-Python host's path to
    the start-up broadcast); without a GPU they return a status code."""
    import numpy as np
    if vt.device_count() > 0:
        pytest.skip("GPU present: covered by tests/test_gpu_host.py")
    with pytest.raises(vt.VtError) as e:
        vt.broadcast_weights_rccl(bytes(128), 1, 0, 0, "/nonexistent.vtw")
    assert e.value.code in (-2, -3)
    with pytest.raises(vt.VtError):
        vt.broadcast_weights_rccl(bytes(128), 2, 5, 0, None)       # rank out of range
