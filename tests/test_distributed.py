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
