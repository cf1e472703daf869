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
        q.put((rank, digest, t.numel(), shard, agg))
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
