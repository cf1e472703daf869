"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in CPU tests). The tracker path shards by stream — one tracked stream is one
sequential frame chain (each crop depends on the previous box) and streams never exchange data —
so there is exactly one collective: broadcasting the weight blob once at start-up.
PyTorch is used for device memory and the collective only; all per-frame work is in
libvittrack_hip.so.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def broadcast_weights(blob_path: str | None, device="cuda", src: int = 0) -> torch.Tensor:
    """Rank `src` reads the blob from disk; every rank returns a uint8 tensor on `device` holding
    it. Two broadcasts: the size (8 B), then the bytes in one message (185 MB for ViT-B: a single
    large transfer suits xGMI's per-link bandwidth better than many small ones)."""
    rank = dist.get_rank()
    if rank == src:
        raw = np.fromfile(blob_path, dtype=np.uint8)
        n = torch.tensor([raw.size], dtype=torch.int64, device=device)
    else:
        raw = None
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.from_numpy(raw).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    return buf


def timed_broadcast_weights(blob_path: str | None, device="cuda", src: int = 0):
    """broadcast_weights plus the record bench.py prints under `collective`: which backend moved the
    bytes, how many ranks the process group really has (dist.get_world_size(), not an argument), the
    message size and the wall time of the two broadcasts on this rank (device work synchronised)."""
    import time
    is_cuda = str(device).startswith("cuda")
    if is_cuda:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    buf = broadcast_weights(blob_path, device=device, src=src)
    if is_cuda:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
           "broadcast_bytes": int(buf.numel()), "broadcast_ms": aggregate_max_time(ms, device=device),
           "collectives_per_frame": 0}
    return buf, rec


def gather_per_rank(local_value: float, device="cpu") -> list:
    """[value of rank 0, value of rank 1, ...] on every rank (one all_gather, outside the timed region);
    without a process group the one local value."""
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([float(local_value)], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]
    return [float(local_value)]


def gather_objects(obj) -> list:
    """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable records: CPU masks, plans)"""
    if dist.is_available() and dist.is_initialized():
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, obj)
        return parts
    return [obj]


def shard_streams(n_streams: int, rank: int, world: int) -> list[int]:
    """Global stream ids owned by `rank` (round-robin; no data moves between ranks)."""
    return list(range(rank, n_streams, world))


def aggregate_max_time(seconds: float, device="cpu") -> float:
    """MAX over ranks of a local wall time (bench contract: whole-job time = slowest rank)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- what bench.py does per rank (kept here so the world-size-2 gloo test covers it) -------------

def plan_rank(rank: int, world: int, streams_per_gpu: int, ring: int) -> dict:
    """Work of one rank of the weak-scaling bench: every rank tracks `streams_per_gpu` independent
    streams of its OWN synthetic clip (seed = rank, so no two ranks see the same pixels); stream i
    of a rank starts `phase[i]` frames into the clip's closed path. Nothing here depends on another
    rank's data: ranks exchange only the start-up weight blob and, at the end, their wall times."""
    if not (0 <= rank < world) or streams_per_gpu < 1 or ring < 1:
        raise ValueError("bad rank / world / streams / ring")
    # consecutive clip positions: the frames one engine needs at a step are then ONE contiguous range of
    # the host's frame ring (two at its wrap), which is what lets the full-frame upload leg move an
    # engine's frames with one copy; distinct positions while streams_per_gpu <= ring
    return {"clip_seed": rank, "phase": [i % ring for i in range(streams_per_gpu)],
            "global_stream_ids": [rank * streams_per_gpu + i for i in range(streams_per_gpu)]}


def aggregate_throughput(local_frames: int, local_seconds: float, device="cpu") -> dict:
    """Whole-job throughput as the bench contract defines it: frames of ALL ranks / MAX over ranks of
    the wall time of the timed region. One all_reduce(SUM) and one all_reduce(MAX), outside the
    timed region."""
    if dist.is_available() and dist.is_initialized():
        f = torch.tensor([float(local_frames)], dtype=torch.float64, device=device)
        t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        frames, seconds = float(f.item()), float(t.item())
    else:
        frames, seconds = float(local_frames), float(local_seconds)
    return {"frames": frames, "seconds": seconds, "frames_per_s": frames / seconds}
