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


def shard_streams(n_streams: int, rank: int, world: int) -> list[int]:
    """Global stream ids owned by `rank` (round-robin; no data moves between ranks)."""
    return list(range(rank, n_streams, world))


def aggregate_max_time(seconds: float, device="cpu") -> float:
    """MAX over ranks of a local wall time (bench contract: whole-job time = slowest rank)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
