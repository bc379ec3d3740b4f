"""One process per GPU.  The only collective on the path is the one-off broadcast of the packed weight
arena from rank 0 (RCCL over xGMI when the backend is "nccl"); clips/windows are then sharded with no
collective in the step."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .packing import PackedWeights


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_weights(weights: Optional[PackedWeights], device: torch.device, src: int = 0) -> PackedWeights:
    """Rank ``src`` holds the packed arena; every other rank receives layout + bytes.  One collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert weights is not None
        return weights
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        assert weights is not None
        meta = [(weights.layout, int(weights.arena.numel()), weights.meta)]
    dist.broadcast_object_list(meta, src=src)
    layout, nbytes, m = meta[0]
    if rank != src:
        weights = PackedWeights.empty_like_layout(layout, nbytes, device, m)
    dist.broadcast(weights.arena, src=src)
    return weights


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin assignment of independent clips / windows to ranks."""
    return list(range(rank, n_items, world))
