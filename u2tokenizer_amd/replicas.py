"""Data-parallel replicas for the forward path: every volume is independent, so N GPUs = N
processes with full weights each, a round-robin split of the volume list and NO collective on the data path.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to line the ranks up for timing and to
reduce the elapsed time with MAX."""
from __future__ import annotations

import os
from typing import List, Optional

import torch


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Returns (dist module or None, rank, world). Reads RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return None, 0, 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if not dist.is_initialized():
        kw = {}
        if backend in (None, "nccl") and device is not None and device.type == "cuda":
            backend, kw = "nccl", {"device_id": device}
        dist.init_process_group(backend or "gloo", rank=rank, world_size=world, **kw)
    return dist, rank, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin: item i goes to rank i % world (mirrors split_dataset_by_node's contiguous-free split used by the
    reference's multi-GPU eval, green_score_accelerate/utils.py)."""
    return list(range(rank, n_items, world))


def barrier(dist, device: Optional[torch.device] = None) -> None:
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(dist, value: float, device: Optional[torch.device] = None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float, device: Optional[torch.device] = None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
