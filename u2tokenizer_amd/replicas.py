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
    if world == 1 and os.environ.get("U2_REPLICAS_FORCE_DIST", "0") != "1":
        return None, 0, 1   # (U2_REPLICAS_FORCE_DIST=1: a process group of one rank anyway -- runs the RCCL init / barrier /
        #                      reductions of the N > 1 launch on a single-GPU box: tests/test_gpu_replicas.py)
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


class StreamRoundRobin:
    """Issue independent forward calls on `n` HIP streams round-robin: with two batch-1 volumes in flight the small
    launches of one volume's tokenizer run under the large GEMMs of the other (96.6 vs 83.6 volumes/s on one MI355X).
    The modules keep one scratch workspace per stream, so nothing is shared between in-flight calls but the weights.

        rr = StreamRoundRobin(2)
        outs = [rr.submit(model.prepare_inputs_for_multimodal, ids, None, None, None, None, vol, qids) for vol in vols]
        rr.wait()
    """

    def __init__(self, n: int = 2, device: Optional[torch.device] = None):
        self.device = device
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n))]
        self.i = 0

    def submit(self, fn, *args, **kwargs):
        s = self.streams[self.i % len(self.streams)]
        self.i += 1
        s.wait_stream(torch.cuda.current_stream(self.device))  # inputs produced on the caller's stream are visible
        with torch.cuda.stream(s):
            return fn(*args, **kwargs)

    def wait(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)
