"""Data-parallel training exchange for the path's replicas (training half): ZeRO-1 semantics of the
reference's DeepSpeed configuration (config/ds_config.json:27-39 -- stage 1, reduce_scatter: true, reduce_bucket_size 2e8,
allgather_bucket_size 2e8, overlap_comm: true; launched through accelerate, config/accelerate_config.yaml:3-6).

Every rank holds the full bf16 parameters and runs forward + backward on its own micro-batch (one process per GPU).
Once per optimiser step:

  1. gradients are packed into flat buckets of <= reduce_bucket_size elements and REDUCE-SCATTERED (mean) over the
     process group -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.  With overlap_comm the
     reduce-scatter of a bucket is launched from a post-accumulate-grad hook as soon as its last gradient exists, on a
     communication stream, while the rest of the backward is still running;
  2. each rank applies AdamW to ITS piece of every bucket, on fp32 master weights + moments it alone keeps (the
     optimiser state is sharded world_size ways: for the 8B build that is the difference between 16 and 2 bytes of
     state per parameter per GPU);
  3. the updated bf16 pieces are ALL-GATHERED bucket by bucket (<= allgather_bucket_size elements) back into the
     parameters.

xGMI is point-to-point (7 links per GPU): a 2e8-element bf16 bucket is 400 MB, large enough that the collective runs at
link bandwidth rather than latency; nothing here assumes a switch.  No collective runs on the forward path.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, offsets, numel_padded, device, dtype, world):
        self.params, self.offsets = params, offsets          # offsets of the params inside the flat bucket
        self.numel = numel_padded                             # multiple of world
        self.piece = numel_padded // world
        self.flat_grad = torch.zeros(numel_padded, dtype=dtype, device=device)
        self.flat_param = torch.empty(numel_padded, dtype=dtype, device=device)
        self.my_grad = torch.empty(self.piece, dtype=dtype, device=device)
        self.ready = 0
        self.work = None                                      # async handle / event of the in-flight reduce-scatter


class Zero1AdamW:
    """AdamW with ZeRO-1 sharding of the optimiser state and bucketed reduce-scatter / all-gather of gradients /
    parameters.  Use like an optimiser:

        opt = Zero1AdamW(model.parameters(), lr=4e-6)
        loss.backward(); opt.step(); opt.zero_grad()
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group=None, reduce_bucket_size: int = int(2e8), allgather_bucket_size: int = int(2e8),
                 overlap_comm: bool = True, max_grad_norm: Optional[float] = None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # global gradient-norm clipping ("gradient_clipping": "auto" of config/ds_config.json:41 = the trainer's max_grad_norm;
        # torch.nn.utils.clip_grad_norm_ semantics on the MEAN gradient): every rank owns a piece of every bucket, so the
        # squared norm is the sum of the pieces' squares, one scalar all-reduce per step
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm: Optional[float] = None
        self.t = 0
        self.overlap = overlap_comm and self.world > 1
        # parameters are bucketed in REVERSE registration order: the backward produces gradients roughly last layer first,
        # so the first bucket to fill is the first one whose reduce-scatter can start
        ps = [p for p in params if p.requires_grad][::-1]
        if not ps:
            raise ValueError("no trainable parameters")
        bsize = max(1, min(reduce_bucket_size, allgather_bucket_size))
        self.buckets: List[_Bucket] = []
        cur, offs, n = [], [], 0

        def close():
            nonlocal cur, offs, n
            if cur:
                pad = (-n) % self.world
                self.buckets.append(_Bucket(cur, offs, n + pad, cur[0].device, cur[0].dtype, self.world))
            cur, offs, n = [], [], 0

        for p in ps:
            if cur and (n + p.numel() > bsize or p.device != cur[0].device or p.dtype != cur[0].dtype):
                close()
            cur.append(p)
            offs.append(n)
            n += p.numel()
        close()
        # sharded state: fp32 master copy + moments of this rank's piece of every bucket
        self.state = []
        for b in self.buckets:
            for p, o in zip(b.params, b.offsets):
                b.flat_param[o:o + p.numel()].copy_(p.detach().reshape(-1))
            mine = b.flat_param[self.rank * b.piece:(self.rank + 1) * b.piece]
            self.state.append(dict(master=mine.float().clone(), m=torch.zeros_like(mine, dtype=torch.float32),
                                   v=torch.zeros_like(mine, dtype=torch.float32)))
        self._comm_stream = None
        self._use_rs = True          # dist.reduce_scatter_tensor; falls back to all_reduce + slice where unsupported (gloo)
        self._hooks = []
        if self.overlap:
            for bi, b in enumerate(self.buckets):
                for pi, p in enumerate(b.params):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, pi)))

    # ------------------------------------------------------------------ gradient exchange
    def _make_hook(self, bi, pi):
        def hook(p):
            b = self.buckets[bi]
            o = b.offsets[pi]
            b.flat_grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
            b.ready += 1
            if b.ready == len(b.params):
                self._launch_reduce(b)
        return hook

    def _launch_reduce(self, b: _Bucket):
        if self.world == 1:
            b.my_grad.copy_(b.flat_grad[:b.piece])
            return
        if b.flat_grad.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=b.flat_grad.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(b.flat_grad.device))
            with torch.cuda.stream(self._comm_stream):
                self._reduce(b, async_op=False)
            b.work = "stream"
        else:
            b.work = self._reduce(b, async_op=True)

    def _reduce(self, b: _Bucket, async_op: bool):
        if self._use_rs:
            try:
                return dist.reduce_scatter_tensor(b.my_grad, b.flat_grad, op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=async_op)
            except (RuntimeError, NotImplementedError):
                self._use_rs = False  # e.g. gloo: no reduce_scatter -- same result through all_reduce + this rank's slice
        w = dist.all_reduce(b.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if w is None:
            b.my_grad.copy_(b.flat_grad[self.rank * b.piece:(self.rank + 1) * b.piece])
        return ("allreduce", w)

    def _finish_reduce(self, b: _Bucket):
        if b.work == "stream":
            torch.cuda.current_stream(b.flat_grad.device).wait_stream(self._comm_stream)
        elif isinstance(b.work, tuple):
            b.work[1].wait()
            b.my_grad.copy_(b.flat_grad[self.rank * b.piece:(self.rank + 1) * b.piece])
        elif b.work is not None:
            b.work.wait()
        b.work = None

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        c1, c2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        grads = []
        for b in self.buckets:
            if b.ready != len(b.params):          # no overlap (or a parameter without a hook firing): pack + reduce now
                for p, o in zip(b.params, b.offsets):
                    if p.grad is not None:
                        b.flat_grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
                    else:
                        b.flat_grad[o:o + p.numel()].zero_()
                self._launch_reduce(b)
            self._finish_reduce(b)
            grads.append(b.my_grad.float() / self.world)                  # mean over ranks, this rank's piece
        if self.max_grad_norm is not None:
            sq = torch.stack([g.pow(2).sum() for g in grads]).sum()
            if self.world > 1:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            total = sq.sqrt()
            self.last_grad_norm = float(total)
            coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
            grads = [g * coef for g in grads]
        for b, st, g in zip(self.buckets, self.state, grads):
            if self.wd:
                st["master"].mul_(1 - self.lr * self.wd)                  # decoupled weight decay (AdamW)
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st["v"] / c2).sqrt_().add_(self.eps)
            st["master"].addcdiv_(st["m"] / c1, denom, value=-self.lr)
            mine = b.flat_param[self.rank * b.piece:(self.rank + 1) * b.piece]
            mine.copy_(st["master"])
            if self.world > 1:
                dist.all_gather_into_tensor(b.flat_param, mine.clone(), group=self.group)
            for p, o in zip(b.params, b.offsets):
                p.copy_(b.flat_param[o:o + p.numel()].view_as(p))
            b.ready = 0

    def zero_grad(self, set_to_none: bool = True):
        for b in self.buckets:
            for p in b.params:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    # ------------------------------------------------------------------ checkpoint / resume of this rank's shard
    def state_dict(self) -> dict:
        """This rank's shard of the optimiser state (fp32 master pieces + moments, step count) -- what DeepSpeed writes per
        rank as zero_pp_rank_*_optim_states.  Loading requires the same parameter order, bucket sizes and world size."""
        return {"step": self.t, "world": self.world, "rank": self.rank,
                "layout": [(b.numel, len(b.params)) for b in self.buckets],
                "buckets": [{k: v.detach().cpu().clone() for k, v in st.items()} for st in self.state]}

    def load_state_dict(self, sd: dict) -> None:
        layout = [(b.numel, len(b.params)) for b in self.buckets]
        if sd["world"] != self.world or sd["rank"] != self.rank or [tuple(x) for x in sd["layout"]] != layout:
            raise ValueError("Zero1AdamW.load_state_dict: the shard was written for another world size / rank / bucket layout")
        self.t = int(sd["step"])
        with torch.no_grad():
            for b, st, src in zip(self.buckets, self.state, sd["buckets"]):
                for k in ("master", "m", "v"):
                    st[k].copy_(src[k])
                # the bf16 parameters follow the restored master weights (this rank's piece; the others arrive by all-gather)
                mine = b.flat_param[self.rank * b.piece:(self.rank + 1) * b.piece]
                mine.copy_(st["master"])
                if self.world > 1:
                    dist.all_gather_into_tensor(b.flat_param, mine.clone(), group=self.group)
                for p, o in zip(b.params, b.offsets):
                    p.copy_(b.flat_param[o:o + p.numel()].view_as(p))

    def state_bytes_per_rank(self) -> int:
        return sum(s["master"].numel() * 12 for s in self.state)
