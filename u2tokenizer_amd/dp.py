"""Data-parallel training exchange for the path's replicas (training half): ZeRO-1 semantics of the
reference's DeepSpeed configuration (config/ds_config.json:27-41 -- stage 1, reduce_scatter: true, reduce_bucket_size 2e8,
allgather_bucket_size 2e8, overlap_comm: true, contiguous_gradients: true, gradient_accumulation_steps / gradient_clipping
"auto"; launched through accelerate, config/accelerate_config.yaml:3-6).

Every rank holds the full bf16 parameters and runs forward + backward on its own micro-batches (one process per GPU).
Layout per bucket (<= reduce_bucket_size elements, parameters in reverse registration order = the order the backward
produces gradients in):

  * `flat_grad`: ONE flat bf16 buffer; every parameter's `.grad` is a VIEW of it (DeepSpeed's contiguous_gradients), so
    autograd accumulates straight into the bucket -- no per-parameter copy, no second copy of the gradients, and gradient
    accumulation over micro-batches needs nothing extra;
  * after the LAST micro-batch of an optimiser step the bucket is REDUCE-SCATTERED (sum) over the process group -- RCCL over
    xGMI on the GPUs (backend "nccl"), gloo in the CPU tests -- from a post-accumulate-grad hook on a communication stream,
    i.e. under the rest of the backward (overlap_comm).  Buckets are launched strictly in index order on every rank, so
    the collective sequence cannot diverge between ranks whatever order their hooks fire in;
  * each rank applies AdamW to ITS piece of every bucket, on fp32 master weights + moments it alone keeps (12 bytes of
    state per parameter divided by the world size), in one fused HIP kernel per bucket on the GPU
    (u2tok_adamw_step: master, m, v, bf16 gradient piece -> bf16 parameter piece);
  * the updated bf16 pieces are ALL-GATHERED into one of two staging buffers and copied into the parameters on the
    communication stream, so the all-gather of bucket i runs under the AdamW of bucket i + 1.

Memory per rank: parameters + flat gradients (which REPLACE the per-parameter .grad tensors) + state / world + two
staging buckets -- against parameters + gradients + 2 flat copies before.

xGMI is point-to-point (7 links per GPU): a 2e8-element bf16 bucket is 400 MB, large enough that the collective runs at
link bandwidth rather than latency; nothing here assumes a switch.  No collective runs on the forward path.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def hf_param_groups(model: torch.nn.Module, weight_decay: float, lr: Optional[float] = None) -> List[dict]:
    """The parameter groups of the reference's optimiser (HF Trainer, optim="adamw_torch": `get_decay_parameter_names`):
    decoupled weight decay on everything except biases and LayerNorm / RMSNorm weights."""
    norm_types = (torch.nn.LayerNorm,)
    no_decay = set()
    for mn, mod in model.named_modules():
        for pn, _ in mod.named_parameters(recurse=False):
            full = f"{mn}.{pn}" if mn else pn
            if pn.endswith("bias") or isinstance(mod, norm_types) or "norm" in type(mod).__name__.lower():
                no_decay.add(full)
    decay = [p for n, p in model.named_parameters() if p.requires_grad and n not in no_decay]
    rest = [p for n, p in model.named_parameters() if p.requires_grad and n in no_decay]
    groups = [{"params": decay, "weight_decay": weight_decay}, {"params": rest, "weight_decay": 0.0}]
    if lr is not None:
        for g in groups:
            g["lr"] = lr
    return [g for g in groups if g["params"]]


class _Bucket:
    def __init__(self, params, offsets, groups, numel_padded, device, dtype, world, rank, multi=None):
        self.params, self.offsets = params, offsets          # offsets of the params inside the flat bucket
        self.numel = numel_padded                             # multiple of world
        self.piece = numel_padded // world
        self.flat_grad = torch.zeros(numel_padded, dtype=dtype, device=device)
        self.views = [self.flat_grad[o:o + p.numel()].view_as(p) for p, o in zip(params, offsets)]
        # this rank's piece of the reduced gradient; with one rank it is the bucket itself
        multi = world > 1 if multi is None else multi
        self.my_grad = torch.empty(self.piece, dtype=dtype, device=device) if multi else self.flat_grad
        # parameter-group index of every element of this rank's piece (padding: group 0, gradient 0)
        gidx = torch.zeros(numel_padded, dtype=torch.uint8)
        for p, o, g in zip(params, offsets, groups):
            gidx[o:o + p.numel()] = g
        self.group_idx = gidx[rank * self.piece:(rank + 1) * self.piece].to(device)
        # parameters whose hooks complete a micro-batch of this bucket.  All of them, unless a strict subset has been the set
        # that fires -- every micro-batch, nothing else -- for 3 consecutive steps (structurally unused parameters, e.g.
        # linear_aggregator.wv / dense of the reference, tta.py:47-48,62-65): then the bucket is launched from its hooks again.
        self.active = frozenset(range(len(params)))
        self.seen = (None, 0)         # (candidate subset, consecutive steps it was observed)
        self.redo = False             # a gradient arrived after an early launch from a learnt subset: reduce again in step()
        self.defer = False            # this step only step() launches the bucket
        self.count = [0] * len(params)  # hooks per parameter since the last step()
        self.got = set()              # parameters of `active` that fired in the current micro-batch
        self.ready = 0                # hooks fired in the current backward
        self.fired = 0                # hooks fired since the last step()
        self.micro = 0                # completed micro-batches since the last step()
        self.full = False             # all micro-batches of this step are in
        self.launched = False
        self.work = None              # async handle / completion event of the in-flight reduce-scatter


class Zero1AdamW:
    """AdamW with ZeRO-1 sharding of the optimiser state and bucketed reduce-scatter / all-gather of gradients /
    parameters.  Use like an optimiser (parameters or torch.optim-style parameter groups with their own lr / weight_decay):

        opt = Zero1AdamW(hf_param_groups(model, 0.0), lr=4e-6, gradient_accumulation_steps=4, max_grad_norm=1.0)
        for micro in range(4): loss(micro).backward()
        opt.step(); opt.zero_grad()

    The gradients live in the optimiser's flat buckets (`p.grad` is a view); step() zeroes them, so any zero_grad() flavour
    of the caller -- including `set_to_none=True`, after which autograd creates fresh tensors that the hooks fold back into
    the buckets -- is harmless.
    """

    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group=None, reduce_bucket_size: int = int(2e8), allgather_bucket_size: int = int(2e8),
                 overlap_comm: bool = True, max_grad_norm: Optional[float] = None, gradient_accumulation_steps: int = 1,
                 force_collectives: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # `force_collectives`: run the reduce-scatter / all-reduce / all-gather sequence even in a world of ONE rank (they are
        # copies then) -- the way to execute the RCCL code path, its streams and events on a single-GPU box
        # (tests/test_gpu_backward.py); needs an initialised process group
        self._multi = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self.betas, self.eps = betas, eps
        # global gradient-norm clipping ("gradient_clipping": "auto" of config/ds_config.json:41 = the trainer's max_grad_norm;
        # torch.nn.utils.clip_grad_norm_ semantics on the MEAN gradient): every rank owns a piece of every bucket, so the
        # squared norm is the sum of the pieces' squares, one scalar all-reduce per step
        self.max_grad_norm = max_grad_norm
        self._last_norm = None
        self.t = 0
        self.accum = max(1, int(gradient_accumulation_steps))
        self.overlap = overlap_comm and self._multi
        # The per-step agreement on which parameters fired (_agree_on_subsets) is host-side bookkeeping: a few hundred flags that must be
        # READ by the host before the step can go on.  On the device group that is an all-reduce followed by a blocking device -> host
        # copy in front of the clip-norm and AdamW launches, every step.  It runs on HOST memory instead: in a world of one
        # rank there is nothing to reduce; with several ranks on GPUs a gloo twin of the group carries it (created here, collectively --
        # every rank constructs its optimiser), so the device queue is never drained for it.
        self._flag_group = self.group
        self._flag_device = None          # None: host tensor
        if self.world > 1 and dist.is_initialized() and dist.get_backend(process_group) != "gloo":
            try:
                ranks = dist.get_process_group_ranks(process_group if process_group is not None else dist.group.WORLD)
                self._flag_group = dist.new_group(ranks=ranks, backend="gloo")
            except Exception:             # no gloo in this build: the device group it is (one small blocking reduction per step)
                self._flag_group, self._flag_device = self.group, "param"
        plist = list(params)
        if plist and isinstance(plist[0], dict):
            self.param_groups = [{"lr": g.get("lr", lr), "weight_decay": g.get("weight_decay", weight_decay)} for g in plist]
            tagged = [(p, gi) for gi, g in enumerate(plist) for p in g["params"] if p.requires_grad]
        else:
            self.param_groups = [{"lr": lr, "weight_decay": weight_decay}]
            tagged = [(p, 0) for p in plist if p.requires_grad]
        if not tagged:
            raise ValueError("no trainable parameters")
        if len(self.param_groups) > 8:
            raise ValueError("at most 8 parameter groups")
        # parameters are bucketed in REVERSE registration order: the backward produces gradients roughly last layer first,
        # so the first bucket to fill is the first one whose reduce-scatter can start
        tagged = tagged[::-1]
        bsize = max(1, min(reduce_bucket_size, allgather_bucket_size))
        self.buckets: List[_Bucket] = []
        cur, offs, grp, n = [], [], [], 0

        def close():
            nonlocal cur, offs, grp, n
            if cur:
                pad = (-n) % self.world
                self.buckets.append(_Bucket(cur, offs, grp, n + pad, cur[0].device, cur[0].dtype, self.world, self.rank, self._multi))
            cur, offs, grp, n = [], [], [], 0

        for p, gi in tagged:
            if cur and (n + p.numel() > bsize or p.device != cur[0].device or p.dtype != cur[0].dtype):
                close()
            cur.append(p)
            offs.append(n)
            grp.append(gi)
            n += p.numel()
        close()
        # sharded state: fp32 master copy + moments of this rank's piece of every bucket
        self.state = []
        for b in self.buckets:
            flat = torch.zeros(b.numel, dtype=torch.float32, device=b.flat_grad.device)
            for p, o in zip(b.params, b.offsets):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            mine = flat[self.rank * b.piece:(self.rank + 1) * b.piece]
            self.state.append(dict(master=mine.clone(), m=torch.zeros_like(mine), v=torch.zeros_like(mine)))
            del flat
        # two staging buffers for the all-gathered bf16 parameters of a bucket (bucket i + 1 is updated while i is gathered)
        # (buckets are split by dtype / device: one pair of staging buffers per kind, or an fp32 parameter behind a bf16 bucket
        # would be rounded through bf16 on every step)
        self._stages = {}
        for b in self.buckets:
            key = (b.flat_grad.dtype, b.flat_grad.device)
            self._stages[key] = max(self._stages.get(key, 0), b.numel)
        self._stages = {k: [torch.empty(n, dtype=k[0], device=k[1]) for _ in range(2 if self._multi else 1)]
                        for k, n in self._stages.items()}
        self._comm_stream = None
        self._use_rs = True          # dist.reduce_scatter_tensor; falls back to all_reduce + slice where unsupported (gloo)
        self._next_launch = 0        # buckets [0, _next_launch) have their reduce-scatter in flight / done for this step
        self._install_grad_views()
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, pi)))

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):  # a scheduler setting one rate for every group
        for g in self.param_groups:
            g["lr"] = value

    @property
    def last_grad_norm(self) -> Optional[float]:
        """Global norm of the mean gradient of the last step (before clipping); converting it synchronises with the device."""
        return None if self._last_norm is None else float(self._last_norm)

    # ------------------------------------------------------------------ gradient exchange
    def _install_grad_views(self):
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                p.grad = v

    def _make_hook(self, bi, pi):
        def hook(p):
            b = self.buckets[bi]
            view = b.views[pi]
            if p.grad is not view:
                # the caller dropped the view (zero_grad(set_to_none=True)): autograd made a fresh tensor holding this
                # backward's gradient; fold it into the bucket and hand the view back
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.add_(p.grad)
                p.grad = view
            b.fired += 1
            b.count[pi] += 1
            if pi not in b.active:
                # a parameter that used to get no gradient got one in this backward (a data-dependent branch, e.g. a
                # text-only batch before).  `active` itself is COLLECTIVE state -- it only changes in step(), from values every
                # rank has seen (below) -- so this rank just stops trusting the subset for the rest of this step.  Not launched
                # yet: the bucket waits for step().  Launched already: the reduced piece lacks this gradient -- the bucket
                # buffer still holds every gradient (the early launch of a learnt subset never reduces in place), so step()
                # reduces it again, after all ranks agreed on which buckets need it.
                if b.launched and self._multi:
                    b.redo = True
                else:
                    b.full, b.defer = False, True
                return
            if b.defer:
                return
            b.got.add(pi)
            if len(b.got) == len(b.active):
                b.got.clear()
                b.micro += 1
                if b.micro == self.accum:     # the last micro-batch of this optimiser step: the bucket can go
                    b.full = True
                    if self.overlap:
                        self._launch_in_order()
        return hook

    def _launch_in_order(self, flush: bool = False):
        """Launch the reduce-scatter of every bucket that is complete, strictly in bucket order (bucket i waits for buckets
        < i): all ranks issue the same collective sequence even when their hooks fire in different orders.  A bucket without
        any used parameter (empty learnt subset) rides along with the next one.  flush: step() launches whatever is left."""
        while self._next_launch < len(self.buckets):
            b = self.buckets[self._next_launch]
            if not (flush or b.full or (not b.active and not b.defer)):
                break
            self._launch_reduce(b)
            b.launched = True
            self._next_launch += 1

    def _launch_reduce(self, b: _Bucket):
        if not self._multi:
            return                                # my_grad IS flat_grad
        if b.work is not None:                    # (defensive: never leave a handle un-waited)
            self._finish_reduce(b)
        if b.flat_grad.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=b.flat_grad.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(b.flat_grad.device))
            with torch.cuda.stream(self._comm_stream):
                self._reduce(b, async_op=False)
                b.work = torch.cuda.Event()
                b.work.record(self._comm_stream)
        else:
            b.work = self._reduce(b, async_op=True)

    def _reduce(self, b: _Bucket, async_op: bool):
        if self._use_rs:
            try:
                return dist.reduce_scatter_tensor(b.my_grad, b.flat_grad, op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=async_op)
            except (RuntimeError, NotImplementedError):
                self._use_rs = False  # e.g. gloo: no reduce_scatter -- same result through all_reduce + this rank's slice
        # (an early launch from a LEARNT count must leave the bucket intact: a late gradient makes step() reduce it again)
        buf = b.flat_grad.clone() if len(b.active) < len(b.params) and not b.redo else b.flat_grad
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if w is None:
            b.my_grad.copy_(buf[self.rank * b.piece:(self.rank + 1) * b.piece])
        return ("allreduce", w, buf)

    def _finish_reduce(self, b: _Bucket):
        if isinstance(b.work, torch.cuda.Event):   # the compute stream waits for THIS bucket's reduce-scatter only
            torch.cuda.current_stream(b.flat_grad.device).wait_event(b.work)
        elif isinstance(b.work, tuple):
            if b.work[1] is not None:
                b.work[1].wait()
            b.my_grad.copy_(b.work[2][self.rank * b.piece:(self.rank + 1) * b.piece])
        elif b.work is not None:
            b.work.wait()
        b.work = None

    def _agree_on_subsets(self):
        """Which parameters of a bucket fire (`active`, what lets a bucket with structurally unused parameters -- e.g.
        linear_aggregator.wv / dense of the reference, tta.py:47-48,62-65 -- leave from its hooks) and which early launches were
        premature (`redo`) are decided from ONE small MAX-reduction per step that every rank always takes part in: per parameter
        "fired on some rank", per bucket "some rank saw a late gradient" and "some rank's hook counts were irregular".  `active`
        therefore is the same set on every rank at every step, whatever each rank's data did (a rank-local decision
        gated this very collective): the union of what fired becomes the subset after it has been the same, with regular counts
        and no repair, for 3 consecutive steps; the moment anything outside it fires anywhere the bucket goes back to all
        parameters; a premature launch is reduced again here, in bucket order, on every rank."""
        np_ = [len(b.params) for b in self.buckets]
        flags = []
        for b in self.buckets:
            flags += [1.0 if c else 0.0 for c in b.count]
        for b in self.buckets:
            irregular = not all(c in (0, self.accum) for c in b.count) or b.defer
            flags += [1.0 if b.redo else 0.0, 1.0 if irregular else 0.0]
        if self.world > 1:
            dev = self.buckets[0].flat_grad.device if self._flag_device == "param" else "cpu"
            t = torch.tensor(flags, dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._flag_group)
            vals = t.tolist()
        else:
            vals = flags                  # one rank: its own flags are the agreement (no device work, no synchronisation)
        pos, tail = 0, sum(np_)
        for bi, b in enumerate(self.buckets):
            fired = frozenset(i for i in range(np_[bi]) if vals[pos + i])
            pos += np_[bi]
            redo, irregular = bool(vals[tail + 2 * bi]), bool(vals[tail + 2 * bi + 1])
            if redo:                      # same reduces, same order, on every rank
                b.redo = True             # (tells _reduce that the bucket may be reduced in place now)
                self._finish_reduce(b)
                self._launch_reduce(b)
            full = frozenset(range(np_[bi]))
            if not fired <= b.active:     # a stranger fired somewhere: back to all parameters, on every rank
                b.active, b.seen = full, (None, 0)
            elif redo or irregular or fired == b.active:
                if fired != b.active:
                    b.seen = (None, 0)
            else:                         # a strict subset of the current set, cleanly: a candidate
                b.seen = (fired, b.seen[1] + 1) if b.seen[0] == fired else (fired, 1)
                if b.seen[1] >= 3:
                    b.active, b.seen = fired, (None, 0)

    # ------------------------------------------------------------------ step
    def _update_piece(self, b: _Bucket, st: dict, out: torch.Tensor, coef: Optional[torch.Tensor]):
        """AdamW on this rank's piece: st (fp32 master / m / v) in place, `out` <- bf16 (parameter dtype) of the new master."""
        lrs = [g["lr"] for g in self.param_groups]
        wds = [g["weight_decay"] for g in self.param_groups]
        if b.my_grad.is_cuda and b.my_grad.dtype == torch.bfloat16:
            from . import ops  # the fused HIP kernel; a GPU run without the library fails loudly here
            ops.adamw_step(st["master"], st["m"], st["v"], b.my_grad[:b.piece], out, self.t, lrs, wds, self.betas, self.eps,
                           grad_scale=1.0 / self.world, grad_coef=coef, group=b.group_idx if len(lrs) > 1 else None)
            return
        b1, b2 = self.betas
        c1, c2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        g = b.my_grad[:b.piece].float() / self.world
        if coef is not None:
            g = g * coef
        if len(lrs) > 1:
            gi = b.group_idx.long()
            lr_e = torch.tensor(lrs, dtype=torch.float32, device=g.device)[gi]
            wd_e = torch.tensor(wds, dtype=torch.float32, device=g.device)[gi]
        else:
            lr_e, wd_e = lrs[0], wds[0]
        st["master"].mul_(1 - lr_e * wd_e)                              # decoupled weight decay (AdamW)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st["v"].sqrt() / (c2 ** 0.5)).add_(self.eps)
        st["master"].sub_(lr_e / c1 * (st["m"] / denom))
        out.copy_(st["master"])

    @torch.no_grad()
    def step(self):
        self.t += 1
        # whatever is not in flight yet goes now, in order: no overlap requested, buckets with parameters that received no
        # gradient, buckets this rank deferred.
        self._launch_in_order(flush=True)
        if self.overlap:
            self._agree_on_subsets()
        coef = None
        if self.max_grad_norm is not None:
            for b in self.buckets:
                self._finish_reduce(b)
            # (one reduction kernel per bucket straight from the bf16 piece: no fp32 temporaries)
            sq = torch.stack([torch.linalg.vector_norm(b.my_grad[:b.piece], 2, dtype=torch.float32).pow(2)
                              for b in self.buckets]).sum() / (self.world * self.world)
            if self._multi:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            total = sq.sqrt()
            self._last_norm = total
            coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0).reshape(1).float()
        cuda = self.buckets[0].flat_grad.is_cuda
        dev = self.buckets[0].flat_grad.device
        if cuda and self._multi and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        nst = 2 if self._multi else 1
        stage_free = {k: [None] * nst for k in self._stages}
        for i, (b, st) in enumerate(zip(self.buckets, self.state)):
            self._finish_reduce(b)
            skey = (b.flat_grad.dtype, b.flat_grad.device)
            stage = self._stages[skey][i % nst]
            if not self._multi:
                self._update_piece(b, st, stage[:b.numel], coef)
                self._copy_out(b, stage)
            elif cuda:
                cur = torch.cuda.current_stream(dev)
                si = i % nst
                if stage_free[skey][si] is not None:
                    cur.wait_event(stage_free[skey][si])    # the last user of this staging buffer has been copied out
                mine = stage[self.rank * b.piece:(self.rank + 1) * b.piece]
                self._update_piece(b, st, mine, coef)
                updated = torch.cuda.Event()
                updated.record(cur)
                with torch.cuda.stream(self._comm_stream):   # all-gather + copy-out of bucket i under the AdamW of i + 1
                    self._comm_stream.wait_event(updated)
                    dist.all_gather_into_tensor(stage[:b.numel], mine, group=self.group)
                    self._copy_out(b, stage)
                    stage_free[skey][si] = torch.cuda.Event()
                    stage_free[skey][si].record(self._comm_stream)
            else:
                mine = stage[self.rank * b.piece:(self.rank + 1) * b.piece]
                self._update_piece(b, st, mine, coef)
                dist.all_gather_into_tensor(stage[:b.numel], mine.clone(), group=self.group)
                self._copy_out(b, stage)
        if cuda and self._multi:
            torch.cuda.current_stream(dev).wait_stream(self._comm_stream)
        # the step owns the gradients: zero the buckets and re-arm
        for b in self.buckets:
            b.flat_grad.zero_()
            b.ready = b.fired = b.micro = 0
            b.full = b.launched = b.redo = b.defer = False
            b.count = [0] * len(b.params)
            b.got.clear()
        self._next_launch = 0
        self._install_grad_views()

    @staticmethod
    def _copy_out(b, stage) -> None:
        """The gathered bf16 bucket -> the parameters it holds: one multi-tensor copy instead of a launch per parameter (the
        u2Qwen3-8B-shaped model has ~700 of them)."""
        srcs = [stage[o:o + p.numel()].view_as(p) for p, o in zip(b.params, b.offsets)]
        dsts = [p.data for p in b.params]
        if hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_(dsts, srcs)
        else:
            for d, s_ in zip(dsts, srcs):
                d.copy_(s_)

    def zero_grad(self, set_to_none: bool = False):
        """The buckets are zeroed by step(); this only matters for a step that is abandoned half-way."""
        for b in self.buckets:
            if b.work is not None:
                self._finish_reduce(b)
            b.flat_grad.zero_()
            b.ready = b.fired = b.micro = 0
            b.full = b.launched = b.redo = b.defer = False
            b.count = [0] * len(b.params)
            b.got.clear()
        self._next_launch = 0
        self._install_grad_views()

    # ------------------------------------------------------------------ checkpoint / resume of this rank's shard
    def state_dict(self) -> dict:
        """This rank's shard of the optimiser state (fp32 master pieces + moments, step count, the groups' lr / weight decay)
        -- what DeepSpeed writes per rank as zero_pp_rank_*_optim_states.  Loading requires the same parameter order, bucket
        sizes and world size."""
        return {"step": self.t, "world": self.world, "rank": self.rank,
                "layout": [(b.numel, len(b.params)) for b in self.buckets],
                "param_groups": [dict(g) for g in self.param_groups],
                "buckets": [{k: v.detach().cpu().clone() for k, v in st.items()} for st in self.state]}

    def load_state_dict(self, sd: dict) -> None:
        layout = [(b.numel, len(b.params)) for b in self.buckets]
        if sd["world"] != self.world or sd["rank"] != self.rank or [tuple(x) for x in sd["layout"]] != layout:
            raise ValueError("Zero1AdamW.load_state_dict: the shard was written for another world size / rank / bucket layout")
        self.t = int(sd["step"])
        if "param_groups" in sd and len(sd["param_groups"]) == len(self.param_groups):
            self.param_groups = [dict(g) for g in sd["param_groups"]]
        with torch.no_grad():
            for b, st, src in zip(self.buckets, self.state, sd["buckets"]):
                for k in ("master", "m", "v"):
                    st[k].copy_(src[k])
                # the bf16 parameters follow the restored master weights (this rank's piece; the others arrive by all-gather)
                stage = self._stages[(b.flat_grad.dtype, b.flat_grad.device)][0]
                mine = stage[self.rank * b.piece:(self.rank + 1) * b.piece]
                mine.copy_(st["master"])
                if self._multi:
                    dist.all_gather_into_tensor(stage[:b.numel], mine.clone(), group=self.group)
                self._copy_out(b, stage)

    def state_bytes_per_rank(self) -> int:
        return sum(s["master"].numel() * 12 for s in self.state)
