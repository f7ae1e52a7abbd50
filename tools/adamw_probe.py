#!/usr/bin/env python
"""The ZeRO-1 shard update on one GPU: u2tok_adamw_step on one 2e8-element bucket (bytes moved / time), then a whole
Zero1AdamW.step() (clip norm + update + copy-out + zeroing, world 1) over a synthetic 2e9-parameter model of 200 tensors.

    python tools/adamw_probe.py
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import dp, ops  # noqa: E402

dev = torch.device("cuda", 0)
ops.device_check()
n = 200_000_000
master, m, v = (torch.randn(n, device=dev) for _ in range(3))
v.abs_()
grad = (torch.randn(n, device=dev) * 0.01).to(torch.bfloat16)
out = torch.empty(n, dtype=torch.bfloat16, device=dev)
group = torch.zeros(n, dtype=torch.uint8, device=dev)
for g_, label in ((None, "no group table"), (group, "with group table")):
    for _ in range(2):
        ops.adamw_step(master, m, v, grad, out, 1, 1e-4, 0.01, group=g_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        ops.adamw_step(master, m, v, grad, out, 2 + i, 1e-4, 0.01, group=g_)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    by = n * (28 + (1 if g_ is not None else 0))
    print(f"u2tok_adamw_step, {n:.1e} elements, {label}: {dt * 1e3:.2f} ms, {by / dt / 1e12:.2f} TB/s")
del master, m, v, grad, out, group
params = [torch.nn.Parameter(torch.randn(10_000_000, device=dev).to(torch.bfloat16)) for _ in range(200)]
opt = dp.Zero1AdamW(params, lr=1e-4, max_grad_norm=1.0)
ts = []
for i in range(4):
    for b in opt.buckets:
        b.flat_grad.normal_(0, 0.01)
        b.fired, b.micro = len(b.params), 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"Zero1AdamW.step(), world 1, 2.0e9 parameters in 200 tensors: {min(ts[1:]):.1f} ms "
      f"({2e9 * 36 / min(ts[1:]) / 1e9:.2f} TB/s of 36 B per parameter: norm 2 + update 28 + copy-out 4 + zero 2)")
