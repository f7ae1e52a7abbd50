#!/usr/bin/env python
"""What the 8 cls rows behind the 16384 patch rows cost each ViT product (the in-launch tail of the big-tile kernels, rows16.h):
microseconds per launch at M = 16384 and at M = 16392, operand sets in rotation.

    python tools/tail_probe.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, sys.argv[sys.argv.index("--root") + 1] if "--root" in sys.argv else str(Path(__file__).resolve().parents[1]))   # (--root DIR: an A/B build, tools/mk_ab_build.sh)
from u2tokenizer_amd import ops  # noqa: E402

D = "cuda"
bf = torch.bfloat16
ops.device_check()
g = torch.Generator(device=D).manual_seed(0)


def timeit(fn, n=48):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, N, K, kw in (("q|k|v", 2304, 768, {}), ("out-projection + bias + residual", 768, 768, dict(bias=1, res=1)),
                       ("fc1 + bias + GELU", 3072, 768, dict(bias=1, gelu=1)), ("fc2 + bias + residual", 768, 3072, dict(bias=1, res=1))):
    us = {}
    for M in (16384, 16392, 16384, 16392):
        xs = [torch.randn(M, K, device=D, generator=g).to(bf) for _ in range(3)]
        ws = [(0.05 * torch.randn(N, K, device=D, generator=g)).to(bf) for _ in range(3)]
        bias = torch.randn(N, device=D, generator=g).to(bf) if kw.get("bias") else None
        res = torch.randn(M, N, device=D, generator=g).to(bf) if kw.get("res") else None
        out = torch.empty((1, M, N), dtype=bf, device=D)
        ctr = [0]

        def fn():
            ctr[0] += 1
            ops.gemm(xs[ctr[0] % 3], ws[ctr[0] % 3], bias=bias, residual=res, gelu=bool(kw.get("gelu")), out=out)
        us.setdefault(M, []).append(timeit(fn))
    a, b = min(us[16384]), min(us[16392])
    print(f"{name:34s} 16384 x {N} x {K}: {a:6.1f} us   with the 8 cls rows: {b:6.1f} us   (+{b - a:.1f})")
