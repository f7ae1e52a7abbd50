#!/usr/bin/env python
"""Row tiles per group of the big-tile kernels' tile walk (option gemm_big_group_m): an XCD runs group_m row tiles x 32 / group_m column
tiles at a time -- tall blocks re-read the weights, wide blocks re-read the activations.  Microseconds per launch, operand sets in rotation.

    python tools/group_m_probe.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

D = "cuda"
bf = torch.bfloat16
ops.device_check()
g = torch.Generator(device=D).manual_seed(0)


def timeit(fn, n=40):
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


GM = (8, 4, 2, 1, 16, 32, 8)
for name, M, N, K, kw in (("ViT q|k|v", 16392, 2304, 768, {}), ("ViT out-projection + bias + residual", 16392, 768, 768, dict(bias=1, res=1)),
                          ("ViT fc1 + bias + GELU", 16392, 3072, 768, dict(bias=1, gelu=1)), ("ViT fc2 + bias + residual", 16392, 768, 3072, dict(bias=1, res=1)),
                          ("SVR packed q|k|v", 2048, 12288, 4096, dict(bias=1)), ("SVR output projection", 2048, 4096, 4096, dict(bias=1)),
                          ("TTA k|v visual", 1792, 8192, 4096, dict(bias=1))):
    nset = 3 if N * K < (1 << 24) else 2
    xs = [torch.randn(M, K, device=D, generator=g).to(bf) for _ in range(nset)]
    ws = [(0.05 * torch.randn(N, K, device=D, generator=g)).to(bf) for _ in range(nset)]
    bias = torch.randn(N, device=D, generator=g).to(bf) if kw.get("bias") else None
    res = torch.randn(M, N, device=D, generator=g).to(bf) if kw.get("res") else None
    out = torch.empty((1, M, N), dtype=bf, device=D)
    ctr = [0]

    def fn():
        ctr[0] += 1
        ops.gemm(xs[ctr[0] % nset], ws[ctr[0] % nset], bias=bias, residual=res, gelu=bool(kw.get("gelu")), out=out)
    row, ref = [], None
    for gm in GM:
        ops.set_option("gemm_big_group_m", gm)
        fn()
        if ref is None:
            ops.gemm(xs[0], ws[0], bias=bias, residual=res, gelu=bool(kw.get("gelu")), out=out); ref = out.clone()
        else:
            ops.gemm(xs[0], ws[0], bias=bias, residual=res, gelu=bool(kw.get("gelu")), out=out); assert torch.equal(out, ref)
        row.append(f"{gm}: {timeit(fn):6.1f}")
    ops.set_option("gemm_big_group_m", 0)
    print(f"{name:38s} {M:5d} x {N:5d} x {K:4d}   " + "  ".join(row))
