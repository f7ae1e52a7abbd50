#!/usr/bin/env python
"""FLAT-encoded LDS-DMA (global_load_lds) against the MUBUF form (buffer_load ... lds) in the two kernels that issue their pieces from HIP
code: tok_attn2_kernel (option tok_wide 1 / 2) and gemm_skinny64_kernel (option gemm_skinny 1 / 2).  Same bits, microseconds per launch.

    python tools/mubuf_probe.py
"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

D = "cuda"
bf = torch.bfloat16
ops.device_check()
torch.set_grad_enabled(False)
g = torch.Generator(device=D).manual_seed(0)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


E, H = 4096, 8
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=D)
ops.set_gemm_scratch(scratch)
tbl = (0.2 * torch.randn((1023, H), device=D, generator=g)).to(bf)
shapes = [("SVR spatial 8 x 256 x 256, bias", 8, 256, 256, True), ("TTA visual 1 x 256 x 1792", 1, 256, 1792, False),
          ("TTA text 1 x 256 x 1024", 1, 256, 1024, False), ("TTA self 1 x 256 x 256", 1, 256, 256, False),
          ("ragged 2 x 200 x 1000, bias", 2, 200, 1000, False)]
for name, nb, Sq, Skv, bias in shapes:
    q = torch.randn((nb, Sq, E), device=D, generator=g).to(bf)
    kv = torch.randn((nb, Skv, 2 * E), device=D, generator=g).to(bf)
    outs, us = {}, {}
    for mode in (1, 2, 1, 2):
        ops.set_option("tok_wide", mode)
        fn = lambda: ops.tok_attention(q, kv[..., :E], kv[..., E:], H, 1 / math.sqrt(E // H), tbl if bias else None, 512 if bias else 0, 0)
        outs[mode] = fn().clone()
        us.setdefault(mode, []).append(timeit(fn))
    same = torch.equal(outs[1], outs[2])
    print(f"tok_attention {name:34s} FLAT {min(us[1]):7.1f} us   MUBUF {min(us[2]):7.1f} us   same bits: {same}")
ops.set_option("tok_wide", 2)

for (M, N, K) in ((256, 4096, 4096), (200, 4096, 4096), (256, 2048, 4096)):
    a = torch.randn(M, K, device=D, generator=g).to(bf)
    ws = [torch.randn(N, K, device=D, generator=g).to(bf) for _ in range(16)]
    bias = torch.randn(N, device=D, generator=g).to(bf)
    out = torch.empty((1, M, N), dtype=bf, device=D)
    outs, us = {}, {}
    for mode in (1, 2, 1, 2):
        ops.set_option("gemm_skinny", mode)
        ctr = [0]

        def fn():
            ctr[0] += 1
            return ops.gemm(a, ws[ctr[0] % 16], bias=bias, out=out)
        ops.gemm(a, ws[0], bias=bias, out=out)
        outs[mode] = out.clone()
        us.setdefault(mode, []).append(timeit(fn, 64))
    print(f"gemm {M} x {N} x {K} (16 cold weights in rotation)   FLAT {min(us[1]):7.1f} us   MUBUF {min(us[2]):7.1f} us   same bits: {torch.equal(outs[1], outs[2])}")
ops.set_option("gemm_skinny", 2)

# the small-tile kernel (gemm.hip): the training path's products -- dW = dY^T X (both operands K-major), dX = dY W (B K-major) -- and a plain
# product on forced 128^2 tiles; option gemm_mubuf 0 / 1
scratch2 = torch.empty(256 << 20, dtype=torch.uint8, device=D)
ops.set_gemm_scratch(scratch2)
cases = [("dW fc1  (3072 x 768, K = 16392), both K-major", lambda: ops.gemm_kmajor(dy1, x1, a_kmajor=True)),
         ("dW qkv  (2304 x 768, K = 16392), both K-major", lambda: ops.gemm_kmajor(dy2, x1, a_kmajor=True)),
         ("dX fc1  (16392 x 768, K = 3072), B K-major", lambda: ops.gemm_kmajor(dy1, w1, a_kmajor=False)),
         ("dX SVR  (2048 x 4096, K = 4096), B K-major", lambda: ops.gemm_kmajor(dy3, w3, a_kmajor=False)),
         ("plain 2048 x 4096 x 4096, forced 128^2 tiles", lambda: ops.gemm(a4, w4))]
dy1 = torch.randn(16392, 3072, device=D, generator=g).to(bf)
dy2 = torch.randn(16392, 2304, device=D, generator=g).to(bf)
x1 = torch.randn(16392, 768, device=D, generator=g).to(bf)
w1 = torch.randn(3072, 768, device=D, generator=g).to(bf)
dy3 = torch.randn(2048, 4096, device=D, generator=g).to(bf)
w3 = torch.randn(4096, 4096, device=D, generator=g).to(bf)
a4, w4 = dy3, w3
for name, fn in cases:
    forced = "forced" in name
    if forced:
        ops.set_option("gemm_tile", 128); ops.set_option("gemm_big", -1)
    outs, us = {}, {}
    for mode in (0, 1, 0, 1):
        ops.set_option("gemm_mubuf", mode)
        outs[mode] = fn().clone()
        us.setdefault(mode, []).append(timeit(fn, 20))
    if forced:
        ops.set_option("gemm_tile", 0); ops.set_option("gemm_big", 0)
    print(f"{name:50s} FLAT {min(us[0]):8.1f} us   MUBUF {min(us[1]):8.1f} us   same bits: {torch.equal(outs[0], outs[1])}")
ops.set_option("gemm_mubuf", 1)
