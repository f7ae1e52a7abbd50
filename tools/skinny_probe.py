#!/usr/bin/env python
"""The M = 256 products of the TTA query chain (25 + 4 per volume) with COLD weights (16 matrices in rotation, 0.5-1.6 GB):
microseconds per product for tile x K-slice combinations of the classic kernel (options gemm_tile / gemm_splitk), next to
the heuristic.  Measurement only.

    python tools/skinny_probe.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops.device_check()
bf = torch.bfloat16
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
ops.set_gemm_scratch(scratch)
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in ((256, 4096, 4096), (256, 12288, 4096)):
    a = torch.randn(M, K, device=dev, generator=g).to(bf)
    ws = [torch.randn(N, K, device=dev, generator=g).to(bf) for _ in range(16)]
    bias = torch.randn(N, device=dev, generator=g).to(bf)
    out = torch.empty((1, M, N), dtype=bf, device=dev)
    ref = None
    name_once = [0]
    print(f"M={M} N={N} K={K}  (weights {N * K * 2 / 1e6:.0f} MB each, HBM floor at 5 TB/s {N * K * 2 / 5e6:.1f} us)")
    for tile, sk in ((0, 0), (0, -9), (64, -1), (64, 2), (64, 4), (64, 8), (128, -1), (128, 2), (128, 4), (128, 8), (128, 16)):
        ops.set_option("gemm_skinny", 0 if sk == -9 else 2)   # (0, 0): the heuristic = the unsplit 64 x 64 x 128 form; (0, -9): the round-5 path
        label = {-9: "round-5 path (64^2 x 4 slices)", 0: "heuristic (64x64x128 unsplit)"}.get(sk)
        sk = 0 if sk < -1 else sk
        ops.set_option("gemm_tile", tile)
        ops.set_option("gemm_splitk", sk)
        ops.set_option("gemm_big", 0 if tile == 0 else -1)
        for i in range(16):
            ops.gemm(a, ws[i], bias=bias, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(4):
            for i in range(16):
                ops.gemm(a, ws[i], bias=bias, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 64 * 1e3
        if ref is None:
            ref = out.float().clone()
        err = (out.float() - ref).abs().max().item()
        name = label if tile == 0 else f"tile {tile} splitk {sk if sk > 0 else 1}"
        print(f"   {name:28s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s   max|diff vs heuristic| {err:.3e}")
    o2 = torch.empty((M, N), dtype=bf, device=dev)
    for i in range(16):
        torch.matmul(a, ws[i].t(), out=o2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for i in range(16):
            torch.matmul(a, ws[i].t(), out=o2)
    e1.record()
    torch.cuda.synchronize()
    print(f"   {'torch.matmul (vendor)':22s} {e0.elapsed_time(e1) / 64 * 1e3:7.1f} us   (no bias)")
ops.set_option("gemm_tile", 0); ops.set_option("gemm_splitk", 0); ops.set_option("gemm_big", 0); ops.set_option("gemm_skinny", 2)
