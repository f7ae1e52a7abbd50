#!/usr/bin/env python
"""Measurement only: the skinny kernel (gemm_skinny.hip) with parts switched off (option gemm_skinny = 2 .. 6: WRONG results)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402
dev = torch.device("cuda", 0); ops.device_check(); bf = torch.bfloat16
scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev); ops.set_gemm_scratch(scratch)
M, N, K = 256, 4096, 4096
a = torch.randn(M, K, device=dev).to(bf); ws = [torch.randn(N, K, device=dev).to(bf) for _ in range(16)]
bias = torch.randn(N, device=dev).to(bf); out = torch.empty((1, M, N), dtype=bf, device=dev)
for mode, name in ((1, "whole"), (2, "no combine"), (3, "no MFMA"), (6, "no fragment reads / MFMA"), (4, "no weight DMA"), (5, "no activation DMA"), (0, "round-5 path")):
    ops.set_option("gemm_skinny", mode)
    if mode == 2:
        scratch[:4096].zero_()
    for i in range(16):
        ops.gemm(a, ws[i], bias=bias, out=out)
    torch.cuda.synchronize()
    if mode == 2:
        scratch[:4096].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for i in range(16):
            ops.gemm(a, ws[i], bias=bias, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 64 * 1e3:7.1f} us", flush=True)
    scratch[:4096].zero_(); torch.cuda.synchronize()
ops.set_option("gemm_skinny", 1)
