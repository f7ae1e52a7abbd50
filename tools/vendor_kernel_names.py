#!/usr/bin/env python
"""Yardstick study only (never on the product path): which kernels does torch.matmul (hipBLASLt / Tensile) pick for the shapes this repo is
behind on?  Run under `rocprofv3 --kernel-trace --stats`: the Tensile kernel names spell the macro tile (MT), split-K (GSU), wave tiling
and prefetch depth."""
import torch
dev, bf = "cuda", torch.bfloat16
for (M, N, K) in ((256, 4096, 4096), (1792, 8192, 4096), (16384, 3072, 768), (4096, 4096, 4096), (8192, 8192, 8192), (16384, 768, 768)):
    a = torch.randn(M, K, device=dev).to(bf)
    ws = [torch.randn(N, K, device=dev).to(bf) for _ in range(6)]
    o = torch.empty(M, N, device=dev, dtype=bf)
    for i in range(12):
        torch.matmul(a, ws[i % 6].t(), out=o)
    torch.cuda.synchronize()
