#!/usr/bin/env python
"""K slices of the training path's weight-gradient products (dW = dY^T X, both operands K-major, K = the 16392 token rows of the ViT or
the 2048 rows of the tokenizer) on the small-tile kernel: microseconds per product (kernel + reduce) for forced slice counts beside the
heuristic.   python tools/dw_splitk_probe.py
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
scratch = torch.empty(512 << 20, dtype=torch.uint8, device=D)
ops.set_gemm_scratch(scratch)


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, N, K in (("dW fc1 / fc2", 3072, 768, 16392), ("dW q|k|v", 2304, 768, 16392), ("dW out-projection", 768, 768, 16392),
                      ("dW SVR q|k|v", 12288, 4096, 2048), ("dW SVR out", 4096, 4096, 2048), ("dW patch embedding", 768, 1024, 16384)):
    dy = torch.randn(K, M, device=D, generator=g).to(bf)
    x = torch.randn(K, N, device=D, generator=g).to(bf)
    row = []
    for sk in (0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
        ops.set_option("gemm_splitk", sk)
        try:
            us = timeit(lambda: ops.gemm_kmajor(dy, x, a_kmajor=True))
            row.append(f"{'heur' if sk == 0 else sk}: {us:6.1f}")
        except Exception as e:   # noqa: BLE001
            row.append(f"{sk}: -")
    ops.set_option("gemm_splitk", 0)
    print(f"{name:20s} {M:5d} x {N:4d} x {K:5d} ({2.0 * M * N * K / 1e9:6.1f} GF)  " + "  ".join(row))
