#!/usr/bin/env python
"""Big-tile GEMM kernel on the hot shapes of the path, warm operands: microseconds per product (TF/s).  The start-stagger
experiment this was written for (profiles/r02_bt_stagger_probe.log) is gone; pass option values to sweep another switch.
Measurement only.      python tools/bt_probe.py [option_name v0 v1 ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops.device_check()
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
OPT = sys.argv[1] if len(sys.argv) > 2 else None
SWEEP = [int(v) for v in sys.argv[2:]] if OPT else [0]
shapes = ((16384, 2304, 768, False), (16384, 3072, 768, False), (16384, 3072, 768, True), (16384, 768, 3072, False),
          (2048, 12288, 4096, False), (1792, 8192, 4096, False))
for (M, N, K, gelu) in shapes:
    a = torch.randn(M, K, device=dev, generator=g).to(bf)
    w = torch.randn(N, K, device=dev, generator=g).to(bf)
    bias = torch.randn(N, device=dev, generator=g).to(bf)
    out = torch.empty((1, M, N), dtype=bf, device=dev)
    row = f"{M}x{N}x{K}{' gelu' if gelu else ''}:"
    for stag in SWEEP:
        if OPT:
            ops.set_option(OPT, stag)
        ops.set_option("gemm_big_gelu", 1)
        for _ in range(3):
            ops.gemm(a, w, bias=bias, gelu=gelu, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, bias=bias, gelu=gelu, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row += f"  s{stag}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:5.0f})"
    print(row)
if OPT:
    ops.set_option(OPT, 0)
ops.set_option("gemm_big_gelu", 1)
