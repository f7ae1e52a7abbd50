#!/usr/bin/env python
"""LayerNorm of the tokenizer's long rows (C = 4096 / 2048): a wave per row (option ln_wide 0) against a workgroup per row (1), operands in
rotation; microseconds per launch and the largest difference between the two (summation order of the two reductions).

    python tools/ln_probe.py
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
for rows, C, with_res in ((2048, 4096, False), (2048, 4096, True), (256, 4096, True), (1024, 4096, False), (1792, 4096, False), (2048, 2048, False),
                          (16392, 768, False)):
    xs = [torch.randn(rows, C, device=D, generator=g).to(bf) for _ in range(8)]
    rs = [torch.randn(rows, C, device=D, generator=g).to(bf) for _ in range(8)] if with_res else [None] * 8
    w, b = torch.randn(C, device=D, generator=g).to(bf), torch.randn(C, device=D, generator=g).to(bf)
    us, outs = {}, {}
    for mode in (0, 1, 0, 1):
        ops.set_option("ln_wide", mode)
        outs[mode] = ops.layernorm(xs[0], w, b, residual=rs[0]).float()
        for i in range(8):
            ops.layernorm(xs[i], w, b, residual=rs[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r_ in range(8):
            for i in range(8):
                ops.layernorm(xs[i], w, b, residual=rs[i])
        e1.record()
        torch.cuda.synchronize()
        us.setdefault(mode, []).append(e0.elapsed_time(e1) / 64 * 1e3)
    mb = rows * C * 2 * (3 if with_res else 2) / 1e6
    print(f"{rows:6d} x {C:5d}{' + residual' if with_res else '':11s}  wave per row {min(us[0]):6.1f} us   workgroup per row {min(us[1]):6.1f} us"
          f"   ({mb:5.1f} MB: {mb / min(us[1]):.2f} TB/s)   max |diff| {(outs[0] - outs[1]).abs().max().item():.3g}")
ops.set_option("ln_wide", 1)
