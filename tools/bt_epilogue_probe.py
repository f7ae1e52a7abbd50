#!/usr/bin/env python
"""The ViT's four big-tile products (M = 16384 patch rows; bias, bias + in-place residual) with cold weights: microseconds per
launch.  `--root DIR` imports the package of another checkout (A/B of two builds on one box)."""
import sys
from pathlib import Path

import torch

root = Path(sys.argv[sys.argv.index("--root") + 1]).resolve() if "--root" in sys.argv else Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from u2tokenizer_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
ops.device_check()
tag = root.name
if "--big" in sys.argv:                      # force a big-tile variant (gemm_big option)
    ops.set_option("gemm_big", int(sys.argv[sys.argv.index("--big") + 1]))
    tag += " big=" + sys.argv[sys.argv.index("--big") + 1]
GELU = "--gelu" in sys.argv                   # only the ViT's fc1 + bias + erf-GELU product
if GELU:
    ops.set_option("gemm_big_gelu", 1)
for (M, N, K, res) in ([(16384, 3072, 768, False)] if GELU else
                       [(16384, 2304, 768, False), (16384, 768, 768, True), (16384, 768, 3072, True), (16384, 3072, 768, False),
                        (2048, 12288, 4096, False)]):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(12)]
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.randn(M, N, device=dev).to(torch.bfloat16)
    r = out if res else None                      # in place, as the ViT's residual products are
    kw = dict(gelu=True) if GELU else {}
    for w in ws[:3]:
        ops.gemm(a, w, bias=bias, residual=r, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        for w in ws:
            ops.gemm(a, w, bias=bias, residual=r, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 48
    print(f"{tag:12s} {M}x{N}x{K} {'bias+gelu' if GELU else 'bias+res' if res else 'bias    '}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s", flush=True)
