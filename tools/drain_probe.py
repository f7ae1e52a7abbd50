#!/usr/bin/env python
"""Round 6: the drain form of the deep 256 x 192 big-tile kernel (gemm_bt_drain_kernel: tile i's epilogue under tile i + 1's K loop)
against the forms it replaces, cold weights (12 weight matrices in rotation), the ViT's M = 16384 patch rows (+ 8 cls rows with --tail):

    fc1 + bias + GELU  16384 x 3072 x 768   drain (four rounds of 192-wide tiles)   vs  256 x 256 two-stage, GELU exposed
    q | k | v          16384 x 2304 x 768   drain                                  vs  deep 256 x 192
    bias only          16384 x 3072 x 768   drain                                  vs  deep 256 x 192 / 256 x 256

    python tools/drain_probe.py [--tail] [--root DIR]
"""
import sys
from pathlib import Path

import torch

root = Path(sys.argv[sys.argv.index("--root") + 1]).resolve() if "--root" in sys.argv else Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from u2tokenizer_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
ops.device_check()
TAIL = 8 if "--tail" in sys.argv else 0


def run(M, N, K, bias, gelu, opts):
    for k, v in opts.items():
        ops.set_option(k, v)
    a = torch.randn(M + TAIL, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(12)]
    b = torch.randn(N, device=dev).to(torch.bfloat16) if bias else None
    out = torch.empty(M + TAIL, N, dtype=torch.bfloat16, device=dev)
    for w in ws[:3]:
        ops.gemm(a, w, bias=b, gelu=gelu, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            for w in ws:
                ops.gemm(a, w, bias=b, gelu=gelu, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 48)
    for k in opts:
        ops.set_option(k, {"gemm_big_drain": 1}.get(k, 0))
    return best


for name, (M, N, K, bias, gelu) in {"fc1 + bias + GELU": (16384, 3072, 768, True, True), "q|k|v (no bias)": (16384, 2304, 768, False, False),
                                    "fc1 shape, bias only": (16384, 3072, 768, True, False), "fc1 shape, plain": (16384, 3072, 768, False, False),
                                    "K = 3072 (N = 2304)": (16384, 2304, 3072, True, False)}.items():
    row = []
    for label, opts in (("drain", {"gemm_big_drain": 1}), ("no drain (heuristic)", {"gemm_big_drain": 0}),
                        ("deep 256x192", {"gemm_big_drain": 0, "gemm_big": 24}), ("256x256", {"gemm_big_drain": 0, "gemm_big": 20})):
        if TAIL and "gemm_big" in opts:
            continue
        us = run(M, N, K, bias, gelu, opts)
        row.append(f"{label} {us:6.1f} us ({2.0 * M * N * K / us / 1e6:4.0f} TF/s)")
    print(f"{name:22s} {M + TAIL}x{N}x{K}: " + " | ".join(row), flush=True)
