"""Tiny driver for rocprofv3 counter passes: python tools/prof_kernels.py flash|gemm|gemm256 [iters]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bf = torch.bfloat16
torch.manual_seed(0)
if what == "flash":
    qkv = torch.randn(8, 2049, 2304, device="cuda").to(bf)
    for _ in range(iters):
        ops.flash_attention_d64(qkv, 12, 0.125)
elif what.startswith("gemm"):
    M, N, K = {"gemm": (16392, 2304, 768), "gemm4k": (2048, 4096, 4096), "gemm256": (256, 4096, 4096)}[what]
    a, b = torch.randn(M, K, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
    for _ in range(iters):
        ops.gemm(a, b)
torch.cuda.synchronize()
