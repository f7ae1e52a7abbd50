"""Tiny driver for rocprofv3 counter passes: python tools/prof_kernels.py flash|gemm|gemm4k|gemm256|gemm8k [iters] [gemm_pp variant] [flash_mode]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bf = torch.bfloat16
torch.manual_seed(0)
if what == "flash":
    ops.set_option("flash_mode", int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    qkv = torch.randn(8, 2049, 2304, device="cuda").to(bf)
    for _ in range(iters):
        ops.flash_attention_d64(qkv, 12, 0.125, extra_last=True)
elif what.startswith("gemm"):
    ops.set_option("gemm_pp", variant)
    M, N, K = {"gemm": (16384, 2304, 768), "gemm4k": (2048, 4096, 4096), "gemm256": (256, 4096, 4096),
               "gemm8k": (8192, 8192, 8192), "gemmmlp": (16384, 3072, 768)}[what]
    a, b = torch.randn(M, K, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
    out = torch.empty((1, M, N), dtype=bf, device="cuda")
    for _ in range(iters):
        ops.gemm(a, b, out=out)
torch.cuda.synchronize()
