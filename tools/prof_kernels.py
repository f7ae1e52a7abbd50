"""Tiny driver for rocprofv3 counter passes:
    python tools/prof_kernels.py flash|flashbwd|kmajor|tokattn|prefillattn|gemm|gemm4k|gemmsvr|gemm256|gemm8k|gemmmlp [iters] [gemm_big: -1 | 0 | 20 .. 26] [flash_mode]
gemm256 = the M = 256 query-side product of the TTA with 16 COLD weight matrices in rotation, as the pipeline runs it (round 6: the unsplit
64 x 64 x 128 kernel; gemm256old: 64 x 64 tiles, 4 K slices + reduce)."""
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
    ops.set_option("flash_q_prescaled", int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    qkv = torch.randn(8, 2049, 2304, device="cuda").to(bf)
    for _ in range(iters):
        ops.flash_attention_d64(qkv, 12, 0.125, extra_last=True)
elif what == "flashbwd":
    qkv = torch.randn(8, 2049, 2304, device="cuda").to(bf)
    dout = torch.randn(8, 2049, 768, device="cuda").to(bf)
    out = ops.flash_attention_d64(qkv, 12, 0.125, extra_last=True)
    for _ in range(iters):
        ops.flash_attention_d64_bwd(qkv, out, dout, 12, 0.125)
elif what == "tokattn":
    ops.set_option("tok_wide", 2 if variant == 0 else 0)   # (third argument 1: the 4-wave form)
    # the SVR's spatial attention core at E = 4096: 8 chunks, 8 heads of 512, 256 x 256, relative bias; packed q | k | v
    E, H = 4096, 8
    qkv = (torch.randn(8, 256, 3 * E, device="cuda") * 0.5).to(bf)
    rb = (torch.randn(1023, H, device="cuda") * 0.1).to(bf)
    for _ in range(iters):
        ops.tok_attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, 512 ** -0.5, rel_bias=rb, max_len=512)
elif what == "prefillattn":
    # the decoder prefill's attention: 32 query / 8 key-value heads of 128, S = 1024, causal
    qkv = (torch.randn(1, 1024, (32 + 16) * 128, device="cuda") * 0.5).to(bf)
    for _ in range(iters):
        ops.attention_gqa(qkv[..., :4096], qkv[..., 4096:5120], qkv[..., 5120:], 32, 8, 128 ** -0.5, causal=True)
elif what == "kmajor":
    # the ViT's fc1 weight gradient: dW (3072, 768) = dY (16392, 3072)^T X (16392, 768), both operands K-major
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ops.set_gemm_scratch(scratch)
    dy = torch.randn(16392, 3072, device="cuda").to(bf)
    x = torch.randn(16392, 768, device="cuda").to(bf)
    for _ in range(iters):
        ops.gemm_kmajor(dy, x, a_kmajor=True)
elif what.startswith("gemm"):
    ops.set_option("gemm_big", variant)
    if what == "gemm256old":       # the round-5 path of the M = 256 products: 64 x 64 tiles x 4 K slices + reduce
        ops.set_option("gemm_skinny", 0)
        what = "gemm256"
    M, N, K = {"gemm": (16384, 2304, 768), "gemm4k": (2048, 4096, 4096), "gemmsvr": (2048, 12288, 4096), "gemm256": (256, 4096, 4096),
               "gemm8k": (8192, 8192, 8192), "gemmmlp": (16384, 3072, 768)}[what]
    a = torch.randn(M, K, device="cuda").to(bf)
    nw = 16 if what == "gemm256" else 8 if what in ("gemm4k", "gemmsvr") else 1      # (cold weights in rotation)
    ws = [torch.randn(N, K, device="cuda").to(bf) for _ in range(nw)]
    bias = torch.randn(N, device="cuda").to(bf)
    out = torch.empty((1, M, N), dtype=bf, device="cuda")
    if what == "gemm256":
        scratch = torch.empty(24 << 20, dtype=torch.uint8, device="cuda")
        ops.set_gemm_scratch(scratch)
    for i in range(iters):
        if what == "gemmmlp":   # the ViT's fc1 as the pipeline runs it: + bias + GELU in the epilogue
            ops.gemm(a, ws[i % nw], bias=bias, gelu=True, out=out)
        else:
            ops.gemm(a, ws[i % nw], bias=bias if what == "gemm256" else None, out=out)
torch.cuda.synchronize()
