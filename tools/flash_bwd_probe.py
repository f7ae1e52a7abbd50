#!/usr/bin/env python
"""Attention backward of one ViT layer at the benchmark shape (8 chunks x 2049 tokens x 12 heads x 64): the fused kernel
pair of attn_bwd.hip against the unfused chain (probabilities rebuilt in HBM), microseconds per call, agreement, and the
matrix-pipe rate of the fused pair (8 matmul units of 2 S^2 64 flop per head).

    python tools/flash_bwd_probe.py [nb S H]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import autograd as AG  # noqa: E402
from u2tokenizer_amd import ops  # noqa: E402

nb, S, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 2049, 12)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops.device_check()
AG.ensure_gemm_scratch(dev)
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(nb, S, 3 * H * 64, device=dev, generator=g).to(torch.bfloat16)
dO = torch.randn(nb, S, H * 64, device=dev, generator=g).to(torch.bfloat16)
out, lse = ops.flash_attention_d64(qkv, H, 0.125, extra_last=S > 1, return_lse=True)
E = H * 64


def fused():
    return ops.flash_attention_d64_bwd(qkv, out, dO, H, 0.125, lse=lse)


def fused_own_stats():
    return ops.flash_attention_d64_bwd(qkv, out, dO, H, 0.125)


def unfused():
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    P = AG._attn_probs(q, k, H, 0.125, None, 0)
    d = torch.empty_like(qkv)
    AG._attn_backward(q, k, v, P, dO, H, 0.125, d[..., :E], d[..., E:2 * E], d[..., 2 * E:], None, 0)
    return d


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, r


tf, df = timeit(fused)
ts, _ = timeit(fused_own_stats)
tu, du = timeit(unfused)
unit = 2.0 * nb * H * S * S * 64
print(f"attention backward nb={nb} S={S} H={H}: fused with the forward's row statistics {tf:.1f} us ({7 * unit / tf / 1e6:.0f} "
      f"TF/s over its 7 matmul units), rebuilding them {ts:.1f} us ({8 * unit / ts / 1e6:.0f} TF/s over 8), unfused chain "
      f"{tu:.1f} us, x{tu / tf:.2f}")
for i, n in enumerate(("dq", "dk", "dv")):
    a, b = df[..., i * E:(i + 1) * E].float(), du[..., i * E:(i + 1) * E].float()
    print(f"  {n}: rel rms fused vs unfused {((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item():.3e}, "
          f"rms {b.pow(2).mean().sqrt().item():.3e}")
