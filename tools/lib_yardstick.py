"""Yardstick only (never on the product path): what does the vendor library (hipBLASLt / rocBLAS behind torch.matmul)
reach on the hot path's GEMM shapes, same random operands, same timing loop as tools/gpu_check.py."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = [(16384, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (2048, 4096, 4096),
          (2048, 12288, 4096), (1792, 8192, 4096), (1024, 8192, 4096), (256, 4096, 4096), (256, 12288, 4096),
          (4096, 4096, 4096), (8192, 8192, 8192)]
print("shape: this repo (classic kernel) us/TFs | torch.matmul (vendor library) us/TFs")
for (M, N, K) in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(bf)
    b = torch.randn(N, K, device=dev, generator=g).to(bf)
    out = torch.empty((1, M, N), dtype=bf, device=dev)
    o2 = torch.empty((M, N), dtype=bf, device=dev)
    t1 = timeit(lambda: ops.gemm(a, b, out=out))
    t2 = timeit(lambda: torch.matmul(a, b.t(), out=o2))
    fl = 2 * M * N * K
    print(f"  {M:5d}x{N:5d}x{K:4d}  {t1 * 1e3:8.1f} us {fl / t1 / 1e9:6.0f} | {t2 * 1e3:8.1f} us {fl / t2 / 1e9:6.0f}", flush=True)
qkv = torch.randn(8, 12, 2049, 64, device=dev).to(bf)
q, k, v = qkv, torch.randn_like(qkv), torch.randn_like(qkv)
try:
    t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    print(f"  SDPA (8,12,2049,64) vendor: {t * 1e3:8.1f} us  {4 * 8 * 12 * 2049 * 2049 * 64 / t / 1e9:6.0f} TF/s")
except Exception as e:  # noqa: BLE001
    print("  SDPA failed:", e)
