"""Yardstick only (never on the product path): what does the vendor library (hipBLASLt / rocBLAS behind torch.matmul)
reach on the hot path's GEMM shapes, same random operands, same timing loop as tools/gpu_check.py."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
_scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
ops.set_gemm_scratch(_scratch)      # as in the pipeline (the split-K scratch of the tokenizer forward: the sliced big-tile products)


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
print("shape [where]: this repo us TF/s | torch.matmul (vendor library) us TF/s -- hot = the same operands every call; cold = operand "
      "sets in rotation, > 300 MB in total (beyond the 256 MB Infinity Cache: what a batch-1 pipeline sees)")
where = {(16384, 2304, 768): "ViT q|k|v", (16384, 768, 768): "ViT out-proj", (16384, 3072, 768): "ViT fc1",
         (16384, 768, 3072): "ViT fc2", (2048, 4096, 4096): "SVR out", (2048, 12288, 4096): "SVR packed qkv",
         (1792, 8192, 4096): "TTA k|v visual", (1024, 8192, 4096): "TTA k|v text", (256, 4096, 4096): "TTA query side",
         (256, 12288, 4096): "TTA packed qkv"}
for (M, N, K) in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    per_set = 2 * (M * K + N * K + M * N)
    nset = max(1, min(16, -(-320_000_000 // per_set)))
    sets = [(torch.randn(M, K, device=dev, generator=g).to(bf), torch.randn(N, K, device=dev, generator=g).to(bf),
             torch.empty((1, M, N), dtype=bf, device=dev), torch.empty((M, N), dtype=bf, device=dev)) for _ in range(nset)]
    fl = 2 * M * N * K
    res = []
    for cold in (False, True):
        it = [0]

        def pick():
            it[0] += 1
            return sets[it[0] % nset if cold else 0]

        def ours():
            a, b, out, _ = pick()
            ops.gemm(a, b, out=out)

        def vendor():
            a, b, _, o2 = pick()
            torch.matmul(a, b.t(), out=o2)

        res.append((timeit(ours, iters=2 * nset + 4), timeit(vendor, iters=2 * nset + 4)))
    (h1, h2), (c1, c2) = res
    print(f"  {M:5d}x{N:5d}x{K:4d} [{where.get((M, N, K), ''):16s}] hot {h1 * 1e3:7.1f} us {fl / h1 / 1e9:5.0f} | {h2 * 1e3:7.1f} us "
          f"{fl / h2 / 1e9:5.0f}   cold({nset:2d}) {c1 * 1e3:7.1f} us {fl / c1 / 1e9:5.0f} | {c2 * 1e3:7.1f} us {fl / c2 / 1e9:5.0f}", flush=True)
    del sets
# the ViT products with their real epilogues (vendor = matmul + the elementwise torch ops it needs)
M = 16384
x = torch.randn(M, 768, device=dev).to(bf)
w1, b1 = torch.randn(3072, 768, device=dev).to(bf), torch.randn(3072, device=dev).to(bf)
h = torch.randn(M, 3072, device=dev).to(bf)
w2, b2 = torch.randn(768, 3072, device=dev).to(bf), torch.randn(768, device=dev).to(bf)
res = torch.randn(M, 768, device=dev).to(bf)
t1 = timeit(lambda: ops.gemm(x, w1, bias=b1, gelu=True))
t2 = timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w1, b1)))
print(f"  fc1 + bias + erf-GELU 16384x3072x768: this repo {t1 * 1e3:7.1f} us | vendor matmul + torch gelu {t2 * 1e3:7.1f} us")
t1 = timeit(lambda: ops.gemm(h, w2, bias=b2, residual=res))
t2 = timeit(lambda: torch.nn.functional.linear(h, w2, b2) + res)
print(f"  fc2 + bias + residual 16384x768x3072: this repo {t1 * 1e3:7.1f} us | vendor matmul + torch add {t2 * 1e3:7.1f} us")
qkv = torch.randn(8, 12, 2049, 64, device=dev).to(bf)
q, k, v = qkv, torch.randn_like(qkv), torch.randn_like(qkv)
try:
    t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    print(f"  SDPA (8,12,2049,64) vendor: {t * 1e3:8.1f} us  {4 * 8 * 12 * 2049 * 2049 * 64 / t / 1e9:6.0f} TF/s")
except Exception as e:  # noqa: BLE001
    print("  SDPA failed:", e)
