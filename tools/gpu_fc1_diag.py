"""GPU diagnostic: the ViT's GEMM shapes under every big-tile form -- bitwise agreement between forms (same MFMA shape, same K order),
repeatability over 6 launches each, and the time of the fc1 + GELU product at M = 16392 as the pipeline launches it."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops
dev, bf = "cuda", torch.bfloat16
ops.device_check()
torch.manual_seed(0)
def run(a, w, bias, big, **kw):
    ops.set_option("gemm_big", big)
    try:
        return ops.gemm(a, w, bias=bias, **kw).clone()
    finally:
        ops.set_option("gemm_big", 0)
for (M, N, K, kw, forms) in [(16392, 3072, 768, dict(gelu=True), (0, -1, 20, 22, 26)), (2049, 3072, 768, dict(gelu=True), (0, -1, 20, 22)),
                             (16392, 2304, 768, {}, (0, -1, 21, 24, 23)), (16392, 768, 3072, {}, (0, -1, 21, 24)), (2048, 12288, 4096, {}, (0, 21, 24, 22))]:
    a = (torch.randn(M, K, device=dev) * 1.0).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev).to(bf)
    ref = None
    for big in forms:
        outs = [run(a, w, bias, big, **kw) for _ in range(6)]
        rep = all(torch.equal(outs[0], o) for o in outs[1:])
        if ref is None:
            ref = outs[0]
        nd = int((outs[0] != ref).sum())
        md = float((outs[0].float() - ref.float()).abs().max())
        print(f"{M}x{N}x{K} {kw} big={big:3d}: repeatable={rep} differs from big=0 in {nd} elements (max {md:.3e})", flush=True)
# timing of fc1 + GELU, M = 16392, default heuristics, 12 weight sets
M, N, K = 16392, 3072, 768
a = torch.randn(M, K, device=dev).to(bf)
ws = [(torch.randn(N, K, device=dev) * 0.05).to(bf) for _ in range(12)]
bias = torch.randn(N, device=dev).to(bf)
out = torch.empty(M, N, device=dev, dtype=bf)
for opt in (1, 0, 1):
    ops.set_option("gemm_big_gelu", opt)
    for w in ws[:3]:
        ops.gemm(a, w, bias=bias, gelu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for w in ws:
            ops.gemm(a, w, bias=bias, gelu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"fc1+gelu M=16392 gemm_big_gelu={opt}: {e0.elapsed_time(e1) * 1e3 / 36:.1f} us", flush=True)
ops.set_option("gemm_big_gelu", 1)
