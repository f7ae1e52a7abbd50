#!/usr/bin/env python
"""Timing of the fused tokenizer attention (u2tok_tok_attention) at the shapes of one 256^3 volume (E = 4096 and 2048), per
key-split count, next to the GEMM -> softmax -> GEMM chain it replaces (same tensors, through the pipeline's building blocks)."""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from u2tokenizer_amd import ops  # noqa: E402

D = "cuda"
bf = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def chain(q, k, v, H, scale, tbl):
    nb, Sq, E = q.shape
    d = E // H
    Skv = k.shape[1]
    s = torch.empty((nb * H, Sq, Skv), dtype=torch.float32, device=D)
    ops.gemm_strided(q, k, s, M=Sq, N=Skv, K=d, lda=q.stride(1), ldb=k.stride(1), ldc=Skv, nz=nb * H, nbh=H, sAb=q.stride(0),
                     sAh=d, sBb=k.stride(0), sBh=d, sCb=H * Sq * Skv, sCh=Sq * Skv, out_f32=True)
    p = ops.softmax_rows(s, scale, tbl, H, 512)
    o = torch.empty((nb, Sq, E), dtype=bf, device=D)
    ops.gemm_strided(p, v, o, M=Sq, N=d, K=Skv, lda=Skv, ldb=v.stride(1), ldc=E, nz=nb * H, nbh=H, sAb=H * Sq * Skv, sAh=Sq * Skv,
                     sBb=v.stride(0), sBh=d, sCb=Sq * E, sCh=d, b_kmajor=True)
    return o


def pmc(layout):
    """20 launches of the SVR spatial shape at E = 4096 for a rocprofv3 --pmc pass (argument: tok_wide 0 / 1)"""
    ops.device_check()
    torch.set_grad_enabled(False)
    ops.set_option("tok_flash", 1)
    ops.set_option("tok_wide", layout)
    g = torch.Generator(device=D).manual_seed(0)
    E, H = 4096, 8
    qkv = torch.randn((8, 256, 3 * E), device=D, generator=g).to(bf)
    tbl = (0.2 * torch.randn((1023, H), device=D, generator=g)).to(bf)
    for _ in range(20):
        ops.tok_attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, 1 / math.sqrt(E // H), tbl, 512, 0)
    torch.cuda.synchronize()


def timed():
    """s_memtime phase breakdown (instrumented build) of the SVR spatial and the TTA visual shapes at E = 4096"""
    from u2tokenizer_amd import _lib
    ops.device_check()
    torch.set_grad_enabled(False)
    h = _lib.load_library()
    g = torch.Generator(device=D).manual_seed(0)
    E, H = 4096, 8
    wide = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ops.set_option("tok_wide", wide)
    nw = 8 if wide else 4   # waves per workgroup: the 8-wave form stamps {DMA wait, barrier, DMA issue, S(t+1), softmax, -, P V}
    names = ["dma wait", "barrier", "K dma issue" if not wide else "dma issue", "QK^T", "softmax", "V dma issue", "PV"]
    for name, nb, Sq, Skv, ns in (("svr spatial", 8, 256, 256, 1), ("tta visual ns=8", 1, 256, 1792, 8), ("tta visual ns=1", 1, 256, 1792, 1)):
        q = torch.randn((nb, Sq, E), device=D, generator=g).to(bf)
        kv = torch.randn((nb, Skv, 2 * E), device=D, generator=g).to(bf)
        grid = nb * H * ((Sq + 63) // 64) * ns
        ns_ = 16 if wide else 8   # slots per wave (8-wave form: + s_memrealtime at entry / loop start / loop end / exit)
        dbg = torch.zeros(grid * nw * ns_, dtype=torch.int64, device=D)
        h.u2tok_tok_attention_debug_buffer(dbg.data_ptr())
        ops.tok_attention(q, kv[..., :E], kv[..., E:], H, 1 / math.sqrt(E // H), None, 512, ns)
        torch.cuda.synchronize()
        h.u2tok_tok_attention_debug_buffer(None)
        d = dbg.view(-1, ns_).double().cpu()
        if wide:  # wall-clock timeline, 10 ns ticks
            r = d[:, 8:16]
            t0 = r[:, 0].min()
            print(f"{name}: prologue us: requests issued {((r[:, 4] - r[:, 0]).mean()).item() / 100:.2f}, K + Q landed {((r[:, 5] - r[:, 4]).mean()).item() / 100:.2f}, "
                  f"barrier {((r[:, 6] - r[:, 5]).mean()).item() / 100:.2f}, first scores {((r[:, 1] - r[:, 6]).mean()).item() / 100:.2f}")
            print(f"{name}: timeline us: entry spread {(r[:, 0].max() - t0).item() / 100:.2f}, prologue {((r[:, 1] - r[:, 0]).mean()).item() / 100:.2f}, "
                  f"loop {((r[:, 2] - r[:, 1]).mean()).item() / 100:.2f} (max {((r[:, 2] - r[:, 1]).max()).item() / 100:.2f}), "
                  f"epilogue {((r[:, 3] - r[:, 2]).mean()).item() / 100:.2f}, first entry -> last exit {(r[:, 3].max() - t0).item() / 100:.2f}")
        tiles = d[:, 7].mean().item()
        per = d[:, :7].mean(0) / tiles
        print(f"{name}: tiles/wave {tiles:.1f}; cycles per tile: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, per.tolist())) +
              f"; sum {per.sum().item():.0f}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "timed":
        return timed()
    if len(sys.argv) > 2 and sys.argv[1] == "pmc":
        return pmc(int(sys.argv[2]))
    ops.device_check()
    torch.set_grad_enabled(False)
    layout = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ops.set_option("tok_flash", layout)
    wide = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ops.set_option("tok_wide", wide)
    print("layout", layout, "tok_wide", wide)
    g = torch.Generator(device=D).manual_seed(0)
    for E in (4096, 2048):
        H, d = 8, E // 8
        cases = [("svr spatial", 8, 256, 256, True), ("tta self", 1, 256, 256, True), ("tta visual", 1, 256, 1792, False),
                 ("tta text", 1, 256, 1024, False)]
        for name, nb, Sq, Skv, bias in cases:
            if Sq == Skv:
                qkv = torch.randn((nb, Sq, 3 * E), device=D, generator=g).to(bf)
                q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
            else:
                q = torch.randn((nb, Sq, E), device=D, generator=g).to(bf)
                kv = torch.randn((nb, Skv, 2 * E), device=D, generator=g).to(bf)
                k, v = kv[..., :E], kv[..., E:]
            tbl = (0.2 * torch.randn((1023, H), device=D, generator=g)).to(bf) if bias else None
            scale = 1 / math.sqrt(d)
            flops = 4.0 * nb * H * Sq * Skv * d
            t_chain = timeit(lambda: chain(q, k, v, H, scale, tbl))
            ref = chain(q, k, v, H, scale, tbl).float()
            row = [f"E={E} {name:12s} chain {t_chain:7.1f} us"]
            for ns in (0, 1, 2, 4, 8, 14):
                if ns > (Skv + 31) // 32:
                    continue
                t = timeit(lambda: ops.tok_attention(q, k, v, H, scale, tbl, 512, ns))
                err = (ops.tok_attention(q, k, v, H, scale, tbl, 512, ns).float() - ref).abs().max().item()
                row.append(f"ns={ns}: {t:6.1f} us ({flops / t / 1e6:5.0f} TF/s, |d| {err:.1e})")
            print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
