#!/usr/bin/env python
"""Per-shape timing of u2tok_gemm_bf16 under forced kernel choices, COLD weights (8-16 weight matrices in rotation, more than
the 256 MB Infinity Cache holds -- what the pipeline sees at batch 1): default heuristic vs the big-tile kernel with 256 / 192-wide tiles
and K slices.

    python tools/bt_sweep.py [--root DIR] [--only NAME,NAME] [shape ...]          shape = MxNxK
"""
import sys
from pathlib import Path

import torch

ROOT = Path(sys.argv[sys.argv.index("--root") + 1]).resolve() if "--root" in sys.argv else Path(__file__).resolve().parents[1]
if "--root" in sys.argv:                      # A/B of two builds on one box: import the package of another checkout
    del sys.argv[sys.argv.index("--root"):sys.argv.index("--root") + 2]
sys.path.insert(0, str(ROOT))
from u2tokenizer_amd import ops  # noqa: E402

SHAPES = [(256, 4096, 4096), (256, 12288, 4096), (2048, 4096, 4096), (1024, 8192, 4096), (1024, 6144, 4096), (1024, 4096, 4096),
          (1024, 4096, 12288), (1792, 8192, 4096), (2048, 12288, 4096)]
CONFIGS = [("default", {}), ("classic", {"gemm_big": -1, "gemm_big_skinny": 0}),
           # (round 3 measured a TWO-stage 256 x 128-tile build of the kernel under the same option value: profiles/r03_bt_sweep.log)
           ("256x128 ring", {"gemm_big": 22}), ("ring s4", {"gemm_big": 22, "gemm_big_splitk": 4}),
           ("ring s8", {"gemm_big": 22, "gemm_big_splitk": 8}), ("ring s16", {"gemm_big": 22, "gemm_big_splitk": 16}), ("192 deepA", {"gemm_big": 23}), ("192 deepB", {"gemm_big": 24}),
           ("256 deepA", {"gemm_big": 25}), ("256 deepB", {"gemm_big": 26}),
           ("256x192", {"gemm_big": 21}), ("256x192 s2", {"gemm_big": 21, "gemm_big_splitk": 2}),
           ("256x192 s4", {"gemm_big": 21, "gemm_big_splitk": 4}),
           ("256x256", {"gemm_big": 20}), ("256x256 s2", {"gemm_big": 20, "gemm_big_splitk": 2}),
           ("256x256 s4", {"gemm_big": 20, "gemm_big_splitk": 4})]
RESET = {"gemm_big": 0, "gemm_big_splitk": 0, "gemm_big_skinny": 1}


def main():
    global CONFIGS
    if "--only" in sys.argv:
        i = sys.argv.index("--only")
        keep = sys.argv[i + 1].split(",")
        del sys.argv[i:i + 2]
        CONFIGS = [c for c in CONFIGS if c[0] in keep]
    print(f"[{ROOT.name}]", flush=True)
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or SHAPES
    dev = torch.device("cuda", 0)
    ops.device_check()
    scratch = torch.empty(160 << 20, dtype=torch.uint8, device=dev)
    ops.set_gemm_scratch(scratch)
    for (M, N, K) in shapes:
        nw = max(8, min(24, (600 << 20) // (N * K * 2)))
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = None
        line = []
        for name, opts in CONFIGS:
            for k, v in {**RESET, **opts}.items():
                ops.set_option(k, v)
            try:
                for w in ws[:3]:
                    ops.gemm(a, w, residual=res, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3
                e0.record()
                for _ in range(reps):
                    for w in ws:
                        ops.gemm(a, w, residual=res, out=out)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (reps * nw)
                got = ops.gemm(a, ws[0], residual=res).float()
                if ref is None:
                    ref = got
                err = (got - ref).abs().max().item()
                line.append(f"{name} {us:.1f}us {2.0 * M * N * K / us / 1e6:.0f}TF" + (f" !diff {err:.2e}" if err > 0.26 else ""))
            except RuntimeError as ex:
                line.append(f"{name} n/a ({str(ex)[:30]})")
        for k, v in RESET.items():
            ops.set_option(k, v)
        print(f"{M}x{N}x{K}: " + " | ".join(line), flush=True)
        del ws


if __name__ == "__main__":
    main()
