#!/usr/bin/env python
"""Interleaved A/B timing of launcher options on ONE box and ONE process (boxes of the pool differ by +-4 %, and a fresh
process warms differently): the configuration of bench.py (E = 4096, 256^3, batch 1) is built once, then every variant
is timed in turn, round after round; the medians are comparable with each other.

    python tools/ab_bench.py "name:opt=val,opt=val[:streams]" ...      e.g.  base: gelu:gemm_big_gelu=1 s3::3
"""
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from u2tokenizer_amd import ops  # noqa: E402

DEFAULTS = dict(ln_wide=1, gemm_mubuf=1, gemm_big=0, gemm_big_ring=1, gemm_big_deep=1, gemm_big_skinny=1, gemm_big_gelu=1, gemm_splitk=0, flash_mode=0, tta_overlap=1, gemm_tile=0, kmajor_b=1, tok_flash=1, tok_wide=2, vit_vt_epilogue=1, gemm_tail_fused=1, gemm_big_drain=1, gemm_skinny=2, gemm_big_grid=256)


def main():
    variants = []
    for spec in sys.argv[1:] or ["base:"]:
        parts = spec.split(":")
        name, opts = parts[0], dict(DEFAULTS)
        if len(parts) > 1 and parts[1]:
            for kv in parts[1].split(","):
                k, v = kv.split("=")
                opts[k] = int(v)
        variants.append((name, opts, int(parts[2]) if len(parts) > 2 and parts[2] else 2))
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ops.device_check()
    E, vocab, steps, rounds = 4096, 151936, 20, 5
    path, _ = bench.build_path(E, vocab, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    vols = [torch.rand((1, 8, 32, 256, 256), device=dev, generator=g).half() for _ in range(4)]
    ids = torch.randint(1, vocab, (1, 1024), device=dev, generator=g)
    qids = torch.zeros((1, 1024), dtype=torch.int64, device=dev)
    qids[:, :40] = torch.randint(1, vocab, (1, 40), device=dev, generator=g)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    torch.cuda.synchronize()

    def run(ns, n):
        for i in range(n):
            if ns <= 1:
                path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % 4], qids)
            else:
                with torch.cuda.stream(streams[i % ns]):
                    path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % 4], qids)
        torch.cuda.synchronize()

    times = {name: [] for name, _, _ in variants}
    for r in range(rounds + 1):
        for name, opts, ns in variants:
            for k, v in opts.items():
                ops.set_option(k, v)
            run(ns, 3)
            t0 = time.perf_counter()
            run(ns, steps)
            if r:  # round 0 warms every variant up
                times[name].append((time.perf_counter() - t0) / steps * 1e3)
    for k, v in DEFAULTS.items():
        ops.set_option(k, v)
    base = statistics.median(times[variants[0][0]])
    for name, opts, ns in variants:
        ms = statistics.median(times[name])
        diff = {k: v for k, v in opts.items() if v != DEFAULTS[k]}
        print(f"{name:14s} streams={ns} {diff!s:40s} {ms:7.3f} ms/volume  {1e3 / ms:7.2f} vol/s  x{base / ms:5.3f}  "
              f"[{min(times[name]):.3f} .. {max(times[name]):.3f}]", flush=True)


if __name__ == "__main__":
    main()
