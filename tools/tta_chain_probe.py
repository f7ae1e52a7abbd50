#!/usr/bin/env python
"""VERDICT r4 item 6 -- the two measurements owed on the TTA query chain (M = 256 rows against 4096 x 4096 weights, 22 products +
26 split-K reduces = 0.77 ms per volume):

  (a) would pulling product i + 1's 33.5 MB weight panel into the Infinity Cache from a side stream while product i runs help?
      chain of 16 dependent products over 16 different weight matrices (537 MB: cold in rotation), timed plain, with a side-stream
      reader one product ahead (a torch reduction over the next weight: 33.5 MB at HBM speed), and with the SAME weight every time
      (everything hot: the ceiling a perfect prefetch could reach);
  (b) what would the two in-flight volumes' chains cost as ONE M = 512 chain?  the same chain at M = 512 against two M = 256 chains.

Prints microseconds per product (split-K reduce included, as the pipeline runs it)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402

D, bf = "cuda", torch.bfloat16
torch.set_grad_enabled(False)
ops.device_check()
scratch = torch.empty(48 << 20, dtype=torch.uint8, device=D)
ops.set_gemm_scratch(scratch)
g = torch.Generator(device=D).manual_seed(0)
NW, E = 16, 4096
W = [(torch.randn((E, E), device=D, generator=g) / 64).to(bf) for _ in range(NW)]
bias = torch.zeros(E, device=D, dtype=bf)
side = torch.cuda.Stream()


def chain(x, prefetch=False, same=False):
    main = torch.cuda.current_stream()
    for i in range(NW):
        if prefetch and i + 1 < NW:
            ev = torch.cuda.Event()
            ev.record(main)                       # product i is next on the main stream: the reader starts with it
            side.wait_event(ev)
            with torch.cuda.stream(side):
                W[i + 1].view(torch.int32).sum()  # 33.5 MB streamed once (fabric-side read: lands in the Infinity Cache)
        x = ops.gemm(x, W[0 if same else i], bias=bias)
    if prefetch:
        main.wait_stream(side)
    return x


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n / NW * 1e3


for M in (256, 512):
    x = (torch.randn((M, E), device=D, generator=g) * 0.5).to(bf)
    cold = timeit(lambda: chain(x))
    pre = timeit(lambda: chain(x, prefetch=True))
    hot = timeit(lambda: chain(x, same=True))
    print(f"M = {M}: cold weights {cold:6.1f} us per product | side-stream reader one product ahead {pre:6.1f} | one hot weight {hot:6.1f}", flush=True)
