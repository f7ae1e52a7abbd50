#!/usr/bin/env python
"""Forward + backward of the path (u2tokenizer_amd/autograd.py) at the benchmark configuration (E = 4096, 256^3, batch 1)
with a dummy loss on the spliced embeddings: milliseconds per step, peak HBM, and the split forward / backward.
Measurement only.

    python tools/train_step_probe.py [hidden]
"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from u2tokenizer_amd import ops  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops.device_check()
path, _ = bench.build_path(E, 32768, dev)
g = torch.Generator(device=dev).manual_seed(1)
vol = torch.rand((1, 8, 32, 256, 256), device=dev, generator=g).half()
ids = torch.randint(1, 32768, (1, 1024), device=dev, generator=g)
qids = torch.zeros((1, 1024), dtype=torch.int64, device=dev)
qids[:, :40] = torch.randint(1, 32768, (1, 40), device=dev, generator=g)
r = bench.train_step(path, ids, qids, vol, E)
fw, bw = r["ms_forward"], r["ms_backward"]
peak = r["peak_hbm_gib"]
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        path.prepare_inputs_for_multimodal(ids, None, None, None, None, vol, qids)
    torch.cuda.synchronize()
    inf = (time.perf_counter() - t0) / 5 * 1e3
nparam = sum(p.numel() for p in path.holder.parameters())
print(f"path fwd+bwd at E={E}, 256^3, batch 1 ({nparam / 1e9:.2f} B parameters incl. a 32768-row embedding table): forward "
      f"(autograd path) {fw:.1f} ms, backward {bw:.1f} ms, peak HBM {peak:.1f} GiB; "
      f"the fused inference forward of the same path: {inf:.1f} ms")
