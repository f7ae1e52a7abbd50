#!/usr/bin/env python
"""Forward + backward of the path only (bench.train_step: 1 warm-up + 3 timed steps), for `rocprofv3 --kernel-trace --stats`:
the per-kernel table of the training path (tools/gpu_train.sh divides by 4 steps)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from u2tokenizer_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ops.device_check()
path, _ = bench.build_path(4096, 32768, dev)
g = torch.Generator(device=dev).manual_seed(1)
vol = torch.rand((1, 8, 32, 256, 256), device=dev, generator=g).half()
ids = torch.randint(1, 32768, (1, 1024), device=dev, generator=g)
qids = torch.zeros((1, 1024), dtype=torch.int64, device=dev)
qids[:, :40] = torch.randint(1, 32768, (1, 40), device=dev, generator=g)
r = bench.train_step(path, ids, qids, vol, 4096)
print(r["ms_forward"], r["ms_backward"])
