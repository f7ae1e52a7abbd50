#!/usr/bin/env python
"""Which stage of the path is not bit-repeatable call after call?  (config 3: E = 4096, 256^3, batch 1)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from u2tokenizer_amd import ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
path, _ = bench.build_path(4096, 4096, dev)
h = path.holder
g = torch.Generator(device=dev).manual_seed(1)
vol = torch.rand((1, 8, 32, 256, 256), device=dev, generator=g).half()
ids = torch.randint(1, 4096, (1, 1024), device=dev, generator=g)
qids = torch.zeros((1, 1024), dtype=torch.int64, device=dev)
qids[:, :40] = torch.randint(1, 4096, (1, 40), device=dev, generator=g)


def rep(name, fn, n=4):
    outs = [fn().clone() for _ in range(n)]
    torch.cuda.synchronize()
    same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
    d = max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:])
    print(f"{name:50s} repeatable={same} max_abs_diff={d:.3e}", flush=True)
    return outs[0]


vit = rep("vit", lambda: h.vision_tower(vol.view(8, 1, 32, 256, 256)))
spp = rep("spp", lambda: h.mm_projector(vit))
tt = rep("embed lookup", lambda: ops.embed_splice(h.embed_tokens.weight, qids))
v = spp.view(1, 8, 256, 4096)
for ov in (1, 0):
    ops.set_option("tta_overlap", ov)
    rep(f"tokenizer tta_overlap={ov}", lambda: h.u2tokenizer(v_token=v, t_token=tt))
    for sk in (0, -1):
        ops.set_option("gemm_splitk", sk)
        rep(f"tokenizer tta_overlap={ov} gemm_splitk={sk}", lambda: h.u2tokenizer(v_token=v, t_token=tt))
    ops.set_option("gemm_splitk", 0)
ops.set_option("tta_overlap", 1)
rep("whole path", lambda: path.prepare_inputs_for_multimodal(ids, None, None, None, None, vol, qids)[4])
x = torch.randn(1, 1024, 4096, device=dev, generator=g).to(torch.bfloat16)
gw, gb = torch.randn(1, 4096, device=dev, generator=g).to(torch.bfloat16), torch.randn(1, device=dev, generator=g).to(torch.bfloat16)
rep("multiscale pool (gated)", lambda: ops.multiscale_pool(x, gw, gb))
