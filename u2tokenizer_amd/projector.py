"""Spatial-pooling projector (drop-in for
/root/reference/src/model/multimodal_projector/spatial_pooling_projector.py:7-59 and builder.py:80-100)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops


class SpatialPoolingProjector(nn.Module):
    def __init__(self, image_size, patch_size, in_dim, out_dim, layer_type, layer_num, pooling_type="spatial",
                 pooling_size=2):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.pooling_size = pooling_size
        self.num_patches_pre = [img // pch for img, pch in zip(image_size, patch_size)]
        self.num_patches_post = [num // pooling_size for num in self.num_patches_pre]
        if layer_type == "linear":
            depth = int(layer_num)
            modules = [nn.Linear(in_dim, out_dim)]
            for _ in range(1, depth):
                modules.append(nn.Linear(out_dim, out_dim))
            self.projector = nn.Sequential(*modules)
        elif layer_type == "mlp":
            depth = int(layer_num)
            modules = [nn.Linear(in_dim, out_dim)]
            for _ in range(1, depth):
                modules.append(nn.GELU())
                modules.append(nn.Linear(out_dim, out_dim))
            self.projector = nn.Sequential(*modules)
        else:
            raise ValueError(f"Unknown projector layer type: {layer_type}")  # reference prints "Projector error!"
        if pooling_type not in ("spatial", "sequence"):
            raise ValueError(f"Unknown pooling type: {pooling_type}")
        self.layer_type = layer_type
        self.layer_num = int(layer_num)
        self.pooling_type = pooling_type
        self._ws = ops._Workspace()

    def forward(self, x):
        pd = next(self.parameters()).dtype   # bf16, or fp16 for a model loaded in float16 (the f16 build of the library)
        with ops.on_device(x, elem=pd):
            return self._forward(x, pd)

    def _forward(self, x, pd):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import autograd as AG  # training: the same kernels behind torch.autograd.Function
            ops.training_needs_bf16(pd, "SpatialPoolingProjector")
            ops._need(x, torch.bfloat16, "image_features")
            with ops.on_device(x):
                return AG.spp_forward(self, x)
        h = _lib.load_library()
        x = ops._need(x, ops.ELEM, "image_features").contiguous()
        nchunk, n, dim = x.shape
        g = self.num_patches_pre
        if dim != self.in_dim or n != g[0] * g[1] * g[2]:
            raise RuntimeError(f"expected (N,{g[0] * g[1] * g[2]},{self.in_dim}), got {tuple(x.shape)}")
        cfg = _lib.SppConfig(nchunk=nchunk, grid=(C.c_int32 * 3)(*g), pooling_size=self.pooling_size,
                             pooling_type=0 if self.pooling_type == "spatial" else 1, in_dim=self.in_dim,
                             out_dim=self.out_dim, layer_type=0 if self.layer_type == "mlp" else 1,
                             layer_num=self.layer_num)
        lins = [m for m in self.projector if isinstance(m, nn.Linear)]
        table = ops.weight_table([t for m in lins for t in (m.weight, m.bias)])
        nbytes = h.u2tok_spp_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("u2tok_spp_workspace_bytes rejected the configuration")
        n_out = self.proj_out_num if self.pooling_type == "spatial" else n // self.pooling_size ** 3
        with ops.on_device(x) as (h, stream):
            ws = self._ws.get(nbytes, x.device)
            out = torch.empty((nchunk, n_out, self.out_dim), dtype=pd, device=x.device)
            _lib.check(h.u2tok_spp_forward(C.byref(cfg), table, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           stream), "u2tok_spp_forward")
        return out

    @property
    def proj_out_num(self):
        num = 1
        for n in self.num_patches_post:
            num *= n
        return num
