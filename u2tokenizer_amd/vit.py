"""3-D ViT tower (drop-in for /root/reference/src/model/multimodal_encoder/vit.py).

State-dict keys follow MONAI 1.3.0's PatchEmbeddingBlock / TransformerBlock / SABlock / MLPBlock names so
that M3D-CLIP `pretrained_ViT.bin` loads with strict=True (reference: u2_arch.py:64-66).  The modules
below only own parameters; `ViT.forward` hands their device pointers to `u2tok_vit_forward`
(include/u2tok.h), which runs im2col -> patch-embed GEMM -> 12 x {LN, QKV GEMM, flash attention,
out-proj GEMM(+residual), LN, MLP GEMMs(+GELU,+residual)} -> LN on one HIP stream.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence

import torch
import torch.nn as nn

from . import _lib, ops


class _Rearrange(nn.Module):
    """Placeholder at index 0 of `patch_embeddings` (einops Rearrange in MONAI; parameter-free) so the Linear
    keeps its MONAI key `patch_embeddings.1.*`.  The im2col itself is a HIP kernel."""

    def forward(self, x):  # pragma: no cover - never called on the HIP path
        raise RuntimeError("patch rearrange runs inside u2tok_vit_forward")


class PatchEmbeddingBlock(nn.Module):
    """MONAI PatchEmbeddingBlock(pos_embed="perceptron") parameters (vit.py:90-99)."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed="perceptron",
                 dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        if pos_embed != "perceptron":
            raise ValueError("only pos_embed='perceptron' is used by ViT3DTower (vit.py:143)")
        if in_channels != 1 or spatial_dims != 3:
            raise ValueError("u2tok HIP patch embedding supports image_channel=1, 3 spatial dims (config.json:36-42)")
        for m, p in zip(img_size, patch_size):
            if m % p != 0:
                raise ValueError("patch_size should be divisible by img_size for perceptron.")
        self.n_patches = 1
        for m, p in zip(img_size, patch_size):
            self.n_patches *= m // p
        self.patch_dim = int(in_channels * patch_size[0] * patch_size[1] * patch_size[2])
        self.patch_embeddings = nn.Sequential(_Rearrange(), nn.Linear(self.patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        nn.init.trunc_normal_(self.patch_embeddings[1].weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
        nn.init.zeros_(self.patch_embeddings[1].bias)
        # MONAI releases up to 1.3.x also register a `cls_token` parameter on this block that nothing reads, so a
        # `pretrained_ViT.bin` written with one of them carries `patch_embedding.cls_token`; later releases do not.  The strict
        # load of u2_arch.py:64-66 has to pass either way: the key is adopted when a checkpoint has it (and then written back by
        # state_dict(), so a re-save is lossless) and absent otherwise.
        self._register_load_state_dict_pre_hook(self._adopt_unused_cls_token)

    def _adopt_unused_cls_token(self, state_dict, prefix, *_):
        key = prefix + "cls_token"
        if key in state_dict and not hasattr(self, "cls_token"):
            ref = self.position_embeddings
            self.cls_token = nn.Parameter(torch.zeros(tuple(state_dict[key].shape), dtype=ref.dtype, device=ref.device),
                                          requires_grad=False)


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        if qkv_bias:
            raise ValueError("qkv_bias=True is not used by the reference tower (vit.py:47,101)")
        self.mlp = MLPBlock(hidden_size, mlp_dim)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(hidden_size)


class ViT(nn.Module):
    """Same constructor as the reference ViT (vit.py:30-47); classification head is disabled there too."""

    def __init__(self, in_channels: int, img_size: Sequence[int] | int, patch_size: Sequence[int] | int,
                 hidden_size: int = 768, mlp_dim: int = 3072, num_layers: int = 12, num_heads: int = 12,
                 pos_embed: str = "conv", classification: bool = False, num_classes: int = 2,
                 dropout_rate: float = 0.0, spatial_dims: int = 3, post_activation="Tanh", qkv_bias: bool = False,
                 save_attn: bool = False) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if hidden_size // num_heads != 64:
            raise ValueError("u2tok flash attention kernel is built for head_dim 64 (ViT-B: 768/12)")
        self.hidden_size = hidden_size
        self.mlp_dim = mlp_dim
        self.num_heads = num_heads
        self.img_size = list(img_size)
        self.patch_size = list(patch_size)
        self.classification = classification
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads,
                                                   pos_embed, dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList(
            [TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias, save_attn)
             for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)
        if self.classification:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        self._ws = ops._Workspace()

    def _weights(self):
        if not hasattr(self, "cls_token"):
            raise RuntimeError("u2tok_vit_forward expects the cls-token variant (classification=True, vit.py:145)")
        w = [self.patch_embedding.position_embeddings, self.patch_embedding.patch_embeddings[1].weight,
             self.patch_embedding.patch_embeddings[1].bias, self.cls_token]
        for b in self.blocks:
            w += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.out_proj.weight, b.attn.out_proj.bias,
                  b.norm2.weight, b.norm2.bias, b.mlp.linear1.weight, b.mlp.linear1.bias, b.mlp.linear2.weight,
                  b.mlp.linear2.bias]
        w += [self.norm.weight, self.norm.bias]
        return w

    def forward_features(self, x: torch.Tensor, keep_cls: bool) -> torch.Tensor:
        """x: (nchunk, 1, D, H, W) fp16/bf16/fp32 on the GPU -> (nchunk, ntok[+1], hidden) in the parameters' 16-bit type
        (bf16, or fp16 for a model loaded in float16 -- evalscipt/ourmodel_amos.py:33: the f16 build of the library)."""
        pd = self.norm.weight.dtype
        if not x.is_cuda:
            raise RuntimeError("images: expected a GPU tensor (the u2tok HIP path has no CPU fallback)")
        with ops.on_device(x, elem=pd):
            return self._forward_features(x, keep_cls, pd)

    def _forward_features(self, x, keep_cls, pd):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            ops.training_needs_bf16(pd, "ViT")
            # training (train_stage1.py:42 runs with freeze_vision_tower False): autograd path, same kernels (autograd.py)
            from . import autograd as AG
            if x.dim() != 5 or x.shape[1] != 1 or list(x.shape[2:]) != self.img_size or not x.is_cuda:
                raise RuntimeError(f"expected GPU images of shape (N,1,{self.img_size}), got {tuple(x.shape)} on {x.device}")
            self._weights()  # (raises for the variant without a cls token)
            with ops.on_device(x, elem=pd):
                return AG.vit_forward(self, x, keep_cls)
        h = _lib.load_library()
        if x.dim() != 5 or x.shape[1] != 1 or list(x.shape[2:]) != self.img_size:
            raise RuntimeError(f"expected images of shape (N,1,{self.img_size}), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("images: expected a GPU tensor (the u2tok HIP path has no CPU fallback)")
        x = x.contiguous()
        nchunk = x.shape[0]
        cfg = _lib.VitConfig(nchunk=nchunk, img=(C.c_int32 * 3)(*self.img_size),
                             patch=(C.c_int32 * 3)(*self.patch_size), hidden=self.hidden_size, mlp_dim=self.mlp_dim,
                             depth=len(self.blocks), heads=self.num_heads, vol_dtype=ops.vol_dtype_code(x.dtype),
                             keep_cls=int(keep_cls), ln_eps=self.norm.eps)
        table = ops.weight_table(self._weights())
        nbytes = h.u2tok_vit_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("u2tok_vit_workspace_bytes rejected the configuration")
        ntok = self.patch_embedding.n_patches + (1 if keep_cls else 0)
        with ops.on_device(x, elem=pd) as (h, stream):
            ws = self._ws.get(nbytes, x.device)
            out = torch.empty((nchunk, ntok, self.hidden_size), dtype=pd, device=x.device)
            _lib.check(h.u2tok_vit_forward(C.byref(cfg), table, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           stream), "u2tok_vit_forward")
        return out

    def forward(self, x):
        # reference returns (last_feature, hidden_states_out) (vit.py:114-126); per-block hidden states are not
        # materialised on the HIP path (only select_layer == -1 is reachable: vit.py:152-153 indexes a list with
        # a string).
        return self.forward_features(x, keep_cls=True), []


class ViT3DTower(nn.Module):
    """Drop-in for ViT3DTower (vit.py:132-176)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.select_layer = config.vision_select_layer
        self.select_feature = config.vision_select_feature
        self.vision_tower = ViT(in_channels=self.config.image_channel, img_size=self.config.image_size,
                                patch_size=self.config.patch_size, pos_embed="perceptron",
                                spatial_dims=len(self.config.patch_size), classification=True)
        # with a FROZEN tower (train_stage1.py:56 `freeze_vision_tower`, u2_arch.py:58) the DPO step runs the
        # same ViT on the same images twice -- once for the policy, once for the reference model (dpo_u2trainer.py builds
        # `cat([images, images])` anew for each).  Opt-in (`share_frozen_vision_tower`): the tower keeps its last input and
        # output and returns the output again when the next input is byte-identical and no parameter has changed.
        self.share_frozen_features = False
        self._feat_cache = None

    def _frozen_key(self):
        """What the cached features depend on besides the images: every parameter's storage and version counter.  Writes
        through `.data` do not bump version counters (an optimiser's copy-out, `p.data.copy_`): the paths that do that to a
        whole model -- _apply (.to / .cuda / .bfloat16), load_state_dict, train() -- drop the cache explicitly, and
        `invalidate_feature_cache()` is there for anything else that rewrites frozen weights in place."""
        ps = list(self.parameters())
        if any(p.requires_grad for p in ps):
            return None
        return (self.select_feature, tuple((p.data_ptr(), p._version) for p in ps), ps[0].device)

    def invalidate_feature_cache(self) -> None:
        self._feat_cache = None

    def _apply(self, fn, *args, **kwargs):
        self._feat_cache = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._feat_cache = None
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode: bool = True):
        self._feat_cache = None
        return super().train(mode)

    def forward(self, images):
        if self.select_layer != -1:
            raise ValueError(f"Unexpected select layer: {self.select_layer}")
        if self.select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        key = self._frozen_key() if self.share_frozen_features else None
        if key is not None and self._feat_cache is not None:
            k0, img0, out0 = self._feat_cache
            if k0 == key and img0.shape == images.shape and img0.dtype == images.dtype and img0.device == images.device \
                    and torch.equal(img0, images):
                return out0.clone()   # (both models get their own tensor: an in-place op downstream cannot reach the cache)
        out = self.vision_tower.forward_features(images, keep_cls=self.select_feature == "cls_patch")
        if key is not None:   # the cache keeps its OWN tensor: the first caller may write into `out` in place too
            self._feat_cache = (key, images.detach().clone(), out.clone())
        return out

    @property
    def dtype(self):
        return self.vision_tower.norm.weight.dtype

    @property
    def device(self):
        return self.vision_tower.norm.weight.device

    @property
    def hidden_size(self):
        return self.vision_tower.hidden_size


def share_frozen_vision_tower(policy, reference) -> None:
    """DPO with a frozen vision tower: make `reference` (the frozen reference model) use `policy`'s tower
    module and let that tower return its last features for a byte-identical input, so the second model's pass over the same
    images costs one comparison instead of twelve ViT blocks.  Both towers must be frozen and hold equal weights (checked)."""
    tp, tr = policy.get_model().get_vision_tower(), reference.get_model().get_vision_tower()
    if any(p.requires_grad for p in tp.parameters()) or any(p.requires_grad for p in tr.parameters()):
        raise RuntimeError("share_frozen_vision_tower: both vision towers must be frozen (requires_grad False)")
    sp, sr = tp.state_dict(), tr.state_dict()
    if sp.keys() != sr.keys() or any(not torch.equal(sp[k], sr[k].to(sp[k].device)) for k in sp):
        raise RuntimeError("share_frozen_vision_tower: the two towers hold different weights")
    reference.get_model().vision_tower = tp
    tp.share_frozen_features = True

