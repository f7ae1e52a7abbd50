"""Deterministic synthetic parameters and inputs (there are no checkpoints or datasets offline).

Every tensor is drawn from its own CPU generator seeded by (seed, crc32(name)), so any two parties that agree on
the state-dict NAMES and SHAPES (reference modules in tests/golden/make_golden.py, the oracle, the HIP modules,
bench.py) get bit-identical values regardless of construction order.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Sequence

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF)
    return g


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """fp32 tensor with a scale chosen by the role the NAME implies (exercises every bias / table path)."""
    g = _gen(seed, name)
    shape = tuple(shape)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if "." not in name:  # bare names are activations / inputs: unit scale
        return x
    leaf = name.rsplit(".", 1)[-1]
    if "norm" in name and leaf == "weight" and len(shape) == 1:
        return 1.0 + 0.1 * x
    if leaf == "bias" and len(shape) == 1:
        return 0.02 * x
    if leaf == "relative_bias":
        return 0.2 * x
    if leaf in ("position_embeddings", "cls_token", "query_tokens"):
        return 0.02 * x if leaf != "query_tokens" else 0.5 * x
    if leaf == "weight" and len(shape) == 2:
        if "embed_tokens" in name or "lm_head" in name:
            return 0.05 * x
        return x / math.sqrt(shape[1])
    return 0.02 * x


def lively_(name: str, t: torch.Tensor, qk_gain: float = 4.0, sel_gain: float = 16.0) -> torch.Tensor:
    """The "lively" parameter set (in place on `t`, returned).  The SVR is a stack of attention layers WITHOUT
    residuals or norms (svr.py:29,35): with unit-scale projections its softmaxes are near-uniform and every token
    collapses onto the token mean after two layers (token diversity 1e-5 after four), which would let any downstream
    comparison pass trivially.  Scaling the query / key projections (attention logits x qk_gain^2) and the DiffTS
    score net keeps the attention selective, so the parity tests carry token-dependent data end to end."""
    if "u2tokenizer" not in name:
        return t
    if name.endswith(".wq.weight") or name.endswith(".wk.weight"):
        t.mul_(qk_gain)
    elif name.endswith(".in_proj_weight"):  # nn.MultiheadAttention: rows [q | k | v]; synth_tensor drew it at 0.02
        t.mul_(1.0 / (0.02 * math.sqrt(t.shape[1])))
        t[: 2 * t.shape[0] // 3].mul_(qk_gain)
    elif name.endswith("token_selection.score_net.weight") and t.shape[0] > 1:  # DiffTS heads (svr.py:96)
        t.mul_(sel_gain)
    return t


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "", lively: bool = False) -> None:
    """In-place: module.state_dict()[k] <- synth_tensor(prefix + k) cast to the parameter's dtype
    (through lively_() when `lively`)."""
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if v.is_floating_point():
                t = synth_tensor(prefix + k, v.shape, seed)
                if lively:
                    lively_(prefix + k, t)
                v.copy_(t.to(v.dtype))


def synth_volume(B: int, C: int, image_size: Sequence[int], seed: int = 1, dtype=torch.float16) -> torch.Tensor:
    """U[0,1) voxels (u2Transform.py:35,51 scales intensities to [0,1]) with the trailing ~20 % of depth slices of the
    volume zeroed like the depth padding of u2Transform.py:93-94.  Shape (B, C, D, H, W)."""
    g = _gen(seed, "volume")
    D, H, W = image_size
    v = torch.rand((B, C * D, H, W), generator=g, dtype=torch.float32)
    v[:, int(0.8 * C * D):] = 0.0
    return v.view(B, C, D, H, W).to(dtype)


def synth_ids(B: int, length: int, n_real: int, vocab: int, pad_id: int = 0, seed: int = 1,
              name: str = "ids") -> torch.Tensor:
    """n_real random ids right-padded with pad_id to `length` (fused_dataset.py:177-179)."""
    g = _gen(seed, name)
    ids = torch.full((B, length), pad_id, dtype=torch.int64)
    ids[:, :n_real] = torch.randint(1, vocab, (B, n_real), generator=g)
    return ids
