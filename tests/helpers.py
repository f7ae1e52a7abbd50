"""Helpers shared by the CPU (oracle) and GPU (HIP vs oracle) parity tests."""
from pathlib import Path

import numpy as np
import math

import torch

from oracle import u2_oracle as O
from u2tokenizer_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}   # (names stay numpy)


def tok_cfg(c) -> O.PathConfig:
    return O.PathConfig(hidden_size=c["E"], u2t_num_heads=c["heads"], u2t_num_layers=c["layers"], u2t_top_k=c["top_k"],
                        use_multi_scale=c["use_multi_scale"], num_3d_query_token=c["Q"], attn_type=c["attn_type"],
                        enable_diffts=c["enable_diffts"], enable_dmtp=c["enable_dmtp"], max_seq_len=c["max_seq_len"])


def module_sd(module, prefix, seed, dtype=torch.float32, lively=False):
    """Name-seeded synthetic state dict for `module` (keys prefixed), in `dtype`."""
    sd = {}
    for k, v in module.state_dict().items():
        if v.is_floating_point():
            t = synth.synth_tensor(prefix + k, v.shape, seed)
            sd[prefix + k] = (synth.lively_(prefix + k, t) if lively else t).to(dtype)
    return sd


def diversity(x: torch.Tensor) -> float:
    """RMS of the token-to-token variation relative to the RMS of the tensor (0 = all tokens identical)."""
    x = x.double().reshape(-1, x.shape[-1])
    return ((x - x.mean(0, keepdim=True)).pow(2).mean().sqrt() / x.pow(2).mean().sqrt().clamp_min(1e-30)).item()


def err_stats(a: torch.Tensor, b: torch.Tensor):
    """distance of a from b.  Besides max / mean / relative-RMS: the contract's literal figure (north_star: "within 1e-3"),
    as the fraction of elements with |a - b| <= 1e-3, and the same distances in bf16 ulps of the reference value (one ulp of
    x = 2^(floor(log2 |x|) - 7): 1e-3 is below one ulp from |x| >= 0.128 on)."""
    a, b = a.double().flatten(), b.double().flatten()
    d = (a - b).abs()
    ulp = torch.exp2(torch.floor(torch.log2(b.abs().clamp_min(2.0 ** -126))) - 7)
    du = d / ulp
    rms = b.pow(2).mean().sqrt()
    return dict(max_abs=d.max().item(), mean_abs=d.mean().item(), ref_rms=rms.item(),
                rel_rms=(d.pow(2).mean().sqrt() / rms.clamp_min(1e-30)).item(),
                frac_within_1e3=(d <= 1e-3).double().mean().item(),
                frac_within_1_bf16_ulp=(du <= 1.0).double().mean().item(),
                frac_within_2_bf16_ulp=(du <= 2.0).double().mean().item(),
                abs_1e3_in_bf16_ulps_at_ref_rms=(1e-3 / 2.0 ** (math.floor(math.log2(max(rms.item(), 2.0 ** -126))) - 7)))
