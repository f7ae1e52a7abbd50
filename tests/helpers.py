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


# ---- decoders and volumes for the greedy-id gates (tests/test_gpu_configs.py)
def decisive_decoder_(m, dseed, embed_gain=4.0, loud=8, loud_gain=4.0):
    """Re-draws the decoder side of a u2*ForCausalLM IN PLACE so that its greedy decisions are worth comparing.

    A random-init decoder with the usual 0.05-scale embeddings decides nothing: its residual stream is the mean over the
    context, the same vector at every step, so it emits one id for ever, whatever the image was (constant
    ids [1844] x 4, top-2 margins below the bf16 noise).  Here (a) the embedding table is scaled by `embed_gain` (done BEFORE the
    oracle runs: the table also conditions the tokenizer) so that the current token matters next to the context and the
    sequence moves; (b) `loud` rows of lm_head, drawn from `dseed`, are scaled by `loud_gain`: the argmax is decided among
    a handful of candidates whose mutual gaps are large against the logit noise of a bf16 run (gap / sigma of the top two
    of N gaussians ~ 1 / sqrt(2 ln N)); (c) the decoder layers / final norm / lm_head are drawn from `dseed` alone -- the
    seeds used by the tests were picked by a HOST-side scan of the fp32 reference only (margins of its own decisions
    against its own bf16 run, tools/scan_id_gate_seeds.py); the tests re-assert those properties of the reference before
    they compare the HIP run with it."""
    import torch
    with torch.no_grad():
        m.get_input_embeddings().weight.mul_(embed_gain)
        for k, v in m.state_dict().items():
            if k.startswith(("model.layers", "model.norm", "lm_head")) and v.is_floating_point():
                v.copy_(synth.synth_tensor(k, v.shape, dseed).to(v.dtype))
        g = torch.Generator().manual_seed(dseed)
        rows = torch.randperm(m.lm_head.weight.shape[0], generator=g)[:loud]
        m.lm_head.weight[rows] *= loud_gain
    return m


def smooth_volume(B, C, image_size):
    """A second, structurally different volume for the image-sensitivity control: a smooth oblique sinusoid in [0, 1] (the
    noise volumes of synth_volume all look alike to the ViT: their tokens differ by ~9 %; this one moves them by 35-80 %,
    the reference's bf16 noise being 1-2 %).  Shape (B, C, D, H, W), fp16."""
    import torch
    D, H, W = image_size
    z, y, x = torch.meshgrid(torch.arange(C * D), torch.arange(H), torch.arange(W), indexing="ij")
    v = 0.5 + 0.5 * torch.sin(2 * math.pi * (x / 37.0 + y / 23.0 + z / 51.0))
    return v.view(1, C, D, H, W).expand(B, C, D, H, W).contiguous().half()


def fp32_top2_margins(scores):
    """top-1 minus top-2 logit of every step of a `generate(..., output_scores=True)` run (batch 1)."""
    out = []
    for sc in scores:
        t = sc[0].float().topk(2).values
        out.append(float(t[0] - t[1]))
    return out


def compare_greedy_ids(got, want, margins, thr):
    """got / want: id sequences of the run under test and of the fp32 reference; margins: the reference's own top-2 margins
    per step; thr: what a legitimate bf16 run may move a logit difference by.  Steps whose margin exceeds thr MUST agree and
    are counted; a step below it is skipped while the ids still agree -- and ends the comparison if they do not (the
    later steps then see another prefix).  Returns the number of steps actually asserted."""
    n = 0
    for t, (a, b) in enumerate(zip(got, want)):
        if margins[t] > thr:
            assert a == b, (t, list(got), list(want), margins, thr)
            n += 1
        elif a != b:
            break
    return n
