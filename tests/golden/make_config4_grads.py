"""Generate tests/golden/config4_grads.npz: the gradient of EVERY parameter of the path at BASELINE configs[3] size (stage-1
training, `src/train/train_stage1.py:244-251`): 12 ViT blocks on 8 chunks of (32,256,256) -> SPP -> 4-layer rma + DiffTS +
DMTP tokenizer at E = 4096 -> embedding splice, differentiated with torch.autograd over oracle/u2_oracle.py in fp32 (the
oracle's forward AND backward are pinned to the reference's own modules by tests/test_oracle_golden.py) and once more in
bf16 (the yardstick: what the reference's own bf16 arithmetic does to these gradients).

Run in the build container (`python tests/golden/make_config4_grads.py`, ~15 min on 8 cores, ~45 GB of host memory); the
GPU box never differentiates the oracle at this size (that would cost minutes of its budget per test run).  Parameters
and inputs are name-seeded (u2tokenizer_amd.synth), so the fixture stores, per parameter, the gradient's norm, a strided
sample of 1024 entries, and the bf16 run's relative error on that sample and on the norm.
"""
import sys
import time
import zlib
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import u2_oracle as O  # noqa: E402
from u2tokenizer_amd import synth  # noqa: E402

from cases import CONFIG4_CASE  # noqa: E402

NS_MAX = 1024


def sample_index(name: str, numel: int) -> torch.Tensor:
    n = min(numel, NS_MAX)
    stride = max(1, numel // n)
    return torch.arange(n) * stride + (zlib.crc32(name.encode()) % stride)


def path_state_dict(c, dtype, qk_gain=2.0):
    from test_gpu_configs import PathHolder, mm_config
    with torch.device("meta"):
        holder = PathHolder(mm_config(c["E"], c["image_size"]), c["vocab"])
    sd = {}
    for k, v in holder.state_dict().items():
        t = synth.synth_tensor("model." + k, v.shape, c["seed"])
        synth.lively_("model." + k, t, qk_gain=qk_gain)
        sd["model." + k] = t.to(dtype)
    return sd


def config4_inputs(c):
    """A structured volume (every 4 x 16 x 16 patch of the noise volume scaled by its own gain: CT volumes are not white
    noise, and distinct patches keep the ViT tokens distinct), ids, question ids and the loss weights."""
    vol = synth.synth_volume(1, 8, c["image_size"], seed=c["seed"], dtype=torch.float16)
    gain = torch.rand((8, 8, 1, 16, 1, 16, 1), generator=synth._gen(c["seed"], "volume_gain"))
    vol = (vol.float().view(8, 8, 4, 16, 16, 16, 16) * gain).view(1, 8, *c["image_size"]).to(torch.float16)
    ids = synth.synth_ids(1, c["S"], c["S"] - 24, c["vocab"], seed=c["seed"], name="input_ids")
    qids = synth.synth_ids(1, c["Lt"], 40, c["vocab"], seed=c["seed"], name="question_ids")
    G = synth.synth_tensor("grad_out", (1, c["S"], c["E"]), c["seed"])
    G[:, c["Q"] + 1:] = 0  # the loss looks at the first token and the spliced visual tokens only (plus the table rows
    return vol, ids, qids, G  # the question reaches through t_token)


SPP_BIAS = "model.mm_projector.projector.2.bias"


def centring_bias(c):
    """The projector's output bias minus the mean projector token of THIS input (fp32 oracle, no grad): the tokenizer then
    sees zero-mean visual tokens.  With the plain synthetic parameters every token shares one large common component, the
    attention of the residual-free SVR stack (svr.py:29,35) becomes query-independent and all tokens are identical after
    one layer; centred, the first two SVR layers carry token-dependent data (diversity 0.75 / 0.07) before the stack
    contracts (a residual-free softmax-averaging stack does, at any moderate gain: tests/test_gpu_configs.py pins layers
    2-3 by teacher forcing instead)."""
    from test_gpu_configs import mm_config, oracle_cfg
    vol = config4_inputs(c)[0]
    sd = path_state_dict(c, torch.float32)
    oc = oracle_cfg(mm_config(c["E"], c["image_size"]))
    with torch.no_grad():
        f = O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol.float().view(8, 1, *c["image_size"]), oc)
        f = O.spp_forward(sd, "model.mm_projector", f, oc)
    return (sd[SPP_BIAS] - f.reshape(-1, f.shape[-1]).mean(0)).detach()


def oracle_run(c, dtype, spp_bias):
    from test_gpu_configs import mm_config, oracle_cfg
    vol, ids, qids, G = config4_inputs(c)
    sd = path_state_dict(c, dtype)
    sd[SPP_BIAS] = spp_bias.detach().clone().to(dtype)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    oc = oracle_cfg(mm_config(c["E"], c["image_size"]))
    t0 = time.perf_counter()
    emb, _ = O.prepare_inputs_for_multimodal(sd, sd["model.embed_tokens.weight"], ids, vol.to(dtype), qids, oc)
    t1 = time.perf_counter()
    (emb.float() * G).sum().backward()
    t2 = time.perf_counter()
    print(f"{dtype}: forward {t1 - t0:.1f} s, backward {t2 - t1:.1f} s", flush=True)
    return emb.detach().float(), {k: v.grad for k, v in sd.items() if v.grad is not None}


def main():
    torch.set_num_threads(torch.get_num_threads())
    c = CONFIG4_CASE
    spp_bias = centring_bias(c)
    out32, g32 = oracle_run(c, torch.float32, spp_bias)
    names = sorted(g32)
    norms = np.array([g32[k].double().norm().item() for k in names])
    samples = np.zeros((len(names), NS_MAX), np.float32)
    for i, k in enumerate(names):
        idx = sample_index(k, g32[k].numel())
        samples[i, : idx.numel()] = g32[k].flatten()[idx].numpy()
    out16, g16 = oracle_run(c, torch.bfloat16, spp_bias)
    assert sorted(g16) == names
    # the metric of tests/test_gpu_backward.py::check_grads: rms(g - g32) / (rms(g32) + floor), floor = 2e-3 x the largest
    # per-tensor RMS (gradients that are tiny next to the largest one are compared against that scale)
    floor = 2e-3 * max(g32[k].double().pow(2).mean().sqrt().item() for k in names)
    rel16, dot, na, nb = [], 0.0, 0.0, 0.0
    for k in names:
        a, b = g32[k].double().flatten(), g16[k].double().flatten()
        rel16.append(((b - a).pow(2).mean().sqrt() / (a.pow(2).mean().sqrt() + floor)).item())
        dot, na, nb = dot + (a @ b).item(), na + (a @ a).item(), nb + (b @ b).item()
    rel16 = np.array(rel16)
    cos16 = dot / (na * nb) ** 0.5
    tok = slice(1, 1 + c["Q"])
    np.savez_compressed(Path(__file__).resolve().parent / "config4_grads.npz", names=np.array(names), norms=norms,
                        samples=samples, rel16=rel16, cos16=np.float64(cos16), spp_bias=spp_bias.numpy(), floor=np.float64(floor),
                        numels=np.array([g32[k].numel() for k in names]),
                        out_tokens_s16=out32[0, tok, ::16].numpy(),
                        out_rel16=np.float64(((out16 - out32)[0, tok].double().norm() / out32[0, tok].double().norm()).item()))
    print("wrote config4_grads.npz:", len(names), "gradients, bf16 cosine", cos16, "worst bf16 rel", rel16.max())


if __name__ == "__main__":
    main()
