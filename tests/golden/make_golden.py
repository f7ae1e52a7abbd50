"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference) on CPU.

Run in the build container only (`python tests/golden/make_golden.py`); /root/reference does not exist on the GPU
box, so the vectors are committed.  Parameters/inputs come from u2tokenizer_amd.synth (name-seeded), hence a
fixture stores just the config, the seeds and the reference OUTPUTS.

MONAI 1.3.0 (vit.py:19-20) is absent offline: the two blocks vit.py imports are provided by a stub whose forward is
oracle/u2_oracle.py's restatement -- the ViT vectors therefore pin the reference's ViT/ViT3DTower composition but
not MONAI itself ("parity unpinned", see oracle/u2_oracle.py header).
"""
import os
import sys
import types
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
from oracle import u2_oracle as O  # noqa: E402
from u2tokenizer_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent
torch.set_grad_enabled(False)


# ----------------------------------------------------------------------------- MONAI stub (parameter names = MONAI's)
def install_monai_stub():
    class PatchEmbeddingBlock(nn.Module):
        def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed, dropout_rate=0.0,
                     spatial_dims=3):
            super().__init__()
            self.patch_size = tuple(patch_size)
            n = 1
            for m, p in zip(img_size, patch_size):
                n *= m // p
            dim = in_channels * int(np.prod(patch_size))
            self.patch_embeddings = nn.Sequential(nn.Identity(), nn.Linear(dim, hidden_size))
            self.position_embeddings = nn.Parameter(torch.zeros(1, n, hidden_size))

        def forward(self, x):
            sd = {"p." + k: v for k, v in self.state_dict().items()}
            return O.patch_embedding_block(sd, "p", x, self.patch_size)

    class _SA(nn.Module):
        def __init__(self, h, heads, bias):
            super().__init__()
            self.out_proj = nn.Linear(h, h)
            self.qkv = nn.Linear(h, 3 * h, bias=bias)

    class _MLP(nn.Module):
        def __init__(self, h, m):
            super().__init__()
            self.linear1 = nn.Linear(h, m)
            self.linear2 = nn.Linear(m, h)

    class TransformerBlock(nn.Module):
        def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
            super().__init__()
            self.heads = num_heads
            self.mlp = _MLP(hidden_size, mlp_dim)
            self.norm1 = nn.LayerNorm(hidden_size)
            self.attn = _SA(hidden_size, num_heads, qkv_bias)
            self.norm2 = nn.LayerNorm(hidden_size)

        def forward(self, x):
            sd = {"b." + k: v for k, v in self.state_dict().items()}
            return O.transformer_block(sd, "b", x, self.heads)

    for name in ("monai", "monai.networks", "monai.networks.blocks", "monai.networks.blocks.patchembedding",
                 "monai.networks.blocks.transformerblock"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["monai.networks.blocks.patchembedding"].PatchEmbeddingBlock = PatchEmbeddingBlock
    sys.modules["monai.networks.blocks.transformerblock"].TransformerBlock = TransformerBlock


def fill(module, prefix, seed, lively=False):
    synth.fill_module_(module, seed=seed, prefix=prefix, lively=lively)


def save(name, **arrs):
    np.savez_compressed(OUT / f"{name}.npz", **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v))
                                                for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


# ----------------------------------------------------------------------------- cases (shared with tests/cases.py)
sys.path.insert(0, str(ROOT / "tests"))
from cases import TOKENIZER_CASES, SPP_CASES, VIT_CASES, FULL_CASES, tokenizer_inputs, spp_inputs  # noqa: E402


def gen_tokenizer(only=None):
    from src.model.u2tokenizer.u2Tokenizer import u2Tokenizer
    for name, c in TOKENIZER_CASES.items():
        if only and name not in only:
            continue
        m = u2Tokenizer(embed_size=c["E"], num_heads=c["heads"], num_layers=c["layers"], top_k=c["top_k"],
                        use_multi_scale=c["use_multi_scale"], num_3d_query_token=c["Q"], hidden_size=c["E"],
                        attn_type=c["attn_type"], enable_diffts=c["enable_diffts"], enable_dmtp=c["enable_dmtp"]).eval()
        fill(m, "u2tokenizer.", c["seed"], c.get("lively", False))
        v, t = tokenizer_inputs(c)
        out = m(v_token=v, t_token=t)
        extra = {}
        if not c["enable_diffts"]:
            # the reference's own (order-unspecified) top-k on its own fp32 scores, for the index gate
            x = m.svt_module.attention_network(v)
            sc = m.svt_module.token_selection.score_net(x).squeeze(-1).view(v.shape[0], -1)
            extra["ref_topk_idx"] = torch.topk(sc, c["top_k"], dim=1).indices
            extra["ref_scores"] = sc
        if c.get("lively"):  # record how much token-to-token variation the reference's SVR output carries
            x = m.svt_module.attention_network(v)
            x = x.reshape(-1, x.shape[-1])
            extra["svr_diversity"] = (x - x.mean(0, keepdim=True)).pow(2).mean().sqrt() / x.pow(2).mean().sqrt()
            # the same reference module in float64: selective softmaxes amplify fp32 summation-order noise to ~1e-4,
            # the fp64 run pins the ALGORITHM to 1e-10 (tests/test_oracle_golden.py compares the oracle in fp64)
            extra["out64"] = m.double()(v_token=v.double(), t_token=t.double())
            m.float()
        save(f"tokenizer_{name}", out=out, **extra)


GRAD_CASES = ("mu2_2l", "hard_2l_live", "rope_2l_live", "linvt_b2_live")


def grad_probe(name, shape, seed):
    """fixed direction a gradient is projected on (name-seeded, so the tests rebuild it)"""
    return synth.synth_tensor(name + "/probe", tuple(shape), seed).double()


def gen_grads():
    """Backward of the REFERENCE's own u2Tokenizer (float64, torch.autograd): d sum(out * G) / d (every parameter, v_token,
    t_token).  A fixture holds, per tensor, the gradient's norm and its projection on a name-seeded random direction (the
    full gradients of a 2-layer tokenizer would be ~50 MB), plus every 8th column of the two input gradients and their norms.  tests/test_oracle_golden.py
    checks torch.autograd over the ORACLE against them (1e-9), which pins the reference the GPU gradient tests compare with."""
    from src.model.u2tokenizer.u2Tokenizer import u2Tokenizer
    for name in GRAD_CASES:
        c = TOKENIZER_CASES[name]
        m = u2Tokenizer(embed_size=c["E"], num_heads=c["heads"], num_layers=c["layers"], top_k=c["top_k"],
                        use_multi_scale=c["use_multi_scale"], num_3d_query_token=c["Q"], hidden_size=c["E"],
                        attn_type=c["attn_type"], enable_diffts=c["enable_diffts"], enable_dmtp=c["enable_dmtp"]).eval()
        fill(m, "u2tokenizer.", c["seed"], c.get("lively", False))
        m = m.double()
        v, t = tokenizer_inputs(c)
        G = synth.synth_tensor("grad_out", (c["B"], c["Q"], c["E"]), c["seed"]).double()
        with torch.enable_grad():
            vin, tin = v.double().requires_grad_(True), t.double().requires_grad_(True)
            for p in m.parameters():
                p.requires_grad_(True)
            out = m(v_token=vin, t_token=tin)
            (out * G).sum().backward()
        names, norms, probes = [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            key = "u2tokenizer." + k
            names.append(key)
            norms.append(p.grad.norm().item())
            probes.append((p.grad * grad_probe(key, p.shape, c["seed"])).sum().item())
        save(f"tokenizer_{name}_grads", names=np.array(names), norms=np.array(norms, dtype=np.float64),
             probes=np.array(probes, dtype=np.float64), d_v_token_s8=vin.grad[..., ::8].float(),
             d_t_token_s8=tin.grad[..., ::8].float(), d_v_token_norm=vin.grad.norm().item(),
             d_t_token_norm=tin.grad.norm().item())


def gen_spp():
    from src.model.multimodal_projector.spatial_pooling_projector import SpatialPoolingProjector
    for name, c in SPP_CASES.items():
        m = SpatialPoolingProjector(image_size=c["image_size"], patch_size=c["patch_size"], in_dim=c["in_dim"],
                                    out_dim=c["E"], layer_type=c["layer_type"], layer_num=c["layer_num"],
                                    pooling_type=c["pooling_type"], pooling_size=c["pooling_size"]).eval()
        fill(m, "mm_projector.", c["seed"])
        out = m(spp_inputs(c))
        save(f"spp_{name}", out=out)


def gen_vit():
    install_monai_stub()
    from src.model.multimodal_encoder.vit import ViT3DTower
    for name, c in VIT_CASES.items():
        cfg = NS(vision_select_layer=-1, vision_select_feature=c["select_feature"], image_channel=1,
                 image_size=c["image_size"], patch_size=c["patch_size"])
        m = ViT3DTower(cfg).eval()
        fill(m, "vision_tower.", c["seed"])
        vol = synth.synth_volume(1, c["nchunk"], c["image_size"], seed=c["seed"], dtype=torch.float32)
        out = m(vol.view(c["nchunk"], 1, *c["image_size"]))
        save(f"vit_{name}", out=out)


def gen_full():
    """prepare_inputs_for_multimodal + first-step logits + greedy ids through the reference's u2LlamaForCausalLM."""
    install_monai_stub()
    from src.model.language_model.u2llama import u2LlamaForCausalLM, u2Config
    from src.model.u2tokenizer.builder import build_u2tokenizer_tower
    for name, c in FULL_CASES.items():
        cfg = u2Config(**c["llama"])
        for k, v in c["mm"].items():
            setattr(cfg, k, v)
        torch.manual_seed(0)
        m = u2LlamaForCausalLM(cfg).eval()
        m.model.u2tokenizer = build_u2tokenizer_tower(cfg)  # in-tree __init__ leaves it out (u2_arch.py:19)
        fill(m, "", c["seed"])
        if cfg.tie_word_embeddings:
            m.lm_head.weight = m.model.embed_tokens.weight
        B, C = c["B"], c["C"]
        vol = synth.synth_volume(B, C, c["mm"]["image_size"], seed=c["seed"], dtype=torch.float32)
        ids = synth.synth_ids(B, c["S"], c["n_real"], cfg.vocab_size, seed=c["seed"], name="input_ids")
        qids = synth.synth_ids(B, c["Lt"], c["n_q"], cfg.vocab_size, seed=c["seed"], name="question_ids")
        r = m.prepare_inputs_for_multimodal(ids, None, None, None, None, vol, qids)
        embeds = r[4]
        logits = m(images=vol, input_ids=ids, question_ids=qids).logits
        gen = m.generate(vol, ids, question_ids=qids, max_new_tokens=c["new_tokens"], do_sample=False)
        save(f"full_{name}", inputs_embeds=embeds, logits_last=logits[:, -1], greedy_ids=gen)


def gen_spp_grads():
    """Backward of the reference's SpatialPoolingProjector in float64: parameter gradients in full (they are small) and the
    input gradient's norm + a column sample."""
    from src.model.multimodal_projector.spatial_pooling_projector import SpatialPoolingProjector
    for name, c in SPP_CASES.items():
        m = SpatialPoolingProjector(image_size=c["image_size"], patch_size=c["patch_size"], in_dim=c["in_dim"],
                                    out_dim=c["E"], layer_type=c["layer_type"], layer_num=c["layer_num"],
                                    pooling_type=c["pooling_type"], pooling_size=c["pooling_size"]).eval()
        fill(m, "mm_projector.", c["seed"])
        m = m.double()
        with torch.enable_grad():
            x = spp_inputs(c).double().requires_grad_(True)
            for p in m.parameters():
                p.requires_grad_(True)
            out = m(x)
            G = synth.synth_tensor("grad_out", tuple(out.shape), c["seed"]).double()
            (out * G).sum().backward()
        names, norms, probes = [], [], []
        for k, p in m.named_parameters():
            key = "mm_projector." + k
            names.append(key)
            norms.append(p.grad.norm().item())
            probes.append((p.grad * grad_probe(key, p.shape, c["seed"])).sum().item())
        save(f"spp_{name}_grads", names=np.array(names), norms=np.array(norms, dtype=np.float64),
             probes=np.array(probes, dtype=np.float64), d_x_s8=x.grad[..., ::8].float(), d_x_norm=x.grad.norm().item())


def gen_full_grads():
    """Stage-1 style training step through the reference's u2LlamaForCausalLM in float64 (train_stage1.py:244-251):
    loss = model(images, input_ids, labels, question_ids).loss, backward; norm + name-seeded projection of the gradient of
    every parameter of the PATH (vision tower, projector, tokenizer, embedding table) and the loss itself."""
    install_monai_stub()
    from src.model.language_model.u2llama import u2LlamaForCausalLM, u2Config
    from src.model.u2tokenizer.builder import build_u2tokenizer_tower
    for name, c in FULL_CASES.items():
        cfg = u2Config(**c["llama"])
        for k, v in c["mm"].items():
            setattr(cfg, k, v)
        torch.manual_seed(0)
        m = u2LlamaForCausalLM(cfg).eval()
        m.model.u2tokenizer = build_u2tokenizer_tower(cfg)
        fill(m, "", c["seed"])
        m = m.double()
        vol = synth.synth_volume(c["B"], c["C"], c["mm"]["image_size"], seed=c["seed"], dtype=torch.float32).double()
        ids = synth.synth_ids(c["B"], c["S"], c["n_real"], cfg.vocab_size, seed=c["seed"], name="input_ids")
        qids = synth.synth_ids(c["B"], c["Lt"], c["n_q"], cfg.vocab_size, seed=c["seed"], name="question_ids")
        labels = ids.clone()
        labels[:, :20] = -100
        with torch.enable_grad():
            for p in m.parameters():
                p.requires_grad_(True)
            loss = m(images=vol, input_ids=ids, labels=labels, question_ids=qids).loss
            loss.backward()
        names, norms, probes = [], [], []
        for k, p in m.named_parameters():
            if p.grad is None or not any(s in k for s in ("vision_tower", "mm_projector", "u2tokenizer", "embed_tokens")):
                continue
            names.append(k)
            norms.append(p.grad.norm().item())
            probes.append((p.grad * grad_probe(k, p.shape, c["seed"])).sum().item())
        save(f"full_{name}_grads", names=np.array(names), norms=np.array(norms, dtype=np.float64),
             probes=np.array(probes, dtype=np.float64), loss=loss.item())


def gen_keys():
    """state_dict key / shape contract of the reference's u2Tokenizer variants (merged into state_dict_keys.json)."""
    import json
    from src.model.u2tokenizer.u2Tokenizer import u2Tokenizer
    p = OUT / "state_dict_keys.json"
    ref = json.loads(p.read_text())
    m = u2Tokenizer(embed_size=512, num_heads=8, num_layers=1, top_k=16, use_multi_scale=True, num_3d_query_token=16,
                    hidden_size=512, attn_type="linvt", enable_diffts=True, enable_dmtp=True)
    ref["linvt_small"] = {"u2tokenizer." + k: list(v.shape) for k, v in m.state_dict().items()}
    p.write_text(json.dumps(ref, indent=0, sort_keys=True))


if __name__ == "__main__":
    which = sys.argv[1:] or ["tokenizer", "spp", "vit", "full"]
    for w in which:
        if w.startswith("tokenizer:"):  # tokenizer:case1,case2 -> only those cases
            gen_tokenizer(set(w.split(":", 1)[1].split(",")))
        else:
            globals()["gen_" + w]()
