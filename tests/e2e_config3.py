"""Reference side of tests/test_gpu_configs.py::test_config3_end_to_end_first_step_logits_and_greedy_ids: the model, the
inputs and the HOST reference (oracle path in fp32 and in bf16 for two volumes + the same HF decoder on the CPU).

The reference is four full-size oracle passes and four decoder runs on the host: 2 minutes on a pool box with a fast host, 5+
on a slow one -- a quarter of the GPU suite's wall time spent re-deriving numbers that depend on nothing but seeds.  What is
fp32, and the two flip thresholds, are therefore committed as a fixture, tests/golden/config3_e2e_ref.npz, made by
tests/golden/make_config3_e2e.py WITH THIS MODULE (reference_live below is the one definition; the maker checks that what `load`
rebuilds is bit-equal to what reference_live returned, and that ids / margins equal what a GPU box's host computed live).  The
bf16 run that serves as the YARDSTICK of the distances (embeddings and logits of the benchmark's volume) stays live, on the host
the test runs on -- bf16 kernels differ between CPU generations, and "the reference's own bf16 run" should be one run, next to
the HIP run, not a file (bf16_noise_run: one oracle pass + one decoder forward).  The test takes the fixture when its header
matches AND the file's oracle fingerprint is the current oracle's (oracle_fingerprint below), and recomputes everything live
otherwise, or when U2_LIVE_ORACLE=1 asks for it.  Nothing here touches the GPU."""
import os
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch

from helpers import decisive_decoder_, fp32_top2_margins, smooth_volume
from oracle import u2_oracle as O
from u2tokenizer_amd import synth

bf = torch.bfloat16
E, VOCAB, S, LT, SEED, DSEED, NEW = 4096, 4096, 1024, 1024, 75, 0, 4
NQ = 256                     # visual tokens spliced behind position 0 (num_3d_query_token)
FIXTURE = Path(__file__).resolve().parent / "golden" / "config3_e2e_ref.npz"
FORMAT = 2                   # 2: the file carries a behavioural fingerprint of the oracle it was made with (oracle_fingerprint)


def _header():
    return np.array([FORMAT, E, VOCAB, S, LT, SEED, DSEED, NEW], dtype=np.int64)


def oracle_fingerprint():
    """What ties the committed file to the oracle's CODE (seeds and sizes alone would let an edited oracle keep
    a stale file): the oracle's whole path -- ViT, projector, 2-layer rma + DiffTS + DMTP + multi-scale tokenizer, splice -- run at a
    tiny size on name-seeded lively parameters; returned are the norm of the spliced visual rows and their projections on eight
    name-seeded directions (float64[9]).  The maker stores it; `load` recomputes it (a second of host time) and refuses the file when it
    differs, tests/test_e2e_fixture.py fails the CPU suite for it.  A refactoring that keeps the arithmetic keeps the fingerprint."""
    from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower
    E_, T_, vocab, seed = 64, 4, 128, 7
    c = dict(vision_tower="vit3d", image_channel=1, image_size=[8, 64, 64], patch_size=[4, 16, 16], vision_select_layer=-1,
             vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp", proj_layer_num=2, proj_pooling_type="spatial",
             proj_pooling_size=2, mm_hidden_size=768, hidden_size=E_, enable_u2tokenizer=True, u2t_num_heads=4, u2t_num_layers=2,
             u2t_top_k=8, use_multi_scale=True, num_3d_query_token=8, attn_type="rma", enable_diffts=True, enable_dmtp=True)
    cfg = NS(**c)
    with torch.device("meta"):
        mods = {"vision_tower": build_vision_tower(cfg), "mm_projector": build_mm_projector(cfg), "u2tokenizer": build_u2tokenizer_tower(cfg)}
    sd = {"model.embed_tokens.weight": synth.synth_tensor("model.embed_tokens.weight", (vocab, E_), seed)}
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            if v.is_floating_point():
                key = f"model.{name}.{k}"
                sd[key] = synth.lively_(key, synth.synth_tensor(key, v.shape, seed))
    oc = O.PathConfig(**{k: c[k] for k in ("image_size", "patch_size", "vision_select_feature", "proj_layer_type", "proj_layer_num",
                                             "proj_pooling_type", "proj_pooling_size", "hidden_size", "u2t_num_heads", "u2t_num_layers",
                                             "u2t_top_k", "use_multi_scale", "num_3d_query_token", "attn_type", "enable_diffts",
                                             "enable_dmtp", "enable_u2tokenizer")})
    vol = synth.synth_volume(1, T_, c["image_size"], seed=seed, dtype=torch.float32)
    ids = synth.synth_ids(1, 24, 20, vocab, seed=seed, name="input_ids")
    qids = synth.synth_ids(1, 12, 6, vocab, seed=seed, name="question_ids")
    with torch.no_grad():
        emb, _ = O.prepare_inputs_for_multimodal(sd, sd["model.embed_tokens.weight"], ids, vol, qids, oc)
    vis = emb[0, 1:1 + c["num_3d_query_token"]].double()
    probes = [float((vis * synth.synth_tensor(f"fingerprint_probe_{i}", tuple(vis.shape), seed).double()).sum()) for i in range(8)]
    return np.array([float(vis.norm())] + probes, dtype=np.float64)


def fingerprint_matches(stored, live=None, rtol=2e-4):
    """fp32 summation order differs between hosts by ~1e-6 of these sums; any change of the arithmetic moves them by percents."""
    live = oracle_fingerprint() if live is None else live
    stored = np.asarray(stored, dtype=np.float64)
    return stored.shape == live.shape and bool(np.all(np.abs(stored - live) <= rtol * np.abs(live[0])))


def setup(mm_config, oracle_cfg):
    """The model (fp32, on the host, decisive decoder) and the inputs; mm_config / oracle_cfg: the test module's own builders."""
    from u2tokenizer_amd.language_model import u2Qwen3Config, u2Qwen3ForCausalLM
    c = mm_config(E, [32, 256, 256])
    cfg = u2Qwen3Config(vocab_size=VOCAB, hidden_size=E, intermediate_size=12288, num_hidden_layers=4,
                        num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048,
                        tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    for k, v in c.items():
        if k != "hidden_size":
            setattr(cfg, k, v)
    m = u2Qwen3ForCausalLM(cfg).eval()
    synth.fill_module_(m, seed=SEED, lively=True)
    decisive_decoder_(m, DSEED)
    vols = {"noise": synth.synth_volume(1, 8, c["image_size"], seed=SEED, dtype=torch.float16),
            "smooth": smooth_volume(1, 8, c["image_size"])}
    ids = synth.synth_ids(1, S, S - 24, VOCAB, seed=SEED, name="input_ids")
    qids = synth.synth_ids(1, LT, 40, VOCAB, seed=SEED, name="question_ids")
    return NS(m=m, c=c, vols=vols, ids=ids, qids=qids, oc=oracle_cfg(c))


def _greedy_reference(m, e32, new):
    """fp32 reference on the host: greedy ids, the top-2 margin of every decision, the first-step logits."""
    from transformers import Qwen3ForCausalLM
    g = Qwen3ForCausalLM.generate(m, inputs_embeds=e32, max_new_tokens=new, do_sample=False, output_scores=True,
                                  return_dict_in_generate=True)
    return g.sequences[0].tolist(), fp32_top2_margins(g.scores), g.scores[0][0].float()


def _bf16_run(sd16, m16, s, vol):
    """The reference's own bf16 run of one volume: spliced embeddings (bf16) and last-position logits (float)."""
    e16, _ = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], s.ids, vol.to(bf), s.qids, s.oc)
    return e16, m16(inputs_embeds=e16).logits[0, -1].float()


@torch.no_grad()
def bf16_noise_run(s):
    """{e16_noise, logits16}: the bf16 yardstick of the benchmark's volume, computed now, on this host.  Converts s.m to bf16
    in place (the test moves that very module to the GPU afterwards)."""
    m16 = s.m.to(bf)
    sd16 = {k: v for k, v in m16.state_dict().items() if v.is_floating_point()}
    e16, l16 = _bf16_run(sd16, m16, s, s.vols["noise"])
    return {"e16_noise": e16, "logits16": l16[None]}


@torch.no_grad()
def reference_live(s):
    """The whole host reference, computed now.  Leaves s.m converted to bf16 (the test moves that very module to the GPU).
    Returns {e32_noise, e16_noise, logits32, logits16, refs {volume: (ids, margins, None)}, thrs {volume: 4 x the largest logit
    deviation of the reference's own bf16 run}, aligned_rel_rms (how far the second volume moves the visual tokens)}."""
    m = s.m
    sd32 = {k: v.clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    e32, refs, first = {}, {}, {}
    for v, vol in s.vols.items():
        e32[v], _ = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], s.ids, vol.float(), s.qids, s.oc)
        ids32, margins, first[v] = _greedy_reference(m, e32[v], NEW)
        refs[v] = (ids32, margins, None)
    logits32 = m(inputs_embeds=e32["noise"]).logits[:, -1]
    del sd32
    m16 = m.to(bf)
    sd16 = {k: v for k, v in m16.state_dict().items() if v.is_floating_point()}
    thrs, e16n, logits16 = {}, None, None
    for v, vol in s.vols.items():
        e16, l16 = _bf16_run(sd16, m16, s, vol)
        thrs[v] = 4 * float((l16 - first[v]).abs().max())
        if v == "noise":
            e16n, logits16 = e16, l16[None]
    tn, ts = e32["noise"][0, 1:1 + NQ], e32["smooth"][0, 1:1 + NQ]
    return {"e32_noise": e32["noise"], "e16_noise": e16n, "logits32": logits32.float(), "logits16": logits16, "refs": refs,
            "thrs": thrs, "aligned_rel_rms": float((tn - ts).pow(2).mean().sqrt() / tn.pow(2).mean().sqrt())}


def save(ref, path=FIXTURE):
    """Only what seeds do not give back, and only the fp32 side: the visual rows of the fp32 embeddings (the other rows are
    embedding-table rows of `ids`), the fp32 logit row, ids / margins / thresholds."""
    arrays = {"header": _header(), "oracle_fingerprint": oracle_fingerprint(),
              "e32_noise_vis": ref["e32_noise"][0, 1:1 + NQ].numpy(),
              "logits32": ref["logits32"].numpy(),
              "aligned_rel_rms": np.float64(ref["aligned_rel_rms"])}
    for v, (ids32, margins, _) in ref["refs"].items():
        arrays[f"ids_{v}"] = np.array(ids32, dtype=np.int64)
        arrays[f"margins_{v}"] = np.array(margins, dtype=np.float64)
        arrays[f"thr_{v}"] = np.float64(ref["thrs"][v])
    np.savez(path, **arrays)


def load(s, path=FIXTURE):
    """The fp32 side of the reference + the thresholds from the fixture ({e32_noise, logits32, refs, thrs, aligned_rel_rms}:
    bf16_noise_run supplies the rest), or None (absent / another format or seed / U2_LIVE_ORACLE=1).  s.m is read only (its
    embedding table supplies the text rows) and must still be in fp32: call this BEFORE converting it."""
    if os.environ.get("U2_LIVE_ORACLE") == "1" or not Path(path).exists():
        return None
    z = np.load(path)
    if "header" not in z.files or not np.array_equal(z["header"], _header()):
        return None
    if "oracle_fingerprint" not in z.files or not fingerprint_matches(z["oracle_fingerprint"]):
        return None                     # made with an oracle that computes something else: run the reference live
    table = s.m.get_input_embeddings().weight.detach()
    if table.dtype != torch.float32:
        return None
    emb = torch.nn.functional.embedding(s.ids, table)
    e32 = torch.cat((emb[:, :1], torch.from_numpy(z["e32_noise_vis"].copy())[None], emb[:, 1 + NQ:]), 1)
    refs = {v: (z[f"ids_{v}"].tolist(), z[f"margins_{v}"].tolist(), None) for v in s.vols}
    thrs = {v: float(z[f"thr_{v}"]) for v in s.vols}
    return {"e32_noise": e32, "logits32": torch.from_numpy(z["logits32"].copy()), "refs": refs, "thrs": thrs,
            "aligned_rel_rms": float(z["aligned_rel_rms"])}
