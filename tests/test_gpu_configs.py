"""GPU: the BASELINE.json configurations at their real sizes, HIP path vs the oracle run in fp32 AND in bf16 on the
host (same name-seeded "lively" parameters, same inputs), stage by stage.

For every stage the three distances SURVEY.md section 8(d) asks for are computed -- |hip - oracle_fp32|,
|hip - oracle_bf16|, |oracle_bf16 - oracle_fp32| -- gated with the bar of tests/test_gpu_path.py (the HIP path may be no
further from the fp32 reference than 1.5x the bf16 reference run is) and written to gpurun_out/r06_parity.json, from
where the round's copy under profiles/ is taken.
"""
import json
import os
import time
from pathlib import Path
from types import SimpleNamespace as NS

import pytest
import torch

from helpers import compare_greedy_ids, decisive_decoder_, diversity, err_stats, fp32_top2_margins, smooth_volume
from oracle import u2_oracle as O
from u2tokenizer_amd import synth

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
D = "cuda"
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from u2tokenizer_amd import ops
    ops.device_check()
    torch.set_grad_enabled(False)
    yield


from suite_budget import record  # noqa: E402  (merges {key: value} into gpurun_out/r06_parity.json)


def three_way(hip, o32, o16):
    return {"hip_vs_o32": err_stats(hip.float().cpu(), o32), "hip_vs_o16": err_stats(hip.float().cpu(), o16.float()),
            "o16_vs_o32": err_stats(o16.float(), o32), "diversity_o32": diversity(o32)}


def gate(d, what, chained=False):
    """The HIP path may be no further from the fp32 reference than the reference's own bf16 run is, with a margin for the
    different (equally legitimate) rounding points: 1.2 x its relative RMS distance + 2e-4 (round 3 gated 1.5 x + 1e-3;
    measured ratios over every stage and configuration: 0.45 .. 1.07, profiles/r03_parity.json, r04_parity.json), and
    1.6 x its largest absolute deviation + four bf16 ulps of the reference RMS.  north_star's literal "within 1e-3 of the
    bf16 reference" is reported beside it (err_stats: fraction of elements within 1e-3, distances in bf16 ulps): at these
    magnitudes 1e-3 is a fraction of ONE bf16 ulp, which two correct bf16 computations cannot promise each other.

    chained=True is the bar for outputs of the whole chain at config 3 with lively 4-layer parameters (round 3's 1.5 x +
    1e-3): there the residual-free tokenizer amplifies WHICH bf16 rounding the ViT attention made -- three correct HIP
    orderings of the same arithmetic (double-pipeline flash loop / 128-row flash loop / unfused attention) land at 1.31 x,
    1.01 x and 0.81 x the bf16 reference's distance on the same inputs and are 0.017 .. 0.020 apart from one another
    (profiles/r04_e2e_rounding_spread.log; the test below records the same three)."""
    e_hip, e_orc = d["hip_vs_o32"], d["o16_vs_o32"]
    k, eps = (1.5, 1e-3) if chained else (1.2, 2e-4)
    assert e_hip["rel_rms"] <= k * e_orc["rel_rms"] + eps, (what, e_hip, e_orc)
    assert e_hip["max_abs"] <= 1.6 * e_orc["max_abs"] + 2.0 ** -8 * e_hip["ref_rms"] * 4, (what, e_hip, e_orc)
    # and against the bf16 reference itself: as close to it as it is to fp32 (the two differ by rounding, not by a bug)
    assert d["hip_vs_o16"]["rel_rms"] <= 1.5 * e_orc["rel_rms"] + 2e-4, (what, d["hip_vs_o16"], e_orc)


def mm_config(E, image_size, **kw):
    c = dict(vision_tower="vit3d", image_channel=1, image_size=image_size, patch_size=[4, 16, 16],
             vision_select_layer=-1, vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp",
             proj_layer_num=2, proj_pooling_type="spatial", proj_pooling_size=2, mm_hidden_size=768, hidden_size=E,
             enable_u2tokenizer=True, u2t_num_heads=8, u2t_num_layers=4, u2t_top_k=1024, use_multi_scale=True,
             num_3d_query_token=256, attn_type="rma", enable_diffts=True, enable_dmtp=True)
    c.update(kw)
    return c


def oracle_cfg(c):
    return O.PathConfig(image_size=c["image_size"], patch_size=c["patch_size"],
                        vision_select_feature=c["vision_select_feature"], proj_layer_type=c["proj_layer_type"],
                        proj_layer_num=c["proj_layer_num"], proj_pooling_type=c["proj_pooling_type"],
                        proj_pooling_size=c["proj_pooling_size"], hidden_size=c["hidden_size"],
                        u2t_num_heads=c["u2t_num_heads"], u2t_num_layers=c["u2t_num_layers"], u2t_top_k=c["u2t_top_k"],
                        use_multi_scale=c["use_multi_scale"], num_3d_query_token=c["num_3d_query_token"],
                        attn_type=c["attn_type"], enable_diffts=c["enable_diffts"], enable_dmtp=c["enable_dmtp"],
                        enable_u2tokenizer=c["enable_u2tokenizer"])


class PathHolder(torch.nn.Module):
    """vision tower + projector (+ tokenizer) + embedding table under the reference's attribute names."""

    def __init__(self, c, vocab):
        super().__init__()
        from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower
        cfg = NS(**c)
        self.vision_tower = build_vision_tower(cfg)
        self.mm_projector = build_mm_projector(cfg)
        if c["enable_u2tokenizer"]:
            self.u2tokenizer = build_u2tokenizer_tower(cfg)
        self.embed_tokens = torch.nn.Embedding(vocab, c["hidden_size"])

    def get_vision_tower(self):
        return self.vision_tower

    def get_u2tokenizer(self):
        return getattr(self, "u2tokenizer", None)


def build_path(c, vocab, seed):
    """Returns (path object with prepare_inputs_for_multimodal on the GPU in bf16, fp32 state dict, bf16 state dict).
    The modules are built on the meta device and filled on the GPU from the name-seeded state dict, so the host only
    ever holds the two state dicts."""
    from u2tokenizer_amd.arch import u2MetaForCausalLM

    class PathOnly(u2MetaForCausalLM):
        def __init__(self, holder):
            self.holder, self.config = holder, NS(**c)

        def get_model(self):
            return self.holder

    with torch.device("meta"):
        holder = PathHolder(c, vocab)
    sd32, sd16 = {}, {}
    for k, v in holder.state_dict().items():
        t = synth.synth_tensor("model." + k, v.shape, seed)
        synth.lively_("model." + k, t)
        sd32["model." + k], sd16["model." + k] = t, t.to(bf)
    holder = holder.to(bf).to_empty(device=D)
    holder.load_state_dict({k[len("model."):]: v for k, v in sd16.items()}, strict=True)
    for p in holder.parameters():
        p.requires_grad_(False)
    return PathOnly(holder), sd32, sd16


def staged_oracle(sd, dt, vol, ids, qids, oc):
    """Stage outputs of the oracle: vit, spp, tokenizer, inputs_embeds (+ wall time per stage)."""
    B, C = vol.shape[:2]
    t, out = {}, {}
    t0 = time.perf_counter()
    out["vit"] = O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol.to(dt).view(B * C, 1, *vol.shape[2:]), oc)
    t["vit"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    out["spp"] = O.spp_forward(sd, "model.mm_projector", out["vit"], oc)
    t["spp"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    emb_w = sd["model.embed_tokens.weight"]
    tt = torch.nn.functional.embedding(qids, emb_w)
    out["tokenizer"], idx = O.tokenizer_forward(sd, "model.u2tokenizer", out["spp"].view(B, C, -1, out["spp"].shape[-1]),
                                                tt, oc)
    t["tokenizer"] = time.perf_counter() - t0
    emb = torch.nn.functional.embedding(ids, emb_w)
    out["inputs_embeds"] = torch.cat((emb[:, :1], out["tokenizer"], emb[:, out["tokenizer"].shape[1] + 1:]), 1)
    return out, t, idx


def staged_hip(path, vol, ids, qids):
    from u2tokenizer_amd import ops
    h = path.holder
    B, C = vol.shape[:2]
    out = {}
    out["vit"] = h.vision_tower(vol.to(D).view(B * C, 1, *vol.shape[2:]))
    out["spp"] = h.mm_projector(out["vit"])
    tt = ops.embed_splice(h.embed_tokens.weight, qids.to(D))
    out["tokenizer"] = h.u2tokenizer(v_token=out["spp"].view(B, C, -1, out["spp"].shape[-1]), t_token=tt)
    out["inputs_embeds"] = path.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4]
    # the product entry point runs the same four stages: its tokens must be the staged ones, bit for bit
    assert torch.equal(out["inputs_embeds"][:, 1:1 + out["tokenizer"].shape[1]], out["tokenizer"])
    return out


def run_full_config(name, c, B, C, S, Lt, seed, vocab=4096):
    path, sd32, sd16 = build_path(c, vocab, seed)
    vol = synth.synth_volume(B, C, c["image_size"], seed=seed, dtype=torch.float16)
    ids = synth.synth_ids(B, S, S - 24, vocab, seed=seed, name="input_ids")
    qids = synth.synth_ids(B, Lt, 40, vocab, seed=seed, name="question_ids")
    oc = oracle_cfg(c)
    o32, t32, _ = staged_oracle(sd32, torch.float32, vol, ids, qids, oc)
    o16, t16, _ = staged_oracle(sd16, bf, vol, ids, qids, oc)
    hip = staged_hip(path, vol, ids, qids)
    rep = {"config": {k: c[k] for k in ("hidden_size", "image_size", "u2t_num_layers", "u2t_top_k", "enable_diffts",
                                        "enable_dmtp", "use_multi_scale", "num_3d_query_token")},
           "B": B, "chunks": C, "prompt": S, "text": Lt, "oracle_seconds_fp32": t32, "oracle_seconds_bf16": t16,
           "host_threads": torch.get_num_threads(), "stages": {}}
    for st in ("vit", "spp", "tokenizer", "inputs_embeds"):
        rep["stages"][st] = three_way(hip[st], o32[st], o16[st])
    # The bf16 yardstick's DiffTS in the REFERENCE's own bf16 arithmetic (svr.py:112-115 rounds every w * x product to bf16; the oracle's
    # default is one matmul, which is more accurate -- VERDICT r5 weak #6): where the 1024-head loop is affordable on the host, the
    # tokenizer stage of the bf16 run is redone in that form and both distances are reported; the gate below keeps the matmul form,
    # whose distance is the smaller (stricter) denominator.
    spp16 = o16["spp"].view(B, C, -1, o16["spp"].shape[-1])
    if c["enable_diffts"] and B * C * spp16.shape[2] * c["u2t_top_k"] * c["hidden_size"] <= 5e9:
        import dataclasses
        tt = torch.nn.functional.embedding(qids, sd16["model.embed_tokens.weight"])
        t0 = time.perf_counter()
        o16l, _ = O.tokenizer_forward(sd16, "model.u2tokenizer", spp16, tt, dataclasses.replace(oc, diffts_loop_form=True))
        rep["stages"]["tokenizer"]["diffts_loop_form_yardstick"] = {
            "o16loop_vs_o32": err_stats(o16l.float(), o32["tokenizer"]), "o16loop_vs_o16": err_stats(o16l.float(), o16["tokenizer"].float()),
            "hip_vs_o16loop": err_stats(hip["tokenizer"].float().cpu(), o16l.float()), "seconds": time.perf_counter() - t0}
    record(name, rep)
    for st in rep["stages"]:
        assert torch.isfinite(hip[st].float()).all(), st
        gate(rep["stages"][st], f"{name}:{st}")
    return rep


# NOTE on what the chained tests can and cannot see.  The SVR is a stack of attention layers without residuals or norms
# (svr.py:29,35).  With random weights it is either contractive -- the ViT features of a noise volume share a large
# common component, attention becomes query-independent and the 4-layer chain reaches the aggregation stages with nearly
# identical tokens (token diversity ~0 at the tokenizer output, recorded per stage) -- or, with query / key gains large
# enough to keep it selective, chaotic: the reference's own bf16 run then differs from its fp32 run by 50-140 % after
# four layers (measured: profiles/r02_parity.json, tests/cases.py "live" cases), and hard top-k indices agree at chance
# level.  The chained tests below therefore pin scales / biases / layouts / the ViT tightly (errors ~1e-2, HIP <= the
# bf16 reference's own error), and token-dependent data flow is pinned by test_tokenizer_one_layer_full_width: ONE
# selective layer at full width, where bf16 noise is 5-20 % and any indexing mistake is 100 %.


@pytest.mark.host_heavy(62)   # nominal seconds, mostly host (tests/suite_budget.py)
def test_config3_full_path_vs_oracle():
    """BASELINE configs[2] -- the benchmark's configuration: u2Qwen3-8B shape (E = 4096), one 256^3 volume = 8 chunks
    of (32,256,256) fp16, ViT-B x12, SPP, 4-layer rma + DiffTS(1024) + DMTP tokenizer, 256 queries, text 1024, prompt
    1024 (u2_arch.py:96-117)."""
    c = mm_config(4096, [32, 256, 256])
    run_full_config("config3_E4096_256cube", c, B=1, C=8, S=1024, Lt=1024, seed=71)


from e2e_config3 import _greedy_reference  # noqa: E402  (one definition: the fixture maker uses it too)


def _id_gate(rep, vols_hip_ids, refs, thrs, need=3):
    """refs[v] = (ids32, margins, _), thrs[v] = 4 x the largest logit deviation of the reference's own bf16 run, vols_hip_ids[v] =
    the ids under test.  Asserted, in this order: the REFERENCE decides clearly (>= `need` steps above the threshold for every
    volume), its ids depend on the image (the two volumes part at a step both decide clearly) and are not one id repeated;
    then the run under test reproduces every clearly decided step -- at least `need` per volume are actually compared."""
    for v, (ids32, margins, _) in refs.items():
        clear = [mg > thrs[v] for mg in margins]
        rep[f"{v}:fp32_ids"], rep[f"{v}:fp32_top2_margins"], rep[f"{v}:flip_threshold"] = ids32, margins, thrs[v]
        assert sum(clear) >= need, (v, margins, thrs[v])
    (ia, ma, _), (ib, mb, _) = refs["noise"], refs["smooth"]
    part = [t for t in range(len(ia)) if ia[t] != ib[t] and ma[t] > thrs["noise"] and mb[t] > thrs["smooth"]]
    assert part, ("the second volume does not change a clearly decided id", ia, ib, ma, mb)
    assert len(set(ia)) > 1 or len(set(ib)) > 1, (ia, ib)                # not one id for ever
    for v, (ids32, margins, _) in refs.items():
        n = compare_greedy_ids(vols_hip_ids[v], ids32, margins, thrs[v])
        rep[f"{v}:hip_ids"], rep[f"{v}:steps_compared"] = list(vols_hip_ids[v]), n
        assert n >= need, (v, vols_hip_ids[v], ids32, margins, thrs[v])
    assert vols_hip_ids["noise"][part[0]] != vols_hip_ids["smooth"][part[0]]   # ... and the HIP path follows the image


@pytest.mark.host_heavy(65)   # nominal seconds, mostly host (tests/suite_budget.py)
def test_config3_end_to_end_first_step_logits_and_greedy_ids():
    """SURVEY 8(d) at the benchmark's configuration, end to end (u2llama.py:76-87,123-126): one 256^3 volume through the HIP
    ViT / SPP / 4-layer tokenizer, spliced into 1024 embeddings, then a Qwen3-8B-WIDTH decoder (hidden 4096, 32 / 8 heads of
    128, MLP 12288; 4 layers) through the fused HIP prefill + decode steps -- first-step logits and 4 greedy ids against the
    oracle path + the same HF decoder in fp32 on the host, with the reference's own bf16 run as the yardstick.

    Round 5: the decoder is drawn so that the reference's decisions are clear (helpers.decisive_decoder_) and the
    ids are compared for TWO volumes -- the benchmark's noise volume and a smooth one whose tokens differ by ~80 % -- whose
    reference ids must differ: a path that ignored the image could not pass."""
    import e2e_config3 as R
    # The HOST reference is four full-size oracle passes + four decoder runs on the CPU (fp32 and bf16, both volumes).  Its fp32
    # side and the two flip thresholds depend on seeds only: tests/golden/config3_e2e_ref.npz holds them (made by
    # tests/golden/make_config3_e2e.py from R.reference_live -- the definition used here when the fixture is absent or
    # U2_LIVE_ORACLE=1 asks for the whole live run, 2 to 5 minutes of host time).  The bf16 run the distances are measured
    # against stays live, on this host, next to the HIP run (R.bf16_noise_run: one oracle pass + one decoder forward).
    s = R.setup(mm_config, oracle_cfg)
    ref = R.load(s)
    reference_source = "fp32 side + thresholds: tests/golden/config3_e2e_ref.npz; bf16 yardstick: live on this host"
    if ref is None:
        ref, reference_source = R.reference_live(s), "computed live on this host"
    else:
        ref.update(R.bf16_noise_run(s))
    vols, ids, qids, new = s.vols, s.ids, s.qids, R.NEW
    e32n, e16n, logits32, logits16, refs, thrs = (ref[k] for k in ("e32_noise", "e16_noise", "logits32", "logits16", "refs", "thrs"))
    m16 = s.m.to(bf)
    mg = m16.to(D)
    vol = vols["noise"]
    out = mg(images=vol.to(D), input_ids=ids.to(D), question_ids=qids.to(D))
    assert hasattr(mg.model.layers[0], "_u2_prefill")                      # the decoder ran through the fused HIP layers
    emb = mg.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4]
    gen = {v: mg.generate(x.to(D), ids.to(D), question_ids=qids.to(D), max_new_tokens=new, do_sample=False).cpu()[0].tolist()
           for v, x in vols.items()}
    # how far apart equally correct bf16 orderings of the ViT attention land after the chain (see gate(chained=True))
    from u2tokenizer_amd import ops
    spread = {"flash_double_pipeline (default)": {"vs_o32": err_stats(emb.float().cpu(), e32n)["rel_rms"]}}
    for name, opt, val, back in (("flash_128_row_units", "flash_mode", 1, 0), ("unfused_attention", "vit_flash", 0, 1)):
        ops.set_option(opt, val)
        try:
            mg.get_model().get_vision_tower().invalidate_feature_cache()
            alt = mg.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4].float().cpu()
        finally:
            ops.set_option(opt, back)
            mg.get_model().get_vision_tower().invalidate_feature_cache()
        spread[name] = {"vs_o32": err_stats(alt, e32n)["rel_rms"], "vs_default": err_stats(alt, emb.float().cpu())["rel_rms"]}
    rep = {"hip_rounding_spread_inputs_embeds": spread, "host_reference": reference_source, "decoder": "Qwen3-8B width (4096 / 12288, 32 q / 8 kv heads of 128), 4 layers, "
           "helpers.decisive_decoder_(dseed 0); fused HIP prefill + decode",
           "inputs_embeds": three_way(emb, e32n, e16n), "logits_last": three_way(out.logits[:, -1], logits32, logits16),
           "aligned_tokens_smooth_vs_noise_rel_rms": ref["aligned_rel_rms"]}
    try:
        _id_gate(rep, gen, refs, thrs)
    finally:
        record("config3_E4096_256cube_end_to_end", rep)
    gate(rep["inputs_embeds"], "config3 e2e inputs_embeds", chained=True)
    gate(rep["logits_last"], "config3 e2e logits", chained=True)


@pytest.mark.host_heavy(15)   # nominal seconds, mostly host (tests/suite_budget.py)
def test_float16_model_under_autocast():
    """evalscipt/ourmodel_amos.py:33,70: the whole model in float16, generate under torch.autocast.  Round 5: the path modules
    run the IEEE-half build of the library on the fp16 parameters themselves (fp32 accumulation; round 4 computed through a
    bf16 copy and landed 17 x further from fp32 than the reference's own fp16 run); the decoder's prefill / decode steps run the
    fused HIP layers of the same build.  Against
    the oracle + HF decoder in fp32 on the SAME (fp16-representable) weights: the spliced embeddings must be as close to fp32
    as the reference's own FLOAT16 run is (1.2 x its relative RMS distance; the bf16 run's distance is recorded beside it),
    greedy ids as in the bf16 test."""
    from transformers import Qwen3ForCausalLM
    from u2tokenizer_amd.language_model import u2Qwen3Config, u2Qwen3ForCausalLM
    E, vocab, S, Lt, seed = 2048, 4096, 320, 1024, 73
    c = mm_config(E, [32, 64, 64], u2t_num_layers=1, u2t_top_k=16, use_multi_scale=False, enable_diffts=False,
                  enable_dmtp=False)
    cfg = u2Qwen3Config(vocab_size=vocab, hidden_size=E, intermediate_size=6144, num_hidden_layers=2,
                        num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048,
                        tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    for k, v in c.items():
        if k != "hidden_size":
            setattr(cfg, k, v)
    m = u2Qwen3ForCausalLM(cfg).eval()
    synth.fill_module_(m, seed=seed, lively=True)
    decisive_decoder_(m, 0)                                # (as test_config1_survey_size_through_qwen3)
    m = m.half().float()                                   # weights that fp16 holds exactly
    sd32 = {k: v.clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    vol = synth.synth_volume(1, 2, c["image_size"], seed=seed, dtype=torch.float16)
    ids = synth.synth_ids(1, S, S - 8, vocab, seed=seed, name="input_ids")
    qids = synth.synth_ids(1, Lt, 40, vocab, seed=seed, name="question_ids")
    oc = oracle_cfg(c)
    e32, _ = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], ids, vol.float(), qids, oc)
    gen32 = Qwen3ForCausalLM.generate(m, inputs_embeds=e32, max_new_tokens=4, do_sample=False, output_scores=True,
                                      return_dict_in_generate=True)
    logits32 = m(inputs_embeds=e32).logits[:, -1]
    sd16 = {k: v.to(bf) for k, v in sd32.items()}
    e16, _ = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], ids, vol.to(bf), qids, oc)
    import copy
    l16 = copy.deepcopy(m).to(bf)(inputs_embeds=e16).logits[0, -1].float()
    thr = 4 * float((l16 - logits32[0]).abs().max())          # (the bf16 reference's noise: the yardstick of the other id gates)
    sdh = {k: v.half() for k, v in sd32.items()}
    eh, _ = O.prepare_inputs_for_multimodal(sdh, sdh["model.embed_tokens.weight"], ids, vol.half(), qids, oc)
    mg = m.half().to(D)
    assert next(mg.get_model().get_u2tokenizer().parameters()).dtype == torch.float16
    with torch.autocast("cuda", dtype=torch.float16):
        emb = mg.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4]
        out = mg(images=vol.to(D), input_ids=ids.to(D), question_ids=qids.to(D))
        gen = mg.generate(vol.to(D), ids.to(D), question_ids=qids.to(D), max_new_tokens=4, do_sample=False).cpu()
    assert emb.dtype == torch.float16 and out.logits.dtype in (torch.float16, torch.float32)
    rep = {"inputs_embeds": three_way(emb, e32, eh), "bf16_oracle_vs_o32": err_stats(e16.float(), e32),
           "fp16_oracle_vs_o32": err_stats(eh.float(), e32),
           "logits_last_hip_vs_o32": err_stats(out.logits[:, -1].float().cpu(), logits32),
           "greedy_ids_hip": gen.tolist(), "greedy_ids_fp32": gen32.sequences.tolist()}
    margins = fp32_top2_margins(gen32.scores)
    rep["fp32_top2_margins"], rep["flip_threshold"] = margins, thr
    assert sum(mg_ > thr for mg_ in margins) >= 3, (margins, thr)
    n = compare_greedy_ids(gen[0].tolist(), gen32.sequences[0].tolist(), margins, thr)
    rep["steps_compared"] = n
    record("float16_model_config1", rep)
    # the yardstick is the reference's own fp16 run (three_way's "o16" slot holds it here): 1.2 x + a quarter of the bf16 eps
    e_hip, e_orc = rep["inputs_embeds"]["hip_vs_o32"], rep["inputs_embeds"]["o16_vs_o32"]
    assert e_hip["rel_rms"] <= 1.2 * e_orc["rel_rms"] + 5e-5, (e_hip, e_orc)
    assert e_hip["max_abs"] <= 1.6 * e_orc["max_abs"] + 2.0 ** -11 * e_hip["ref_rms"] * 4, (e_hip, e_orc)
    assert n >= 3 and gen.shape == gen32.sequences.shape, (gen, gen32.sequences, margins, thr)


@pytest.mark.host_heavy(28)   # nominal seconds, mostly host (tests/suite_budget.py)
def test_config2_full_path_vs_oracle():
    """BASELINE configs[1]: E = 2048, 128^3 volumes = 4 chunks of (32,128,128), batch 4, full tokenizer."""
    c = mm_config(2048, [32, 128, 128])
    run_full_config("config2_E2048_128cube_b4", c, B=4, C=4, S=1024, Lt=1024, seed=72)


def test_config1_survey_size_through_qwen3():
    """BASELINE configs[0] at the size SURVEY.md section 8(d) gives it: one 64^3 volume = 2 chunks of (32,64,64), E = 2048,
    1 tokenizer layer, hard top-k 16, no multi-scale, 256 queries, text 1024 -- through u2Qwen3ForCausalLM (2-layer
    Qwen3 decoder of Qwen3-1.7B width, helpers.decisive_decoder_) on the GPU vs oracle + the same HF decoder on the host: spliced
    embeddings, first-step logits, greedy ids for two volumes (see _id_gate)."""
    from u2tokenizer_amd.language_model import u2Qwen3Config, u2Qwen3ForCausalLM
    E, vocab, S, Lt, seed, dseed = 2048, 4096, 320, 1024, 73, 0
    c = mm_config(E, [32, 64, 64], u2t_num_layers=1, u2t_top_k=16, use_multi_scale=False, enable_diffts=False,
                  enable_dmtp=False)
    cfg = u2Qwen3Config(vocab_size=vocab, hidden_size=E, intermediate_size=6144, num_hidden_layers=2,
                        num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048,
                        tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    for k, v in c.items():
        if k != "hidden_size":
            setattr(cfg, k, v)
    m = u2Qwen3ForCausalLM(cfg).eval()
    synth.fill_module_(m, seed=seed, lively=True)
    decisive_decoder_(m, dseed)
    sd32 = {k: v.clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    sd16 = {k: v.to(bf) for k, v in sd32.items()}
    vols = {"noise": synth.synth_volume(1, 2, c["image_size"], seed=seed, dtype=torch.float16),
            "smooth": smooth_volume(1, 2, c["image_size"])}
    ids = synth.synth_ids(1, S, S - 8, vocab, seed=seed, name="input_ids")
    qids = synth.synth_ids(1, Lt, 40, vocab, seed=seed, name="question_ids")
    oc = oracle_cfg(c)
    new = 4
    e32, refs, idx32, idx16 = {}, {}, None, None
    for v, vol in vols.items():
        e32[v], idx = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], ids, vol.float(), qids, oc)
        idx32 = idx if v == "noise" else idx32
        refs[v] = _greedy_reference(m, e32[v], new)
    logits32 = m(inputs_embeds=e32["noise"]).logits[:, -1]
    m16 = m.to(bf)
    e16, thrs = {}, {}
    for v, vol in vols.items():
        e16[v], idx = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], ids, vol.to(bf), qids, oc)
        idx16 = idx if v == "noise" else idx16
        l16 = m16(inputs_embeds=e16[v]).logits[0, -1].float()
        thrs[v] = 4 * float((l16 - refs[v][2]).abs().max())
        if v == "noise":
            logits16 = l16[None]
    mg = m16.to(D)
    vol = vols["noise"]
    tok = mg.get_u2tokenizer()
    tok.capture_svr_tokens = True
    r = mg.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))
    assert r[0] is None and r[4].shape == (1, S, E)
    # north_star's "bit-exact token indices" AT THE PATH: the indices the HIP path selected inside this whole
    # forward are, index for index and in order, what the oracle's TokenSelection (svr.py:75-91) selects from the refined tokens the
    # HIP selection stage itself saw.  Against the fp32 run end to end the measure can only be SET agreement -- and the yardstick for
    # that is the reference's own bf16 run, recorded beside it (is ITS index list equal to the fp32 one on these inputs?).
    ih = tok.last_topk_indices.cpu()
    svr = tok.last_svr_tokens.cpu()
    T_ = vol.shape[1]
    _, oidx = O.token_selection(sd16, "model.u2tokenizer.svt_module.token_selection",
                                svr.view(1, T_, svr.shape[1] // T_, E), c["u2t_top_k"])
    assert torch.equal(ih, oidx), "hard top-k indices differ from the oracle's selection on the HIP path's own SVR output"
    tok.capture_svr_tokens = False
    out = mg(images=vol.to(D), input_ids=ids.to(D), question_ids=qids.to(D))
    topk_equal = bool(torch.equal(tok.last_topk_indices.cpu(), idx32))
    k_ = c["u2t_top_k"]
    ov = lambda a, b: len(set(a[0].tolist()) & set(b[0].tolist())) / k_  # noqa: E731
    gen = {v: mg.generate(x.to(D), ids.to(D), question_ids=qids.to(D), max_new_tokens=new, do_sample=False).cpu()[0].tolist()
           for v, x in vols.items()}
    rep = {"inputs_embeds": three_way(r[4], e32["noise"], e16["noise"]), "logits_last": three_way(out.logits[:, -1], logits32, logits16),
           "topk_idx_equal_fp32": topk_equal, "topk_idx_equal_fp32_of_the_bf16_oracle": bool(torch.equal(idx16, idx32)),
           "topk_set_overlap_fp32": ov(ih, idx32), "topk_set_overlap_fp32_of_the_bf16_oracle": ov(idx16, idx32),
           "topk_idx_equal_oracle_on_own_svr_output": True}
    try:
        # end to end the HIP selection is as close to the fp32 one as the reference's own bf16 run is (one index of slack)
        assert rep["topk_set_overlap_fp32"] >= rep["topk_set_overlap_fp32_of_the_bf16_oracle"] - 1.0 / k_ - 1e-9, rep
        _id_gate(rep, gen, refs, thrs)
    finally:
        record("config1_E2048_64cube_qwen3", rep)
    gate(rep["inputs_embeds"], "config1 inputs_embeds")
    gate(rep["logits_last"], "config1 logits")


@pytest.mark.parametrize("E", [2048, 4096])
def test_tokenizer_one_layer_full_width(E):
    """One selective SVR layer + DiffTS(1024) + DMTP pooling + one TTA layer + aggregation at the width, token counts and
    text length of BASELINE configs 2 / 3, on synthetic visual tokens without a common mode (query / key gain 3: the
    attention picks specific keys, yet one layer keeps the reference's bf16 run within tens of percent of its fp32 run).
    Besides the usual bar, the HIP output must be far closer to the fp32 reference than the same reference with its
    query rows permuted is -- i.e. the comparison is not vacuous."""
    from helpers import module_sd
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    seed, args = 77, (E, 8, 1, 1024, True, 256, E, "rma", True, True)
    sd32 = module_sd(u2Tokenizer(*args), "u2tokenizer.", seed)
    for name, t in sd32.items():
        synth.lively_(name, t, qk_gain=3.0)
    sd16 = {k: v.to(bf) for k, v in sd32.items()}
    with torch.device("meta"):
        tok = u2Tokenizer(*args)
    tok = tok.to(bf).to_empty(device=D)
    tok.load_state_dict({k[len("u2tokenizer."):]: v for k, v in sd16.items()})
    B = 2
    v = synth.synth_tensor("v_token", (B, 8, 256, E), seed)
    t = 0.25 * synth.synth_tensor("t_token", (B, 1024, E), seed)
    oc = O.PathConfig(hidden_size=E, u2t_num_layers=1)
    o32, _ = O.tokenizer_forward(sd32, "u2tokenizer", v, t, oc)
    o16, _ = O.tokenizer_forward(sd16, "u2tokenizer", v.to(bf), t.to(bf), oc)
    got = tok(v_token=v.to(bf).to(D), t_token=t.to(bf).to(D))
    d = three_way(got, o32, o16)
    d["permuted_rows_vs_o32"] = err_stats(o32.roll(1, dims=1), o32)
    record(f"tokenizer_one_layer_E{E}", d)
    gate(d, f"one layer E={E}")
    assert d["diversity_o32"] > 0.3, d["diversity_o32"]
    assert d["o16_vs_o32"]["rel_rms"] < 0.6, "chaotic regime: the comparison would be vacuous"
    assert d["hip_vs_o32"]["rel_rms"] < 0.5 * d["permuted_rows_vs_o32"]["rel_rms"], d


@pytest.mark.host_heavy(39)
@pytest.mark.parametrize("E", [4096])
def test_tokenizer_layers_teacher_forced_full_width(E):
    """Every layer of the 4-layer tokenizer at full width on NON-collapsed inputs, in ONE forward of the product path.

    The chained configuration tests cannot see layers 1-3: the SVR stack has no residuals or norms (svr.py:29,35), so its
    fp32 chain is either contracted onto the token mean by the second layer (token diversity 7e-3 / 2e-4 / 1e-5 after
    layers 1 / 2 / 3 at E = 4096, query / key gain 1) or chaotic (gain 3: the reference's own bf16 run is 40 % off after two
    layers).  Here each SVR layer, the visual tokens of the aggregation stage and each TTA layer get an independent lively
    input through the parity taps of u2tok_tokenizer_forward_taps (teacher forcing), the same tensors the oracle layer is
    run on in fp32 and in bf16 -- so the weight-table offsets, scratch reuse and side-stream ordering of layers 1-3 are
    exercised with token-dependent data.  Per layer: the usual three-distance gate AND the HIP output must be at least
    twice as close to the fp32 reference as that reference with its rows rotated by one is."""
    from helpers import module_sd
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    seed, L, B, T, N, Q, Lt, k = 78, 4, 1, 8, 256, 256, 1024, 1024
    args = (E, 8, L, k, True, Q, E, "rma", True, True)
    sd32 = module_sd(u2Tokenizer(*args), "u2tokenizer.", seed)
    for name, t in sd32.items():
        synth.lively_(name, t, qk_gain=3.0)
    sd16 = {kk: v.to(bf) for kk, v in sd32.items()}
    with torch.device("meta"):
        tok = u2Tokenizer(*args)
    tok = tok.to(bf).to_empty(device=D)
    tok.load_state_dict({kk[len("u2tokenizer."):]: v for kk, v in sd16.items()})
    oc = O.PathConfig(hidden_size=E, u2t_num_layers=L)
    Lv = k + k // 2 + k // 4
    # teacher-forced inputs (bf16-representable, so fp32 oracle, bf16 oracle and the HIP path read identical values)
    v = synth.synth_tensor("v_token", (B, T, N, E), seed).to(bf)
    t = (0.25 * synth.synth_tensor("t_token", (B, Lt, E), seed)).to(bf)
    svr_in = [v] + [(0.5 * synth.synth_tensor(f"svr_in_{l}", (B, T, N, E), seed)).to(bf) for l in range(1, L)]
    visual_in = (0.3 * synth.synth_tensor("visual_in", (B, Lv, E), seed)).to(bf)
    tta_in = [None] + [synth.synth_tensor(f"tta_in_{l}", (B, Q, E), seed).to(bf) for l in range(1, L)]
    out, taps = tok.forward_with_taps(v.to(D), t.to(D), svr_in=[None] + [x.to(D) for x in svr_in[1:]],
                                      visual_in=visual_in.to(D), tta_in=[None] + [x.to(D) for x in tta_in[1:]])
    rep, p = {}, "u2tokenizer"

    def check(name, hip, o32, o16, discriminate=True):
        d = three_way(hip, o32, o16)
        d["permuted_rows_vs_o32"] = err_stats(o32.roll(1, dims=-2), o32)
        rep[name] = d
        assert torch.isfinite(hip.float()).all(), name
        gate(d, name)
        if discriminate:  # (the attention layers; selection / aggregation outputs discriminate when they are diverse)
            assert d["diversity_o32"] > 0.2, (name, d["diversity_o32"])
        if d["diversity_o32"] > 0.2:
            assert d["hip_vs_o32"]["rel_rms"] < 0.5 * d["permuted_rows_vs_o32"]["rel_rms"], (name, d)

    last = {}
    for l in range(L):
        lp = f"{p}.svt_module.attention_network.layers.{l}"
        o32 = O.st_attention_layer(sd32, lp, svr_in[l].float(), oc)
        o16 = O.st_attention_layer(sd16, lp, svr_in[l], oc)
        check(f"svr_layer_{l}", taps["svr_out"][l], o32, o16)
        last = {"o32": o32, "o16": o16}
    # selection + pooling on the last SVR layer's (teacher-forced) output
    vis = {}
    for key, sd_, dt in (("o32", sd32, torch.float32), ("o16", sd16, bf)):
        x = O.diff_token_selection(sd_, f"{p}.svt_module.token_selection", last[key])
        vis[key] = O.multi_scale_pool(sd_, f"{p}.svt_module.dynamic_pool", x)
    check("selection_pooling", taps["visual_out"], vis["o32"], vis["o16"], discriminate=False)
    q32, q16 = sd32[f"{p}.query_tokens"].expand(B, -1, -1), sd16[f"{p}.query_tokens"].expand(B, -1, -1)
    for l in range(L):
        lp = f"{p}.tta_module.layers_vt.{l}"
        i32 = q32 if l == 0 else tta_in[l].float()
        i16 = q16 if l == 0 else tta_in[l]
        o32 = O.tta_layer(sd32, lp, i32, visual_in.float(), t.float(), oc)
        o16 = O.tta_layer(sd16, lp, i16, visual_in, t, oc)
        check(f"tta_layer_{l}", taps["tta_out"][l], o32, o16)
        last = {"o32": o32, "o16": o16}
    lin = f"{p}.tta_module.layer_linagg.linear_aggregator"
    check("linear_aggregation", out, O.cross_attention(sd32, lin, last["o32"], visual_in.float(), 8, is_compress=True),
          O.cross_attention(sd16, lin, last["o16"], visual_in, 8, is_compress=True), discriminate=False)
    record(f"tokenizer_layers_teacher_forced_E{E}", rep)


def test_cls_patch_feature_selection():
    """select_feature = "cls_patch" (vit.py:159-160): the kernel keeps the cls rows after all patch rows internally and
    restores the reference's [cls | patches] order in its final LayerNorm."""
    from helpers import module_sd
    from u2tokenizer_amd.vit import ViT3DTower
    img = [32, 64, 64]
    m = ViT3DTower(NS(vision_select_layer=-1, vision_select_feature="cls_patch", image_channel=1, image_size=img,
                      patch_size=[4, 16, 16]))
    sd32 = module_sd(m, "vision_tower.", 74)
    synth.fill_module_(m, seed=74, prefix="vision_tower.")
    vol = synth.synth_volume(1, 3, img, seed=74, dtype=torch.float16).view(3, 1, *img)
    oc = O.PathConfig(image_size=img, vision_select_feature="cls_patch")
    o32 = O.vit_tower_forward(sd32, "vision_tower.vision_tower", vol.float(), oc)
    o16 = O.vit_tower_forward({k: v.to(bf) for k, v in sd32.items()}, "vision_tower.vision_tower", vol.to(bf), oc)
    got = m.to(bf).to(D)(vol.to(D))
    assert got.shape == (3, 129, 768)
    d = three_way(got, o32, o16)
    record("vit_cls_patch", d)
    gate(d, "cls_patch")
    # the cls row really is row 0: compare it on its own
    gate(three_way(got[:, :1], o32[:, :1], o16[:, :1]), "cls row")


def test_path_without_u2tokenizer():
    """enable_u2tokenizer = False (u2_arch.py:111-112): one resized (B,1,D,H,W) volume through ViT + SPP, its
    proj_out_num tokens spliced directly (the M3D-style baseline)."""
    E, vocab, S, seed = 512, 512, 40, 75
    c = mm_config(E, [32, 64, 64], enable_u2tokenizer=False)
    path, sd32, sd16 = build_path(c, vocab, seed)
    vol = synth.synth_volume(2, 1, c["image_size"], seed=seed, dtype=torch.float16)  # (B, 1, D, H, W)
    ids = synth.synth_ids(2, S, S - 4, vocab, seed=seed, name="input_ids")
    oc = oracle_cfg(c)
    e32, _ = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], ids, vol.float(), None, oc)
    e16, _ = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], ids, vol.to(bf), None, oc)
    r = path.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), None)
    assert r[4].shape == (2, S, E)
    d = three_way(r[4], e32, e16)
    record("path_without_u2tokenizer", d)
    gate(d, "no tokenizer")
    # rows outside the 16 spliced positions are plain embedding rows: bit-exact
    emb = sd16["model.embed_tokens.weight"][ids]
    assert torch.equal(r[4][:, 17:].cpu(), emb[:, 17:]) and torch.equal(r[4][:, :1].cpu(), emb[:, :1])


@pytest.mark.parametrize("layers,qk_gain", [(1, 2.0), (2, 2.0), (4, 4.0)])
def test_hard_topk_end_to_end_agreement(layers, qk_gain):
    """The path's integer output at BASELINE size (8 x 256 tokens -> top 1024, E = 2048): how well do the indices of the
    bf16 HIP pipeline agree with the fp32 reference run END TO END?  Any bf16 pipeline perturbs the scores by its
    rounding error and the residual-free SVR stack amplifies that (measured here: 1 layer keeps most of the set, 4
    selective layers scramble it to chance level for the reference's own bf16 run as well), so the yardstick is the
    reference's bf16 run: the HIP indices must agree with the fp32 ones at least as well as the bf16 oracle's do
    (set overlap, minus 2 % slack; at chance level: inside the 4-sigma band around k/n).  Bit-exactness given identical inputs is pinned by test_hard_topk_full_size_replay
    and test_tokenizer_vs_reference."""
    from helpers import module_sd
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    E, k, seed = 2048, 1024, 76
    args = (E, 8, layers, k, True, 256, E, "rma", False, True)
    sd32 = module_sd(u2Tokenizer(*args), "u2tokenizer.", seed)
    for name, t in sd32.items():
        if name.endswith(".wq.weight") or name.endswith(".wk.weight"):
            t.mul_(qk_gain)
    sd16 = {kk: v.to(bf) for kk, v in sd32.items()}
    with torch.device("meta"):
        tok = u2Tokenizer(*args)
    tok = tok.to(bf).to_empty(device=D)
    tok.load_state_dict({kk[len("u2tokenizer."):]: v for kk, v in sd16.items()})
    v = synth.synth_tensor("v_token", (2, 8, 256, E), seed)
    t = 0.25 * synth.synth_tensor("t_token", (2, 64, E), seed)
    oc = O.PathConfig(hidden_size=E, enable_diffts=False, u2t_num_layers=layers)
    _, i32 = O.tokenizer_forward(sd32, "u2tokenizer", v, t, oc)
    _, i16 = O.tokenizer_forward(sd16, "u2tokenizer", v.to(bf), t.to(bf), oc)
    tok(v_token=v.to(bf).to(D), t_token=t.to(bf).to(D))
    ih = tok.last_topk_indices.cpu()

    def agree(a, b):
        sets = [len(set(a[r].tolist()) & set(b[r].tolist())) / k for r in range(a.shape[0])]
        return min(sets), (a == b).float().mean().item()

    hs, ho = agree(ih, i32)
    os_, oo = agree(i16, i32)
    record(f"hard_topk_end_to_end_L{layers}", {"hip_vs_fp32": {"set_overlap": hs, "same_position": ho},
                                               "oracle_bf16_vs_fp32": {"set_overlap": os_, "same_position": oo},
                                               "k": k, "n": 2048, "chance_overlap": k / 2048, "qk_gain": qk_gain})
    # Two k-subsets of n drawn independently overlap in k/n of their elements, +- sigma (hypergeometric).  Once the reference's
    # own bf16 run has fallen to that level (four selective layers), it and the HIP run are two independent draws around it: the
    # minimum over the batch rows then only has to stay inside the band (4 sigma), it cannot be asked to beat another draw by 2 %
    # (round 5: the 8-wave attention kernel sums in another order and drew 0.469 against 0.497).
    n, chance = 2048, k / 2048
    sigma = (k * chance * (1 - chance) * (n - k) / (n - 1)) ** 0.5 / k
    floor = os_ - 0.02 if os_ > chance + 4 * sigma else min(os_ - 0.02, chance - 4 * sigma)
    assert hs >= floor, (hs, os_, chance, sigma)


def test_frozen_vision_tower_is_shared_between_policy_and_reference():
    """DPO with a frozen vision tower runs the ViT on the same images for the policy and for the reference model
    (dpo_u2trainer.py builds the image batch anew for each): after `share_frozen_vision_tower` the second pass returns the
    first pass's features (byte-identical input, unchanged weights) -- same embeddings as an unshared run, the ViT launched
    once; another image, or an updated tower, computes again."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    from u2tokenizer_amd import share_frozen_vision_tower
    E, vocab = 2048, 4096
    pol, _ = bench.build_path(E, vocab, D)
    ref, _ = bench.build_path(E, vocab, D)         # (same seed: equal weights, separate modules)
    g = torch.Generator(device=D).manual_seed(3)
    vol = torch.rand((1, 8, 32, 256, 256), device=D, generator=g).half()
    ids = torch.randint(1, vocab, (1, 1024), device=D, generator=g)
    qids = torch.zeros((1, 1024), dtype=torch.int64, device=D)
    qids[:, :30] = torch.randint(1, vocab, (1, 30), device=D, generator=g)
    with torch.no_grad():
        want = ref.prepare_inputs_for_multimodal(ids, None, None, None, None, vol, qids)[4].clone()
        pol.holder.vision_tower.requires_grad_(True)
        with pytest.raises(RuntimeError):
            share_frozen_vision_tower(pol, ref)        # a trainable tower cannot be shared
        pol.holder.vision_tower.requires_grad_(False)
        ref.holder.vision_tower.requires_grad_(False)
        share_frozen_vision_tower(pol, ref)
        assert ref.holder.vision_tower is pol.holder.vision_tower
        tower = pol.holder.vision_tower
        calls = []
        inner = tower.vision_tower.forward_features
        tower.vision_tower.forward_features = lambda x, keep_cls: (calls.append(1), inner(x, keep_cls))[1]
        a = pol.prepare_inputs_for_multimodal(ids, None, None, None, None, vol.clone(), qids)[4]
        b = ref.prepare_inputs_for_multimodal(ids, None, None, None, None, vol.clone(), qids)[4]   # another tensor, same bytes
        assert len(calls) == 1
        assert torch.equal(a, want) and torch.equal(b, want)
        vol2 = vol.clone()
        vol2[0, 3, 5, 7, 9] += 0.25
        c = ref.prepare_inputs_for_multimodal(ids, None, None, None, None, vol2, qids)[4]
        assert len(calls) == 2 and not torch.equal(c, want)
        tower.vision_tower.norm.weight.mul_(1.5)                                             # the tower changed: no stale features
        d = pol.prepare_inputs_for_multimodal(ids, None, None, None, None, vol2.clone(), qids)[4]
        assert len(calls) == 3 and not torch.equal(d, c)

