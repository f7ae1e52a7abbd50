"""GPU diagnostic: is the spliced-embedding result of the config-3 end-to-end case (tests/test_gpu_configs.py) a function
of its inputs only?  Runs prepare_inputs_for_multimodal under a list of ambient states (feature cache hit / miss, TTA
overlap, a registered split-K scratch, after a fused prefill, after an autograd step, other flash loops) and prints each
result's distance from the first and from the host oracles (fp32, bf16)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import err_stats  # noqa: E402
from oracle import u2_oracle as O  # noqa: E402
from u2tokenizer_amd import ops, synth  # noqa: E402
from u2tokenizer_amd.language_model import u2Qwen3Config, u2Qwen3ForCausalLM  # noqa: E402
from test_gpu_configs import mm_config, oracle_cfg  # noqa: E402

bf, D = torch.bfloat16, "cuda"
torch.set_grad_enabled(False)
with_oracle = "--oracle" in sys.argv
E, vocab, S, Lt, seed = 4096, 4096, 1024, 1024, 75
c = mm_config(E, [32, 256, 256])
cfg = u2Qwen3Config(vocab_size=vocab, hidden_size=E, intermediate_size=12288, num_hidden_layers=1,
                    num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048,
                    tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
for k, v in c.items():
    if k != "hidden_size":
        setattr(cfg, k, v)
m = u2Qwen3ForCausalLM(cfg).eval()
synth.fill_module_(m, seed=seed, lively=True)
vol = synth.synth_volume(1, 8, c["image_size"], seed=seed, dtype=torch.float16)
ids = synth.synth_ids(1, S, S - 24, vocab, seed=seed, name="input_ids")
qids = synth.synth_ids(1, Lt, 40, vocab, seed=seed, name="question_ids")
e32 = e16 = None
if with_oracle:
    t0 = time.time()
    sd32 = {k: v.clone() for k, v in m.state_dict().items() if v.is_floating_point()}
    sd16 = {k: v.to(bf) for k, v in sd32.items()}
    oc = oracle_cfg(c)
    e32, _ = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], ids, vol.float(), qids, oc)
    e16, _ = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], ids, vol.to(bf), qids, oc)
    print(f"oracles {time.time() - t0:.1f} s; o16_vs_o32 rel_rms {err_stats(e16.float(), e32)['rel_rms']:.6f}", flush=True)
mg = m.to(bf).to(D)
tower = mg.get_model().get_vision_tower()
volD, idsD, qD = vol.to(D), ids.to(D), qids.to(D)


def prep():
    out = mg.prepare_inputs_for_multimodal(idsD, None, None, None, None, volD, qD)[4]
    torch.cuda.synchronize()
    return out.float().cpu()


r0 = None


def show(name, r):
    global r0
    if r0 is None:
        r0 = r
    line = f"{name:44s} vs first: rel_rms {err_stats(r, r0)['rel_rms']:.6f} max {float((r - r0).abs().max()):.5f}"
    if e32 is not None:
        line += f" | vs o32 {err_stats(r, e32)['rel_rms']:.6f} vs o16 {err_stats(r, e16.float())['rel_rms']:.6f}"
    print(line, flush=True)


show("fresh", prep())
show("again (feature cache hit)", prep())
tower.invalidate_feature_cache()
show("cache invalidated", prep())
ops.set_option("tta_overlap", 0)
tower.invalidate_feature_cache()
show("tta_overlap 0", prep())
ops.set_option("tta_overlap", 1)
scratch = torch.empty(72 << 20, dtype=torch.uint8, device=D)
ops.set_gemm_scratch(scratch)
tower.invalidate_feature_cache()
show("ambient split-K scratch", prep())
ops.set_gemm_scratch(None)
out = mg(images=volD, input_ids=idsD, question_ids=qD)
torch.cuda.synchronize()
print("fused prefill on:", hasattr(mg.model.layers[0], "_u2_prefill"), flush=True)
show("after a full forward (prefill registered)", prep())
tower.invalidate_feature_cache()
show("  + cache invalidated", prep())
for name, val in (("flash_mode", 1), ("flash_q_prescaled", 1), ("vit_flash", 0)):
    ops.set_option(name, val)
    tower.invalidate_feature_cache()
    try:
        show(f"{name}={val}", prep())
    finally:
        ops.set_option(name, {"flash_mode": 0, "flash_q_prescaled": 0, "vit_flash": 1}[name])
# an autograd step on a small tokenizer (registers the training path's state on this context)
from u2tokenizer_amd.tokenizer import u2Tokenizer  # noqa: E402
with torch.enable_grad():
    tk = u2Tokenizer(embed_size=512, num_heads=8, num_layers=1, top_k=64, use_multi_scale=False, num_3d_query_token=64,
                     hidden_size=512, attn_type="rma", enable_diffts=True, enable_dmtp=False)
    synth.fill_module_(tk, seed=3)
    tk = tk.to(bf).to(D).train()
    v = torch.randn(1, 2, 64, 512, device=D, dtype=bf, requires_grad=True)
    t = torch.randn(1, 16, 512, device=D, dtype=bf)
    tk(v_token=v, t_token=t).float().square().mean().backward()
torch.cuda.synchronize()
tower.invalidate_feature_cache()
show("after an autograd step elsewhere", prep())
for i in range(3):
    tower.invalidate_feature_cache()
    show(f"repeat {i}", prep())
