"""Shared test-case table: used by tests/golden/make_golden.py (reference side) and the parity tests."""
import torch

from u2tokenizer_amd import synth

_B = dict(heads=8, max_seq_len=512)

TOKENIZER_CASES = {
    # BASELINE.json config 1 flavour: 1 layer, hard top-k, no multi-scale
    "hard_1l": dict(_B, E=512, layers=1, B=1, T=2, N=16, Lt=24, Q=16, top_k=16, use_multi_scale=False,
                    attn_type="rma", enable_diffts=False, enable_dmtp=False, seed=11),
    # shipped flavour (rma + diffts + dmtp + multi-scale), scaled down; B=2 exercises the batch strides
    "mu2_2l": dict(_B, E=512, layers=2, B=2, T=4, N=64, Lt=40, Q=32, top_k=64, use_multi_scale=True,
                   attn_type="rma", enable_diffts=True, enable_dmtp=True, seed=12),
    # rope attention, hard top-k feeding fixed multi-scale pooling; odd sizes exercise tails (N, Lt, top_k % 8 != 0)
    "rope_fix": dict(_B, E=512, layers=1, B=1, T=3, N=20, Lt=13, Q=10, top_k=30, use_multi_scale=True,
                     attn_type="rope", enable_diffts=False, enable_dmtp=False, seed=13),
    # diffts with fixed pooling (the "diffts" ablation)
    "diffts_fix": dict(_B, E=512, layers=1, B=1, T=2, N=32, Lt=16, Q=16, top_k=24, use_multi_scale=True,
                       attn_type="rma", enable_diffts=True, enable_dmtp=False, seed=14),
    # attn_type outside {rma, rope} -> stock nn.MultiheadAttention read sequence-first (the "linvt" ablation)
    "linvt_2l": dict(_B, E=512, layers=2, B=1, T=4, N=24, Lt=20, Q=16, top_k=32, use_multi_scale=True,
                     attn_type="linvt", enable_diffts=True, enable_dmtp=True, seed=15),
    # ---- "lively" parameter sets (synth.lively_scale): attention stays selective through the residual-free SVR stack,
    # so token-dependent data reaches the selection, pooling and aggregation stages (the plain sets above collapse to
    # the token mean after two SVR layers -- they pin the bias / table / layout paths, these pin the data path).
    # shipped flavour, full depth (4 layers), T = 8 chunks like a 256^3 volume
    "mu2_4l_live": dict(_B, E=512, layers=4, B=2, T=8, N=32, Lt=40, Q=32, top_k=96, use_multi_scale=True,
                        attn_type="rma", enable_diffts=True, enable_dmtp=True, seed=16, lively=True),
    # hard top-k on selective attention output: the reference's torch.topk order must equal the canonical order
    "hard_2l_live": dict(_B, E=512, layers=2, B=2, T=4, N=32, Lt=24, Q=16, top_k=48, use_multi_scale=True,
                         attn_type="rma", enable_diffts=False, enable_dmtp=True, seed=17, lively=True),
    "rope_2l_live": dict(_B, E=512, layers=2, B=1, T=5, N=24, Lt=18, Q=12, top_k=40, use_multi_scale=True,
                         attn_type="rope", enable_diffts=True, enable_dmtp=False, seed=18, lively=True),
    # T > 16 chunks (the reference's own smoke uses 64 frames, svr.py:190-205)
    "mu2_t24_live": dict(_B, E=512, layers=1, B=1, T=24, N=16, Lt=16, Q=16, top_k=64, use_multi_scale=True,
                         attn_type="rma", enable_diffts=True, enable_dmtp=True, seed=19, lively=True),
    # nn.MultiheadAttention read sequence-first with B = 2: attention runs ACROSS the batch entries (svr.py:28-35)
    "linvt_b2_live": dict(_B, E=512, layers=2, B=2, T=4, N=24, Lt=20, Q=16, top_k=32, use_multi_scale=True,
                          attn_type="linvt", enable_diffts=True, enable_dmtp=True, seed=20, lively=True),
}

SPP_CASES = {
    "mlp2": dict(image_size=[32, 64, 64], patch_size=[4, 16, 16], in_dim=768, E=512, layer_type="mlp", layer_num=2,
                 pooling_type="spatial", pooling_size=2, nchunk=3, seed=21),
    "seq_lin": dict(image_size=[32, 64, 64], patch_size=[4, 16, 16], in_dim=768, E=256, layer_type="linear",
                    layer_num=2, pooling_type="sequence", pooling_size=2, nchunk=2, seed=22),
}

VIT_CASES = {
    # BASELINE.json config 1 geometry: 64^3 volume = 2 chunks of (32,64,64) -> 128 patches (+cls) per chunk
    "c64": dict(image_size=[32, 64, 64], patch_size=[4, 16, 16], nchunk=2, select_feature="patch", seed=31),
}

_LLAMA_TINY = dict(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                   num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=256,
                   tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
FULL_CASES = {
    # config 1: single 64^3 volume, 1-scale, 1 attention block, hard top-k 16, tiny decoder standing in for the LLM
    "cfg1": dict(llama=_LLAMA_TINY, B=1, C=2, S=48, n_real=40, Lt=32, n_q=12, new_tokens=4, seed=41,
                 mm=dict(vision_tower="vit3d", image_channel=1, image_size=[32, 64, 64], patch_size=[4, 16, 16],
                         vision_select_layer=-1, vision_select_feature="patch", mm_projector_type="spp",
                         proj_layer_type="mlp", proj_layer_num=2, proj_pooling_type="spatial", proj_pooling_size=2,
                         mm_hidden_size=768, enable_u2tokenizer=True, u2t_num_heads=8, u2t_num_layers=1,
                         u2t_top_k=16, use_multi_scale=False, num_3d_query_token=16, attn_type="rma",
                         enable_diffts=False, enable_dmtp=False)),
}


def tokenizer_inputs(c):
    v = synth.synth_tensor("v_token", (c["B"], c["T"], c["N"], c["E"]), c["seed"])
    t = 0.25 * synth.synth_tensor("t_token", (c["B"], c["Lt"], c["E"]), c["seed"])
    return v, t


def spp_inputs(c):
    g = [i // p for i, p in zip(c["image_size"], c["patch_size"])]
    return synth.synth_tensor("spp_in", (c["nchunk"], g[0] * g[1] * g[2], c["in_dim"]), c["seed"])

# BASELINE configs[3] (stage-1 training step of the path at its real size): tests/golden/make_config4_grads.py differentiates
# the oracle here, tests/test_gpu_backward.py::test_config4_full_depth_full_width_gradients the HIP path on the GPU
CONFIG4_CASE = dict(E=4096, image_size=[32, 256, 256], vocab=4096, S=1024, Lt=1024, Q=256, seed=95)
