"""CPU: the part of the oracle the reference cannot pin.  vit.py:19-20 imports two MONAI 1.3.0 blocks that are absent
offline, so oracle/u2_oracle.py restates them ("parity unpinned").  These tests narrow that surface by checking the
restatement against INDEPENDENT implementations of the same published semantics that ship with torch itself (and, for the
block stack, with transformers):

  * SABlock (qkv Linear without bias, "b h (qkv l d) -> qkv b l h d", softmax(q k^T d^-0.5) v, out_proj)
        == torch's own multi-head attention kernel with in_proj_weight = qkv.weight (same [q | k | v] x [head] x [d]
           feature order), no in-projection bias;
  * TransformerBlock (x + attn(norm1 x); x + mlp(norm2 x), exact-erf GELU)
        == nn.TransformerEncoderLayer(norm_first=True, activation=gelu) carrying the same parameters;
  * PatchEmbeddingBlock("perceptron": Rearrange "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)" + Linear + pos)
        == a strided Conv3d(kernel = stride = patch) whose filters are the Linear rows reshaped to (p1, p2, p3, c).

  * a stack of TransformerBlocks under the tower's final LayerNorm
        == transformers' own ViT layers (separate q / k / v Linears split from the fused qkv rows) carrying the same parameters.

They do not replace running MONAI (the header of the oracle keeps saying so); they rule out a private mistake in the
restatement's index gymnastics, which is where a restatement goes wrong.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import u2_oracle as O
from u2tokenizer_amd import synth

torch.set_grad_enabled(False)


def _sd(shapes, seed=5):
    return {k: synth.synth_tensor(k, s, seed) for k, s in shapes.items()}


def test_sablock_equals_torch_multihead_attention():
    hid, heads, B, S = 96, 4, 3, 17
    sd = _sd({"a.qkv.weight": (3 * hid, hid), "a.out_proj.weight": (hid, hid), "a.out_proj.bias": (hid,)})
    x = synth.synth_tensor("x", (B, S, hid), 5)
    got = O.sa_block(sd, "a", x, heads)
    want, _ = F.multi_head_attention_forward(
        x.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1), hid, heads, sd["a.qkv.weight"], None, None, None, False,
        0.0, sd["a.out_proj.weight"], sd["a.out_proj.bias"], training=False, need_weights=False)
    assert torch.allclose(got, want.transpose(0, 1), rtol=1e-5, atol=1e-6)
    # and F.scaled_dot_product_attention on the explicitly split heads
    q, k, v = [t.view(B, S, heads, hid // heads).transpose(1, 2) for t in F.linear(x, sd["a.qkv.weight"]).chunk(3, -1)]
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, hid)
    assert torch.allclose(got, F.linear(o, sd["a.out_proj.weight"], sd["a.out_proj.bias"]), rtol=1e-5, atol=1e-6)


def test_transformer_block_equals_torch_encoder_layer():
    hid, mlp, heads, B, S = 64, 160, 4, 2, 11
    shapes = {"b.norm1.weight": (hid,), "b.norm1.bias": (hid,), "b.norm2.weight": (hid,), "b.norm2.bias": (hid,),
              "b.attn.qkv.weight": (3 * hid, hid), "b.attn.out_proj.weight": (hid, hid), "b.attn.out_proj.bias": (hid,),
              "b.mlp.linear1.weight": (mlp, hid), "b.mlp.linear1.bias": (mlp,), "b.mlp.linear2.weight": (hid, mlp),
              "b.mlp.linear2.bias": (hid,)}
    sd = _sd(shapes)
    x = synth.synth_tensor("x", (B, S, hid), 6)
    layer = nn.TransformerEncoderLayer(hid, heads, mlp, dropout=0.0, activation=F.gelu, batch_first=True, norm_first=True).eval()
    layer.self_attn.in_proj_weight.copy_(sd["b.attn.qkv.weight"])
    layer.self_attn.in_proj_bias.zero_()                       # qkv_bias=False in the reference tower (vit.py:47,101)
    layer.self_attn.out_proj.weight.copy_(sd["b.attn.out_proj.weight"])
    layer.self_attn.out_proj.bias.copy_(sd["b.attn.out_proj.bias"])
    for a, b in (("norm1", "norm1"), ("norm2", "norm2"), ("linear1", "mlp.linear1"), ("linear2", "mlp.linear2")):
        getattr(layer, a).weight.copy_(sd[f"b.{b}.weight"])
        getattr(layer, a).bias.copy_(sd[f"b.{b}.bias"])
    assert torch.allclose(O.transformer_block(sd, "b", x, heads), layer(x), rtol=2e-5, atol=2e-6)


def test_perceptron_patch_embedding_equals_strided_conv3d():
    c, hid, patch, img = 1, 48, (4, 16, 16), (8, 32, 48)
    n = (img[0] // patch[0]) * (img[1] // patch[1]) * (img[2] // patch[2])
    kp = patch[0] * patch[1] * patch[2] * c
    sd = _sd({"p.patch_embeddings.1.weight": (hid, kp), "p.patch_embeddings.1.bias": (hid,),
              "p.position_embeddings": (1, n, hid)})
    x = synth.synth_tensor("x", (2, c, *img), 7)
    got = O.patch_embedding_block(sd, "p", x, patch)
    w = sd["p.patch_embeddings.1.weight"].view(hid, *patch, c).permute(0, 4, 1, 2, 3)     # features are (p1 p2 p3 c)
    conv = F.conv3d(x, w, sd["p.patch_embeddings.1.bias"], stride=patch)                  # (b, hid, h, w, d)
    want = conv.flatten(2).transpose(1, 2) + sd["p.position_embeddings"]                  # tokens in (h w d) order
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_antialias_gaussian_taps_are_monai_unnormalised():
    """GaussianFilter.forward calls gaussian_1d(..., normalize=False): the taps keep the truncated tail out of their sum."""
    from oracle import u2_preprocess_oracle as P
    for sigma in (0.5, 1.0, 2.5):
        k = P.gaussian_1d(sigma)
        tail = (k.numel() - 1) // 2
        assert tail == int(max(sigma * 4.0, 0.5) + 0.5)
        x = torch.arange(-tail, tail + 1, dtype=torch.float64)
        cdf = lambda t: 0.5 * (1 + torch.erf(t / (sigma * 2 ** 0.5)))  # noqa: E731
        assert torch.allclose(k.double(), cdf(x + 0.5) - cdf(x - 0.5), atol=1e-6)        # bin integrals of N(0, sigma)
        assert 0.999 < float(k.sum()) < 1.0                                               # NOT renormalised
    assert P.gaussian_1d(0.0).tolist() == [0.0, 1.0, 0.0]


def test_block_stack_equals_huggingface_vit_layers():
    """A third implementation of the same published block: transformers' ViT layers (pre-LN blocks, separate q / k / v Linears
    without bias, erf-GELU MLP) stacked under a final LayerNorm, carrying the restatement's parameters -- the [q | k | v] x
    [head] x [d] row order of MONAI's fused qkv.weight is what splits into the three Linears."""
    import pytest
    from transformers import ViTConfig
    from transformers.models.vit.modeling_vit import ViTLayer
    hid, mlp, heads, depth, B, S = 64, 160, 4, 3, 2, 13
    cfg = ViTConfig(hidden_size=hid, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=mlp, hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5, qkv_bias=False)
    cfg._attn_implementation = "eager"
    layers = [ViTLayer(cfg).eval() for _ in range(depth)]
    if not all(hasattr(layers[0].attention, a) for a in ("q_proj", "k_proj", "v_proj", "o_proj")) or not hasattr(layers[0], "mlp"):
        pytest.skip("this transformers release lays its ViT layer out differently")
    shapes = {"norm.weight": (hid,), "norm.bias": (hid,)}
    for i in range(depth):
        shapes.update({f"blocks.{i}.norm1.weight": (hid,), f"blocks.{i}.norm1.bias": (hid,), f"blocks.{i}.norm2.weight": (hid,),
                       f"blocks.{i}.norm2.bias": (hid,), f"blocks.{i}.attn.qkv.weight": (3 * hid, hid),
                       f"blocks.{i}.attn.out_proj.weight": (hid, hid), f"blocks.{i}.attn.out_proj.bias": (hid,),
                       f"blocks.{i}.mlp.linear1.weight": (mlp, hid), f"blocks.{i}.mlp.linear1.bias": (mlp,),
                       f"blocks.{i}.mlp.linear2.weight": (hid, mlp), f"blocks.{i}.mlp.linear2.bias": (hid,)})
    sd = _sd(shapes, seed=8)
    for k in sd:                                               # LayerNorm weights around 1, as in a trained tower
        if k.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            sd[k] = 1.0 + 0.1 * sd[k]
    for i, layer in enumerate(layers):
        p = f"blocks.{i}."
        q, k, v = sd[p + "attn.qkv.weight"].chunk(3, 0)
        att = layer.attention
        assert att.q_proj.bias is None
        for dst, src in ((att.q_proj.weight, q), (att.k_proj.weight, k), (att.v_proj.weight, v),
                         (att.o_proj.weight, sd[p + "attn.out_proj.weight"]), (att.o_proj.bias, sd[p + "attn.out_proj.bias"]),
                         (layer.layernorm_before.weight, sd[p + "norm1.weight"]), (layer.layernorm_before.bias, sd[p + "norm1.bias"]),
                         (layer.layernorm_after.weight, sd[p + "norm2.weight"]), (layer.layernorm_after.bias, sd[p + "norm2.bias"]),
                         (layer.mlp.fc1.weight, sd[p + "mlp.linear1.weight"]), (layer.mlp.fc1.bias, sd[p + "mlp.linear1.bias"]),
                         (layer.mlp.fc2.weight, sd[p + "mlp.linear2.weight"]), (layer.mlp.fc2.bias, sd[p + "mlp.linear2.bias"])):
            dst.copy_(src)
    x = synth.synth_tensor("x", (B, S, hid), 8)
    got, want = x, x
    for i, layer in enumerate(layers):
        got = O.transformer_block(sd, f"blocks.{i}", got, heads)
        want = layer(want)
        want = want[0] if isinstance(want, tuple) else want
    got = F.layer_norm(got, (hid,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    want = F.layer_norm(want, (hid,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6), float((got - want).abs().max())
