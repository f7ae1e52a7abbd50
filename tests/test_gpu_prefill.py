"""GPU: the decoder prefill on the spliced embeddings (u2llama.py:76-87,123-126) through the HIP kernels
(u2tokenizer_amd/prefill.py) against the stock HuggingFace decoder -- building blocks against fp32 torch expressions, whole
models (Qwen3 with its per-head q / k norms and Llama, grouped-query heads) against the same model in fp32 on the host with the
stock bf16 GPU run as the yardstick, and `generate` through the patched layers (prefill fused, decode steps stock, one cache)."""
import math

import pytest
import torch
import torch.nn.functional as F

from u2tokenizer_amd import synth

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
D = "cuda"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from u2tokenizer_amd import ops as _ops
    _ops.device_check()
    torch.set_grad_enabled(False)
    return _ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed * 7919 + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(bf)


ULP, EPS = 2.0 ** -8, 1e-3   # of the element type under test (tests/test_gpu_f16.py re-runs this module with bf = float16: 2^-11, 1.5e-4)


def close_bf16(got, ref, rounds=2):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all()
    tol = rounds * ULP * ref.abs() + ULP * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{bad.sum().item()} elements off; worst {(got - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("rows,C", [(1024, 4096), (77, 2048), (5, 512), (300, 8192)])
def test_rmsnorm(ops, rows, C):
    x, w = rnd(rows, C, seed=1), (1 + 0.1 * rnd(C, seed=2).float()).to(bf)
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(bf).float() * w.float()   # HF's rounding points
    close_bf16(ops.rmsnorm(x.to(D), w.to(D), 1e-6), ref)


@pytest.mark.parametrize("rows,Hq,Hkv,d,norm,f32", [(1024, 32, 8, 128, True, False), (70, 8, 4, 64, False, True),
                                                     (33, 4, 4, 128, True, True), (9, 32, 8, 64, False, False)])
def test_qk_norm_rope(ops, rows, Hq, Hkv, d, norm, f32):
    """Qwen3Attention: q_norm / k_norm over head_dim, then apply_rotary_pos_emb (rotate_half) -- V untouched."""
    qkv = rnd(rows, (Hq + 2 * Hkv) * d, seed=3)
    wq, wk = (1 + 0.1 * rnd(d, seed=4).float()).to(bf), (1 + 0.1 * rnd(d, seed=5).float()).to(bf)
    pos = torch.arange(rows, dtype=torch.float32)
    inv = 1.0 / (1e6 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.cat([pos[:, None] * inv[None]] * 2, -1)
    cos, sin = fr.cos(), fr.sin()
    if not f32:
        cos, sin = cos.to(bf), sin.to(bf)
    x = qkv.float().view(rows, Hq + 2 * Hkv, d)
    ref = x.clone()
    for lo, hi, w in ((0, Hq, wq), (Hq, Hq + Hkv, wk)):
        h = x[:, lo:hi]
        if norm:
            h = ((h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + 1e-6)).to(bf).float() * w.float()).to(bf).float()
        rot = torch.cat((-h[..., d // 2:], h[..., :d // 2]), -1)
        ref[:, lo:hi] = h * cos.float()[:, None] + rot * sin.float()[:, None]
    got = qkv.to(D)
    ops.qk_norm_rope(got, wq.to(D) if norm else None, wk.to(D) if norm else None, cos.to(D), sin.to(D), Hq, Hkv, d, 1e-6)
    close_bf16(got, ref.reshape(rows, -1))
    assert torch.equal(got[:, (Hq + Hkv) * d:].cpu(), qkv[:, (Hq + Hkv) * d:])
    # the KV-cache form: the same values in place, and keys / values once more as dense (batch, kv heads, S, d) tensors
    for S in {rows, rows // 3 if rows % 3 == 0 else rows}:
        got2 = qkv.to(D)
        r = ops.qk_norm_rope(got2, wq.to(D) if norm else None, wk.to(D) if norm else None, cos.to(D), sin.to(D), Hq, Hkv, d,
                             1e-6, kv_cache_seq=S)
        assert torch.equal(r[0], got)
        g4 = got.view(rows // S, S, Hq + 2 * Hkv, d)
        assert r[1].is_contiguous() and r[1].shape == (rows // S, Hkv, S, d)
        assert torch.equal(r[1], g4[:, :, Hq:Hq + Hkv].transpose(1, 2))
        assert torch.equal(r[2], g4[:, :, Hq + Hkv:].transpose(1, 2))
        # ... and into the middle of larger (batch, kv heads, capacity, d) buffers: an append-in-place cache
        cap, pos = S + 7, 3
        kb = torch.full((rows // S, Hkv, cap, d), 7.0, dtype=bf, device=D)
        vb = torch.full((rows // S, Hkv, cap, d), 7.0, dtype=bf, device=D)
        got3 = qkv.to(D)
        ops.qk_norm_rope(got3, wq.to(D) if norm else None, wk.to(D) if norm else None, cos.to(D), sin.to(D), Hq, Hkv, d, 1e-6,
                         kv_cache_seq=S, kv_out=(kb, vb), kv_pos=pos)
        assert torch.equal(kb[:, :, pos:pos + S], r[1]) and torch.equal(vb[:, :, pos:pos + S], r[2])
        assert (kb[:, :, :pos] == 7).all() and (kb[:, :, pos + S:] == 7).all() and (vb[:, :, pos + S:] == 7).all()


def test_swiglu(ops):
    gu = rnd(300, 2 * 1536, scale=2.0, seed=6)
    g, u = gu[:, :1536].float(), gu[:, 1536:].float()
    close_bf16(ops.swiglu(gu.to(D)), F.silu(g).to(bf).float() * u)


@pytest.mark.parametrize("rows,K,I", [(1024, 4096, 12288), (300, 512, 1536), (257, 128, 1040), (7, 192, 16), (640, 1024, 96),
                                      (1, 4096, 12288), (16, 512, 1536), (3, 2048, 48)])   # <= 16 rows: the few-rows kernel
def test_gemm_swiglu_pair_equals_the_two_step_form(ops, rows, K, I):
    """u2tok_gemm_bf16 flag 512: SiLU(gate) * up in the epilogue of the packed gate | up product -- bit for bit the values of the
    GEMM followed by u2tok_swiglu_bf16 (same accumulation order, same rounding points), tiles that straddle M and I included."""
    from u2tokenizer_amd import prefill
    prefill._scratch.clear()            # (an ambient split-K scratch left by a fused prefill would slice the plain product's K
    ops.set_gemm_scratch(None)          #  loop: fp32 partial sums in another order -- equal to rounding, not bit for bit)
    x = rnd(rows, K, seed=11).to(D)
    w = rnd(2 * I, K, scale=2.0 / math.sqrt(K), seed=12).to(D)
    assert ops.gemm_swiglu_supported(rows, K, I)
    two = ops.swiglu(ops.gemm(x, w))
    one = ops.gemm_swiglu(x, w)
    assert one.shape == two.shape == (rows, I)
    assert torch.equal(one, two), (one.float() - two.float()).abs().max()
    g, u = (x.float() @ w[:I].float().T).to(bf).float(), (x.float() @ w[I:].float().T).to(bf).float()
    close_bf16(one, (F.silu(g).to(bf).float() * u).cpu(), rounds=4)
    assert not ops.gemm_swiglu_supported(rows, K + 8, I) and not ops.gemm_swiglu_supported(rows, K, I + 8)


@pytest.mark.parametrize("nb,Sq,Skv,Hq,Hkv,d", [(1, 1024, 1024, 32, 8, 128), (2, 77, 77, 8, 4, 64), (1, 50, 50, 4, 4, 128),
                                                (1, 200, 200, 32, 8, 64), (1, 40, 100, 8, 2, 128), (3, 1, 1, 2, 1, 64)])
def test_attention_gqa_causal(ops, nb, Sq, Skv, Hq, Hkv, d):
    """Grouped-query causal attention (the prefill's attention; Skv > Sq = a query block at the end of a longer key range)
    against torch on the same bf16 inputs in fp32; q / k / v are column slices of one packed buffer, as prefill.py passes them."""
    if Sq == Skv:
        buf = rnd(nb, Sq, (Hq + 2 * Hkv) * d, seed=Sq + d)
        q, k, v = buf[..., :Hq * d], buf[..., Hq * d:(Hq + Hkv) * d], buf[..., (Hq + Hkv) * d:]
        dbuf = buf.to(D)
        dq, dk, dv = dbuf[..., :Hq * d], dbuf[..., Hq * d:(Hq + Hkv) * d], dbuf[..., (Hq + Hkv) * d:]
    else:
        q, kv = rnd(nb, Sq, Hq * d, seed=1), rnd(nb, Skv, 2 * Hkv * d, seed=2)
        k, v = kv[..., :Hkv * d], kv[..., Hkv * d:]
        dq, dkv = q.to(D), kv.to(D)
        dk, dv = dkv[..., :Hkv * d], dkv[..., Hkv * d:]
    scale = 1.5 / math.sqrt(d)
    got = [ops.attention_gqa(dq, dk, dv, Hq, Hkv, scale, causal=True) for _ in range(2)]
    assert torch.equal(got[0], got[1])
    qh = q.float().view(nb, Sq, Hq, d).transpose(1, 2)
    kh = k.float().view(nb, Skv, Hkv, d).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    vh = v.float().view(nb, Skv, Hkv, d).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    s = qh @ kh.transpose(-1, -2) * scale
    i, j = torch.arange(Sq)[:, None], torch.arange(Skv)[None, :]
    s = s.masked_fill(j > i + (Skv - Sq), float("-inf"))
    ref = (F.softmax(s, -1) @ vh).transpose(1, 2).reshape(nb, Sq, Hq * d)
    close_bf16(got[0], ref)
    # and without the mask (grouped heads only)
    ref2 = (F.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(nb, Sq, Hq * d)
    close_bf16(ops.attention_gqa(dq, dk, dv, Hq, Hkv, scale, causal=False), ref2)


def _small(kind, layers=3, wide=False):
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen3Config, Qwen3ForCausalLM
    common = dict(vocab_size=1024, hidden_size=512, intermediate_size=1536, num_hidden_layers=layers, num_attention_heads=8,
                  num_key_value_heads=4, head_dim=64, max_position_embeddings=512, tie_word_embeddings=False,
                  pad_token_id=0, bos_token_id=1, eos_token_id=2)
    if wide:  # one layer at the Qwen3-8B width: the kernel variants the real decoder takes (E = 4096, I = 12288, 32 / 8 heads of 128)
        common.update(hidden_size=4096, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8, head_dim=128)
    if kind == "qwen3":
        m = Qwen3ForCausalLM(Qwen3Config(**common))
    else:
        m = LlamaForCausalLM(LlamaConfig(**common, rope_theta=500000.0))
    synth.fill_module_(m, seed=17, prefix="decoder.")
    return m.eval()


def _err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize("kind,wide", [("qwen3", False), ("llama", False), ("qwen3", True)])
def test_fused_prefill_matches_the_stock_decoder(ops, kind, wide):
    """Logits and the KV cache of a prefill through the patched layers: no further from the fp32 model than 1.5 x the stock bf16
    GPU run is; a padded batch must take the stock layers (bit-identical to the unpatched model).  wide: two layers at the
    Qwen3-8B width over the path's 1024 spliced embeddings -- the kernel variants the real prefill takes (q|k|v and gate|up on
    the big-tile kernel in pair / K-sliced form at M = 1024, 64-key causal tiles at head dim 128)."""
    from u2tokenizer_amd.prefill import disable_fused_prefill, enable_fused_prefill
    nl, B, S, E = (2, 1, 1024, 4096) if wide else (3, 2, 70, 512)
    m32 = _small(kind, nl, wide)
    x = 0.5 * synth.synth_tensor("inputs_embeds", (B, S, E), 3)
    ref = m32(inputs_embeds=x, use_cache=True)
    mg = _small(kind, nl, wide).to(bf).to(D)
    xd = x.to(bf).to(D)
    stock = mg(inputs_embeds=xd, use_cache=True)
    assert enable_fused_prefill(mg) == nl
    fused = mg(inputs_embeds=xd, use_cache=True)
    e_stock, e_fused = _err(stock.logits.float().cpu(), ref.logits), _err(fused.logits.float().cpu(), ref.logits)
    assert e_fused <= 1.5 * e_stock + EPS, (e_fused, e_stock)
    assert not torch.equal(fused.logits, stock.logits)          # (it really took another code path)
    for li in (0, nl - 1):
        for name in ("keys", "values"):
            r = getattr(ref.past_key_values.layers[li], name)
            es = _err(getattr(stock.past_key_values.layers[li], name).float().cpu(), r)
            ef = _err(getattr(fused.past_key_values.layers[li], name).float().cpu(), r)
            assert ef <= 1.5 * es + EPS, (li, name, ef, es)
    mask = torch.ones((B, S), dtype=torch.int64, device=D)
    mask[B - 1, :5] = 0
    padded = mg(inputs_embeds=xd, attention_mask=mask, use_cache=True)
    disable_fused_prefill(mg)
    assert torch.equal(padded.logits, mg(inputs_embeds=xd, attention_mask=mask, use_cache=True).logits)


class _LoraLikeLinear(torch.nn.Module):
    """What peft's lora.Linear looks like from outside: `.weight` / `.bias` are the BASE layer's, forward adds the adapter."""

    def __init__(self, base, rank=4):
        super().__init__()
        self.base_layer = base
        g = torch.Generator().manual_seed(3)
        self.lora_A = torch.nn.Parameter(0.05 * torch.randn(rank, base.in_features, generator=g).to(base.weight))
        self.lora_B = torch.nn.Parameter(0.05 * torch.randn(base.out_features, rank, generator=g).to(base.weight))

    weight = property(lambda self: self.base_layer.weight)
    bias = property(lambda self: self.base_layer.bias)

    def forward(self, x):
        return self.base_layer(x) + (x @ self.lora_A.t()) @ self.lora_B.t()


def test_fused_prefill_leaves_wrapped_or_hooked_layers_stock(ops):
    """The reference trains the decoder with LoRA (train_stage1.py:342-353).  A layer whose projection is not exactly
    nn.Linear (an unmerged adapter), or that carries a forward hook, must run its own forward: outputs bit-identical to the
    unpatched model; the untouched layers still take the fused path."""
    from u2tokenizer_amd.prefill import disable_fused_prefill, enable_fused_prefill
    mg = _small("qwen3").to(bf).to(D)
    xd = (0.5 * synth.synth_tensor("inputs_embeds", (1, 70, 512), 3)).to(bf).to(D)
    plain = mg(inputs_embeds=xd).logits
    lay = mg.model.layers[1]
    lay.self_attn.q_proj = _LoraLikeLinear(lay.self_attn.q_proj).to(D)
    seen = []
    handle = mg.model.layers[2].mlp.register_forward_hook(lambda m, a, out: seen.append(out.shape))
    want = mg(inputs_embeds=xd).logits                     # adapter + hook active, stock layers
    assert not torch.equal(want, plain) and len(seen) == 1
    assert enable_fused_prefill(mg) == 3
    got = mg(inputs_embeds=xd).logits
    assert len(seen) == 2                                  # the hook fired: that layer ran its own forward
    # layer 0 is fused, layers 1 and 2 stock: close to, not equal to, the all-stock run; the adapter's effect is in the result
    assert _err(got.float(), want.float()) < 2e-2 and _err(got.float(), plain.float()) > 0.5 * _err(want.float(), plain.float())
    handle.remove()
    lay.self_attn.q_proj = lay.self_attn.q_proj.base_layer
    again = mg(inputs_embeds=xd).logits                    # adapter merged away, hook gone: all three layers fused again
    disable_fused_prefill(mg)
    assert _err(again.float(), plain.float()) < 2e-2 and not torch.equal(again, plain)


@pytest.mark.parametrize("nb,T,g,d", [(8, 1100, 4, 128), (3, 70, 2, 64), (16, 1792, 4, 128), (2, 5, 1, 128)])
def test_attention_gqa_split_keys_single_query_row(ops, nb, T, g, d):
    """The decode step's attention: one query row per (batch x kv head) entry, g query heads on one K / V head, the keys split
    over workgroups and merged in a fixed order (u2tok_attention_gqa_split) -- against the softmax in fp32; repeatable."""
    q = rnd(nb, 1, g * d, seed=21)
    k, v = rnd(nb, T, d, seed=22), rnd(nb, T, d, seed=23)
    sc = (q.float().view(nb, g, d) @ k.float().transpose(1, 2)) * d ** -0.5      # (nb, g, T)
    ref = (torch.softmax(sc, -1) @ v.float()).reshape(nb, 1, g * d)
    outs = [ops.attention_gqa(q.to(D), k.to(D), v.to(D), g, 1, d ** -0.5, causal=False, split_keys=True) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    close_bf16(outs[0], ref, rounds=3)


@pytest.mark.parametrize("kind,B,wide", [("qwen3", 1, False), ("llama", 2, False), ("qwen3", 2, True)])
def test_fused_decode_step_matches_the_stock_decoder(ops, kind, B, wide):
    """One decode step after a prefill (both through the patched layers, then the same two calls through the stock layers):
    logits and the new cache entries no further from the fp32 model than 1.5 x the stock bf16 GPU run is; with decode=False the
    step takes the stock layers (bit-identical to an unpatched model on the same cache)."""
    from u2tokenizer_amd.prefill import disable_fused_prefill, enable_fused_prefill
    nl, E = (1, 4096) if wide else (3, 512)
    m32 = _small(kind, nl, wide)
    x = 0.5 * synth.synth_tensor("inputs_embeds", (B, 40, E), 7)
    x1 = 0.5 * synth.synth_tensor("inputs_embeds", (B, 1, E), 8)
    p32 = m32(inputs_embeds=x, use_cache=True)
    ref = m32(inputs_embeds=x1, past_key_values=p32.past_key_values, use_cache=True)
    mg = _small(kind, nl, wide).to(bf).to(D)
    xd, x1d = x.to(bf).to(D), x1.to(bf).to(D)
    ps = mg(inputs_embeds=xd, use_cache=True)
    stock = mg(inputs_embeds=x1d, past_key_values=ps.past_key_values, use_cache=True)
    enable_fused_prefill(mg)
    pf = mg(inputs_embeds=xd, use_cache=True)
    fused = mg(inputs_embeds=x1d, past_key_values=pf.past_key_values, use_cache=True)
    assert type(pf.past_key_values.layers[0]).__name__ == "AppendLayer"          # the in-place cache layer took the prompt
    assert fused.logits.shape == stock.logits.shape == (B, 1, ref.logits.shape[-1])
    e_stock, e_fused = _err(stock.logits.float().cpu(), ref.logits), _err(fused.logits.float().cpu(), ref.logits)
    assert e_fused <= 1.5 * e_stock + EPS, (e_fused, e_stock)
    assert not torch.equal(fused.logits, stock.logits)
    for li in (0, nl - 1):
        for name in ("keys", "values"):
            r = getattr(ref.past_key_values.layers[li], name)
            gk = getattr(fused.past_key_values.layers[li], name)
            assert gk.shape == r.shape and gk.shape[2] == 41
            es = _err(getattr(stock.past_key_values.layers[li], name).float().cpu(), r)
            assert _err(gk.float().cpu(), r) <= 1.5 * es + EPS, (li, name)
    disable_fused_prefill(mg)
    enable_fused_prefill(mg, decode=False)
    pf2 = mg(inputs_embeds=xd, use_cache=True)
    only_prefill = mg(inputs_embeds=x1d, past_key_values=pf2.past_key_values, use_cache=True)
    disable_fused_prefill(mg)
    pf3 = mg(inputs_embeds=xd, use_cache=True)          # stock prefill, then compare a stock step on the fused prefill's cache
    assert torch.isfinite(only_prefill.logits).all() and pf3.logits.shape == pf2.logits.shape


def test_generate_prefills_fused_and_decodes_on_the_same_cache(ops):
    """HF generate over the patched model: the prefill AND every decode step go through the HIP layers, on the HF cache
    (`decode=False`: the steps take the stock layers on the cache the fused prefill filled -- checked as well).  Greedy ids equal the unpatched model's unless the fp32 model's own top-2 margin at that step
    is below the bf16 noise (a bf16 run may legitimately flip such an argmax)."""
    from u2tokenizer_amd.prefill import enable_fused_prefill
    m32 = _small("qwen3", layers=2)
    x = 0.5 * synth.synth_tensor("inputs_embeds", (1, 48, 512), 5)
    new = 6
    g32 = m32.generate(inputs_embeds=x, max_new_tokens=new, do_sample=False, output_scores=True, return_dict_in_generate=True)
    mg = _small("qwen3", layers=2).to(bf).to(D)
    g_stock = mg.generate(inputs_embeds=x.to(bf).to(D), max_new_tokens=new, do_sample=False).cpu()
    from u2tokenizer_amd.prefill import disable_fused_prefill
    for decode in (True, False):
        enable_fused_prefill(mg, decode=decode)
        g_fused = mg.generate(inputs_embeds=x.to(bf).to(D), max_new_tokens=new, do_sample=False).cpu()
        disable_fused_prefill(mg)
        assert g_fused.shape == g_stock.shape == g32.sequences.shape
        for t in range(new):
            top2 = g32.scores[t][0].topk(2).values
            if (top2[0] - top2[1]).item() > 0.05:
                assert g_fused[0, t] == g32.sequences[0, t] == g_stock[0, t], (decode, t, g_fused, g_stock, g32.sequences)
            else:
                break   # past an ambiguous step the continuations may differ legitimately
