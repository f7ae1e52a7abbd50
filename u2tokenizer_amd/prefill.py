"""Decoder prefill on the spliced embeddings through the HIP kernels (the step AFTER the path:
`LlamaForCausalLM / Qwen3ForCausalLM.forward(inputs_embeds=...)`, /root/reference/src/model/language_model/u2llama.py:76-87,
and the first forward of `generate`, u2llama.py:123-126).

The decoder stays the stock HuggingFace module tree: its parameters (names, shapes, state dict), its KV cache and its
`generate` loop are untouched.  `enable_fused_prefill(model)` replaces the `forward` of every decoder layer by one that,
for the PREFILL call only (no grad, bf16 on the GPU, more than one position, empty cache for that layer, full attention, no
padding), runs the layer as

    RMSNorm -> ONE q|k|v GEMM -> per-head RMSNorm (Qwen3) + rotary embedding -> causal grouped-query attention
    -> out-projection GEMM with the residual in its epilogue -> RMSNorm -> ONE gate|up GEMM with SiLU(gate) * up in its
    epilogue -> down-projection GEMM with the residual in its epilogue

on the library's MFMA GEMMs (include/u2tok.h: u2tok_gemm_bf16), the fused attention kernel of tokattn.hip
(u2tok_attention_gqa: grouped-query heads, causal mask, scores never in HBM) and the row kernels of decoder.hip.  Everything
else -- decode steps, training, CPU tensors, sliding-window layers, padded batches -- takes the layer's original forward.
q|k|v and gate|up are packed the way the tokenizer packs its projections: the nn.Parameters keep their names and shapes,
their storage becomes a view of one buffer, so the stock modules keep working on them.
"""
from __future__ import annotations

import types
from typing import Optional

import torch

from . import ops


def _pack(linears):
    """Lay the weights (and biases, if any) of `linears` back to back in one buffer; returns (W, b | None)."""
    ws = [lin.weight for lin in linears]
    es = ws[0].element_size()
    st0 = ws[0].untyped_storage()
    adjacent = all(w.is_contiguous() and w.untyped_storage().data_ptr() == st0.data_ptr() for w in ws) and all(
        ws[i + 1].data_ptr() == ws[i].data_ptr() + ws[i].numel() * es for i in range(len(ws) - 1))  # (one storage: neighbours
    # in two allocations are not a packed buffer)
    if not adjacent:
        W = torch.cat([w.data for w in ws], 0).contiguous()
        o = 0
        for w in ws:
            w.data = W[o:o + w.shape[0]]
            o += w.shape[0]
    W = ws[0].data.as_strided((sum(w.shape[0] for w in ws), ws[0].shape[1]), (ws[0].shape[1], 1))
    b = None
    if all(lin.bias is not None for lin in linears):
        b = torch.cat([lin.bias.data for lin in linears], 0).contiguous()  # (biases are tiny: a copy, refreshed per call)
    elif any(lin.bias is not None for lin in linears):
        raise RuntimeError("mixed bias / no-bias projections cannot be packed")
    return W, b


_scratch = {}


def _ensure_gemm_scratch(device) -> None:
    """Split-K scratch of the calling stream (at S = 1024 the out projection is 256 tiles of 128 x 128: cut in two along K it puts
    two workgroups on every CU; q|k|v and the down projection take the big-tile kernel in 2 / 4 K slices, gemm_bt.hip:
    bt_pick_sliced), registered on the active context like the training path's."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, id(ops.active_context(device)))
    if key not in _scratch:
        buf = torch.empty(72 << 20, dtype=torch.uint8, device=device)  # 4 slices of (1024, 4096) fp32 sums + slack
        with torch.cuda.device(device):
            ops.set_gemm_scratch(buf)
        _scratch[key] = buf


def _layer_forward(self, hidden_states, *args, **kwargs):
    st = self._u2_prefill
    x = hidden_states
    pe = kwargs.get("position_embeddings")
    cache = kwargs.get("past_key_values")
    att = self.self_attn
    common = (not args and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3
              and pe is not None and st["owner"]._u2_prefill_mask_ok
              and getattr(att, "sliding_window", None) is None and att.head_dim in (64, 128))
    if common and x.shape[1] == 1 and x.shape[0] <= 16 and st["owner"]._u2_fused_decode \
            and _plain_dynamic_layer(cache, att.layer_idx) is not None:
        return _decode_step(self, x, pe, cache)
    fused = common and x.shape[1] > 1 and (cache is None or cache.get_seq_length(att.layer_idx) == 0)
    if not fused:
        return st["orig"](hidden_states, *args, **kwargs)
    B, S, E = x.shape
    rows = B * S
    cfg = att.config
    Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, att.head_dim
    _ensure_gemm_scratch(x.device)
    with ops.on_device(x):
        Wqkv, bqkv = _pack((att.q_proj, att.k_proj, att.v_proj))
        Wgu, bgu = _pack((self.mlp.gate_proj, self.mlp.up_proj))
        x2 = x.reshape(rows, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        cos, sin = pe
        cos = cos.expand(B, S, d).reshape(rows, d)
        sin = sin.expand(B, S, d).reshape(rows, d)
        xn = ops.rmsnorm(x2, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        qkv = ops.gemm(xn, Wqkv, bias=bqkv)
        qn, kn = getattr(att, "q_norm", None), getattr(att, "k_norm", None)
        r = ops.qk_norm_rope(qkv, None if qn is None else qn.weight, None if kn is None else kn.weight, cos, sin, Hq, Hkv, d,
                             qn.variance_epsilon if qn is not None else 1e-6, kv_cache_seq=S if cache is not None else 0)
        kc, vc = (r[1], r[2]) if cache is not None else (None, None)  # keys / values in the cache's own (B, H_kv, S, d) layout
        q3 = qkv.view(B, S, -1)
        k3, v3 = q3[..., Hq * d:(Hq + Hkv) * d], q3[..., (Hq + Hkv) * d:]
        ctx = ops.attention_gqa(q3[..., :Hq * d], k3, v3, Hq, Hkv, float(att.scaling), causal=True)
        h = ops.gemm(ctx.view(rows, Hq * d), att.o_proj.weight, bias=att.o_proj.bias, residual=x2)
        hn = ops.rmsnorm(h, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
        if bgu is None and ops.gemm_swiglu_supported(rows, E, Wgu.shape[0] // 2):
            act = ops.gemm_swiglu(hn, Wgu)  # SiLU(gate) * up in the epilogue of the pair product: no (rows, 2 I) tensor
        else:
            act = ops.swiglu(ops.gemm(hn, Wgu, bias=bgu))
        out = ops.gemm(act, self.mlp.down_proj.weight, bias=self.mlp.down_proj.bias, residual=h)
        if cache is not None:
            _cache_prefill(cache, kc, vc, att.layer_idx)
    return out.view(B, S, E)


def _plain_dynamic_layer(cache, layer_idx: int):
    """The cache layer if `cache` is a plain HF DynamicCache (no offloading) whose layer `layer_idx` is a non-empty DynamicLayer
    -- the case the fused decode step handles (its `update` is a torch.cat: dense (B, H_kv, T, d) tensors come back)."""
    try:
        from transformers.cache_utils import DynamicCache, DynamicLayer
    except ImportError:
        return None
    layers = getattr(cache, "layers", None)
    if type(cache) is not DynamicCache or not isinstance(layers, list) or getattr(cache, "offloading", False) \
            or layer_idx >= len(layers):
        return None
    lay = layers[layer_idx]
    return lay if type(lay) is DynamicLayer and lay.get_seq_length() > 0 else None


def _decode_step(self, x, pe, cache):
    """One decode step of a layer (B <= 16 new tokens, one each, against the KV cache): the step `generate` repeats up to 768
    times per report (eval/mrg.py:74-77).  Every product is weight streaming -- q|k|v, out, gate|up and down go through the
    few-rows GEMM (gemm.hip: gemm_rows16_kernel, all loads of a wave in flight before its first MFMA) --, the attention is the
    fused kernel with the KEYS split over workgroups (u2tok_attention_gqa_split: batch x kv-head entries of (T, d) keys, the
    query heads of a group as its heads), 10 launches per layer instead of the stock layer's ~40."""
    att = self.self_attn
    B, _, E = x.shape
    cfg = att.config
    Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, att.head_dim
    g = Hq // Hkv
    with ops.on_device(x):
        Wqkv, bqkv = _pack((att.q_proj, att.k_proj, att.v_proj))
        Wgu, bgu = _pack((self.mlp.gate_proj, self.mlp.up_proj))
        x2 = x.reshape(B, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        cos, sin = pe
        cos = cos.expand(B, 1, d).reshape(B, d)
        sin = sin.expand(B, 1, d).reshape(B, d)
        xn = ops.rmsnorm(x2, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        qkv = ops.gemm(xn, Wqkv, bias=bqkv)
        qn, kn = getattr(att, "q_norm", None), getattr(att, "k_norm", None)
        _, kc, vc = ops.qk_norm_rope(qkv, None if qn is None else qn.weight, None if kn is None else kn.weight, cos, sin, Hq, Hkv,
                                     d, qn.variance_epsilon if qn is not None else 1e-6, kv_cache_seq=1)
        K, V = cache.update(kc, vc, att.layer_idx)          # DynamicLayer: torch.cat -> dense (B, H_kv, T, d)
        if not K.is_contiguous():
            K = K.contiguous()
        if not V.is_contiguous():
            V = V.contiguous()
        T = K.shape[2]
        q = qkv[:, :Hq * d].reshape(B * Hkv, 1, g * d)       # query heads i g .. i g + g - 1 read kv head i (repeat_kv's order)
        ctx = ops.attention_gqa(q, K.view(B * Hkv, T, d), V.view(B * Hkv, T, d), g, 1, float(att.scaling), causal=False,
                                split_keys=True)
        h = ops.gemm(ctx.view(B, Hq * d), att.o_proj.weight, bias=att.o_proj.bias, residual=x2)
        hn = ops.rmsnorm(h, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
        act = ops.swiglu(ops.gemm(hn, Wgu, bias=bgu))
        out = ops.gemm(act, self.mlp.down_proj.weight, bias=self.mlp.down_proj.bias, residual=h)
    return out.view(B, 1, E)


def _cache_prefill(cache, keys, values, layer_idx: int) -> None:
    """Hand the prefill's keys / values (fresh dense (B, H_kv, S, d) tensors nobody else holds) to the KV cache.  An EMPTY
    `DynamicLayer` of a plain `DynamicCache` would only `torch.cat` them onto its empty tensors (two more copies per layer: 72
    launches per Qwen3-8B prefill) -- it takes them as they are; every other cache type goes through its `update`."""
    try:
        from transformers.cache_utils import DynamicCache, DynamicLayer
    except ImportError:  # (older transformers: no per-layer cache objects)
        DynamicCache = DynamicLayer = None
    layers = getattr(cache, "layers", None)
    if DynamicLayer is not None and type(cache) is DynamicCache and isinstance(layers, list) \
            and not getattr(cache, "offloading", False):
        repl = getattr(cache, "layer_class_to_replicate", None)
        if repl is DynamicLayer:
            while len(layers) <= layer_idx:
                layers.append(repl())
        lay = layers[layer_idx] if layer_idx < len(layers) else None
        if type(lay) is DynamicLayer and lay.get_seq_length() == 0 and hasattr(lay, "lazy_initialization"):
            try:
                if not getattr(lay, "is_initialized", False):
                    lay.lazy_initialization(keys, values)
                lay.keys, lay.values = keys, values
                return
            except TypeError:  # (another transformers version's signature: let the cache do it its way)
                pass
    cache.update(keys, values, layer_idx)


def _mask_hook(module, args, kwargs):
    """Forward pre-hook of the decoder stack: the fused layers assume no padding (the path's prompts are left-aligned and the
    reference evaluates at batch 1, eval/mrg.py:74); a 2-D mask with zeros sends the whole call to the stock layers."""
    m = kwargs.get("attention_mask")
    ok = m is None or (torch.is_tensor(m) and m.dim() == 2 and bool(m.to(torch.bool).all()))
    module._u2_prefill_mask_ok = ok
    return None


def enable_fused_prefill(model, decode: bool = True) -> int:
    """Patch the decoder layers of an HF Llama / Qwen3 causal LM (u2LlamaForCausalLM / u2Qwen3ForCausalLM included) for the
    fused prefill and (decode=True) the fused decode step.  Idempotent; returns the number of layers patched.
    `disable_fused_prefill` restores the stock forwards."""
    base = model.get_model() if hasattr(model, "get_model") else getattr(model, "model", model)
    layers = getattr(base, "layers", None)
    if layers is None:
        raise RuntimeError("enable_fused_prefill: no decoder layers found (expected an HF Llama / Qwen3 model)")
    n = 0
    for layer in layers:
        if hasattr(layer, "_u2_prefill"):
            continue
        needed = all(hasattr(layer, a) for a in ("self_attn", "mlp", "input_layernorm", "post_attention_layernorm")) and \
            all(hasattr(layer.self_attn, a) for a in ("q_proj", "k_proj", "v_proj", "o_proj", "head_dim", "scaling")) and \
            all(hasattr(layer.mlp, a) for a in ("gate_proj", "up_proj", "down_proj"))
        if not needed:
            raise RuntimeError(f"enable_fused_prefill: unsupported decoder layer {type(layer).__name__}")
        layer._u2_prefill = {"orig": layer.forward, "owner": base}
        layer.forward = types.MethodType(_layer_forward, layer)
        n += 1
    base._u2_fused_decode = bool(decode)
    if not hasattr(base, "_u2_prefill_hook"):
        base._u2_prefill_mask_ok = True
        base._u2_prefill_hook = base.register_forward_pre_hook(_mask_hook, with_kwargs=True)
    return n


def disable_fused_prefill(model) -> None:
    base = model.get_model() if hasattr(model, "get_model") else getattr(model, "model", model)
    for layer in base.layers:
        st = layer.__dict__.pop("_u2_prefill", None)
        if st is not None:
            layer.forward = st["orig"]
    hook = base.__dict__.pop("_u2_prefill_hook", None)
    if hook is not None:
        hook.remove()
