"""Decoder prefill on the spliced embeddings through the HIP kernels (the step AFTER the path:
`LlamaForCausalLM / Qwen3ForCausalLM.forward(inputs_embeds=...)`, /root/reference/src/model/language_model/u2llama.py:76-87,
and the first forward of `generate`, u2llama.py:123-126).

The decoder stays the stock HuggingFace module tree: its parameters (names, shapes, state dict), its KV cache and its
`generate` loop are untouched.  `enable_fused_prefill(model)` replaces the `forward` of every decoder layer by one that,
for the PREFILL call (no grad, bf16 on the GPU, more than one position, empty cache for that layer, full attention, no
padding) runs the layer as

    RMSNorm -> ONE q|k|v GEMM -> per-head RMSNorm (Qwen3) + rotary embedding -> causal grouped-query attention
    -> out-projection GEMM with the residual in its epilogue -> RMSNorm -> ONE gate|up GEMM with SiLU(gate) * up in its
    epilogue -> down-projection GEMM with the residual in its epilogue

on the library's MFMA GEMMs (include/u2tok.h: u2tok_gemm_bf16), the fused attention kernel of tokattn.hip
(u2tok_attention_gqa: grouped-query heads, causal mask, scores never in HBM) and the row kernels of decoder.hip, and for a
DECODE step (one new position per sequence, batch <= 16, a plain HF DynamicCache) as two library calls around the cache update
(`_decode_step`: weight-streaming few-rows products, attention with the keys split over workgroups).  Everything else --
training, CPU tensors, sliding-window layers, padded batches, other cache types -- takes the layer's original forward.
q|k|v and gate|up are packed the way the tokenizer packs its projections: the nn.Parameters keep their names and shapes,
their storage becomes a view of one buffer, so the stock modules keep working on them.
"""
from __future__ import annotations

import types

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


_HOOK_TABLES = ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")


def _is_stock(layer) -> bool:
    """The fused forwards read the projections' `.weight` / `.bias` and never CALL the submodules.  That is only the same
    computation while every projection is exactly torch.nn.Linear and nothing hangs on the modules that are skipped: a
    peft lora.Linear exposes `.weight` as its BASE weight (the reference trains the decoder with LoRA, train_stage1.py:342-353:
    an unmerged adapter would be silently ignored), forward hooks (output_attentions recorders, activation probes) would not
    fire.  Checked on every call: adapters and hooks come and go after enable_fused_prefill."""
    att, mlp = layer.self_attn, layer.mlp
    for m in (att.q_proj, att.k_proj, att.v_proj, att.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj):
        if type(m) is not torch.nn.Linear:
            return False
    for m in (att.q_proj, att.k_proj, att.v_proj, att.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj, att, mlp,
              layer.input_layernorm, layer.post_attention_layernorm):
        for t in _HOOK_TABLES:
            if getattr(m, t, None):
                return False
    return True


def _layer_forward(self, hidden_states, *args, **kwargs):
    st = self._u2_prefill
    x = hidden_states
    pe = kwargs.get("position_embeddings")
    cache = kwargs.get("past_key_values")
    att = self.self_attn
    # `past_key_value` (singular) is the layer protocol of transformers 4.46 .. 4.5x, whose layers also return tuples: never
    # patched (enable_fused_prefill checks the signature), and a caller that passes it anyway gets the stock layer
    common = (not args and "past_key_value" not in kwargs and not kwargs.get("output_attentions")
              and not torch.is_grad_enabled() and x.is_cuda and x.dtype in ops.ELEM_OF and x.dim() == 3
              and att.q_proj.weight.dtype == x.dtype      # (bf16 weights under an fp16 autocast hand fp16 activations on: stock layers)
              and pe is not None and st["owner"]._u2_prefill_mask_ok and _is_stock(self)
              and getattr(att, "sliding_window", None) is None and att.head_dim in (64, 128))
    if common and x.shape[1] == 1 and x.shape[0] <= 16 and st["owner"]._u2_fused_decode \
            and _plain_dynamic_layer(cache, att.layer_idx) is not None:
        out = _decode_step(self, x, pe, cache)
        if out is not None:
            return out
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
        # keys / values in the cache's own (B, H_kv, S, d) layout: straight into an append-in-place layer of a plain DynamicCache
        # (room for the decode steps behind them), else as dense tensors for the cache's own `update`
        lay = _prefill_append_layer(cache, att.layer_idx, B, Hkv, S, d, x) if cache is not None else None
        r = ops.qk_norm_rope(qkv, None if qn is None else qn.weight, None if kn is None else kn.weight, cos, sin, Hq, Hkv, d,
                             qn.variance_epsilon if qn is not None else 1e-6, kv_cache_seq=S if cache is not None else 0,
                             kv_out=None if lay is None else (lay._kb, lay._vb))
        kc, vc = (r[1], r[2]) if cache is not None and lay is None else (None, None)
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
        if lay is not None:
            lay._commit(S)
        elif cache is not None:
            cache.update(kc, vc, att.layer_idx)
    return out.view(B, S, E)


_APPEND_LAYER = None


def _append_layer_class():
    """A DynamicLayer that APPENDS IN PLACE: keys / values are views [:, :, :T] of buffers with room to grow (doubling), so a decode
    step writes one row instead of re-copying the whole cache (DynamicLayer.update is a torch.cat: two launches and 2 x T rows per
    layer and step).  Everything else -- crop, batch selection, beam reordering, the stock attention reading `.keys` -- is
    DynamicLayer's: those reassign `.keys` / `.values`, after which the next update re-homes them."""
    global _APPEND_LAYER
    if _APPEND_LAYER is None:
        from transformers.cache_utils import DynamicLayer

        class AppendLayer(DynamicLayer):
            _kb = _vb = None

            def _room(self, n: int, like: torch.Tensor) -> int:
                """Make sure `n` more positions fit behind the current ones; returns the current length."""
                T0 = self.keys.shape[-2] if self.is_initialized and self.keys.numel() else 0
                kb = self._kb
                homed = kb is not None and (T0 == 0 or (self.keys.data_ptr() == kb.data_ptr() and self.values.data_ptr() == self._vb.data_ptr()
                                                        and self.keys.shape[:2] == kb.shape[:2]))
                if not homed or T0 + n > kb.shape[-2] or kb.shape[:2] != like.shape[:2]:
                    cap = max(256, 2 * (T0 + n))
                    shape = (like.shape[0], like.shape[1], cap, like.shape[3])
                    nk = torch.empty(shape, dtype=like.dtype, device=like.device)
                    nv = torch.empty(shape, dtype=like.dtype, device=like.device)
                    if T0:
                        nk[:, :, :T0].copy_(self.keys)
                        nv[:, :, :T0].copy_(self.values)
                    self._kb, self._vb = nk, nv
                    self.keys, self.values = nk[:, :, :T0], nv[:, :, :T0]
                return T0

            def _commit(self, T: int) -> None:
                self.keys, self.values = self._kb[:, :, :T], self._vb[:, :, :T]

            def update(self, key_states, value_states, *args, **kwargs):
                if not self.is_initialized:
                    self.lazy_initialization(key_states, value_states)
                n = key_states.shape[-2]
                T0 = self._room(n, key_states)
                self._kb[:, :, T0:T0 + n].copy_(key_states)
                self._vb[:, :, T0:T0 + n].copy_(value_states)
                self._commit(T0 + n)
                return self.keys, self.values

        _APPEND_LAYER = AppendLayer
    return _APPEND_LAYER


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
    return lay if type(lay) in (DynamicLayer, _APPEND_LAYER) and lay.get_seq_length() > 0 else None


def _decode_state(self, B: int, device):
    """Per-layer constants of the decode step (weight pointers of the packed projections, the config struct), rebuilt when a
    weight moved; per-model scratch (workspace, q|k|v row, new cache entries) shared by all layers."""
    import ctypes as C
    from . import _lib
    st = self._u2_prefill
    att, mlp = self.self_attn, self.mlp
    key = (att.q_proj.weight.data_ptr(), mlp.gate_proj.weight.data_ptr(), att.o_proj.weight.data_ptr(),
           mlp.down_proj.weight.data_ptr(), B)
    d = st.get("dec")
    if d is None or d["key"] != key:
        cfg = att.config
        Hq, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, att.head_dim
        Wqkv, bqkv = _pack((att.q_proj, att.k_proj, att.v_proj))
        Wgu, bgu = _pack((mlp.gate_proj, mlp.up_proj))
        qn, kn = getattr(att, "q_norm", None), getattr(att, "k_norm", None)
        E, inter = Wqkv.shape[1], Wgu.shape[0] // 2
        c = _lib.DecodeConfig(B=B, E=E, Hq=Hq, Hkv=Hkv, D=hd, I=inter, eps=float(self.input_layernorm.variance_epsilon),
                              qk_eps=float(qn.variance_epsilon) if qn is not None else 1e-6, scale=float(att.scaling))
        ok = (E % 32 == 0 and inter % 32 == 0 and att.o_proj.weight.is_contiguous() and mlp.down_proj.weight.is_contiguous()
              and self.post_attention_layernorm.variance_epsilon == self.input_layernorm.variance_epsilon)
        p = ops._ptr
        d = {"key": key, "ok": ok, "cfg": c, "cfg_ref": C.byref(c), "keep": (Wqkv, bqkv, Wgu, bgu), "nq": (Hq + 2 * Hkv) * hd,
             "Hkv": Hkv, "hd": hd, "E": E,
             "pre": (p(self.input_layernorm.weight), p(Wqkv), p(bqkv), p(None if qn is None else qn.weight),
                     p(None if kn is None else kn.weight)),
             "post": (p(att.o_proj.weight), p(att.o_proj.bias), p(self.post_attention_layernorm.weight), p(Wgu), p(bgu),
                      p(mlp.down_proj.weight), p(mlp.down_proj.bias))}
        st["dec"] = d
    owner = st["owner"]
    # (per model, batch size AND stream: two generate() calls in flight on different streams must not share the step's scratch)
    stream = torch.cuda.current_stream(device).cuda_stream
    pool = owner.__dict__.setdefault("_u2_decode_scratch", {})
    edt = self.input_layernorm.weight.dtype      # bf16, or fp16 for a decoder loaded in float16 (the f16 build of the library)
    sc = pool.get((B, device, stream))
    if sc is not None and sc["qkv"].dtype != edt:
        sc = None
    if sc is None:
        if len(pool) >= 8:
            pool.clear()
        sc = {"B": B, "device": device, "ws": None, "T": 0,
              "qkv": torch.empty((B, d["nq"]), dtype=edt, device=device),
              "kc": torch.empty((B, d["Hkv"], 1, d["hd"]), dtype=edt, device=device),
              "vc": torch.empty((B, d["Hkv"], 1, d["hd"]), dtype=edt, device=device)}
        pool[(B, device, stream)] = sc
    return d, sc


def _decode_step(self, x, pe, cache):
    """One decode step of a layer (B <= 16 new tokens, one each, against the KV cache): the step `generate` repeats up to 768
    times per report (eval/mrg.py:74-77).  Every product is weight streaming -- q|k|v, out, gate|up and down go through the
    few-rows GEMM (gemm.hip: gemm_rows16_kernel, all loads of a wave in flight before its first MFMA) --, the attention is the
    fused kernel with the KEYS split over workgroups (batch x kv-head entries of (T, d) keys, the query heads of a group as its
    heads).  TWO library calls per layer (u2tok_decoder_decode_pre / _post, 10 launches) around the cache's own `update`: with a
    Python call per kernel the step was bound by the host (7.9 ms against ~4 ms of kernels)."""
    from . import _lib
    att = self.self_attn
    B, _, E = x.shape
    d, sc = _decode_state(self, B, x.device)
    if not d["ok"]:
        return None                                   # (the caller takes the stock layer)
    hd = d["hd"]
    with ops.on_device(x) as (h, stream):
        x2 = x.reshape(B, E)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        cos, sin = pe
        cos = cos.expand(B, 1, hd).reshape(B, hd)
        sin = sin.expand(B, 1, hd).reshape(B, hd)
        if cos.dtype != sin.dtype or cos.dtype not in (torch.float32, x.dtype) or cos.stride(1) != 1 or sin.stride(1) != 1 \
                or cos.stride(0) != sin.stride(0):
            cos, sin = cos.float().contiguous(), sin.float().contiguous()
        T1 = cache.get_seq_length(att.layer_idx) + 1
        if sc["ws"] is None or sc["T"] < T1:
            Tcap = max(2048, 2 * T1)                      # (grows geometrically: the workspace depends on T through the key splits)
            d["cfg"].B = B
            sc["ws"] = torch.empty(h.u2tok_decoder_decode_workspace_bytes(d["cfg_ref"], Tcap), dtype=torch.uint8, device=x.device)
            sc["T"] = Tcap
        ws, nws = sc["ws"].data_ptr(), sc["ws"].numel()
        lay = cache.layers[att.layer_idx]
        out = torch.empty((B, 1, E), dtype=x.dtype, device=x.device)
        if type(lay) is _APPEND_LAYER and lay._kb is not None and lay._kb.shape[0] == B:
            # append in place: the rotary kernel writes the step's keys / values at position T0 of the layer's buffers
            T0 = lay._room(1, sc["kc"])
            kb, vb = lay._kb, lay._vb
            _lib.check(h.u2tok_decoder_decode_pre(d["cfg_ref"], x2.data_ptr(), *d["pre"], cos.data_ptr(), sin.data_ptr(),
                                                  int(cos.dtype == torch.float32), cos.stride(0), sc["qkv"].data_ptr(),
                                                  kb.data_ptr(), vb.data_ptr(), kb.stride(1), T0, ws, nws, stream),
                       "u2tok_decoder_decode_pre")
            lay._commit(T0 + 1)
            _lib.check(h.u2tok_decoder_decode_post(d["cfg_ref"], x2.data_ptr(), sc["qkv"].data_ptr(), kb.data_ptr(), vb.data_ptr(),
                                                   T0 + 1, kb.stride(1), *d["post"], out.data_ptr(), ws, nws, stream),
                       "u2tok_decoder_decode_post")
            return out
        _lib.check(h.u2tok_decoder_decode_pre(d["cfg_ref"], x2.data_ptr(), *d["pre"], cos.data_ptr(), sin.data_ptr(),
                                              int(cos.dtype == torch.float32), cos.stride(0), sc["qkv"].data_ptr(),
                                              sc["kc"].data_ptr(), sc["vc"].data_ptr(), 0, 0, ws, nws, stream),
                   "u2tok_decoder_decode_pre")
        K, V = cache.update(sc["kc"], sc["vc"], att.layer_idx)   # DynamicLayer: torch.cat -> dense (B, H_kv, T, d)
        if not K.is_contiguous():
            K = K.contiguous()
        if not V.is_contiguous():
            V = V.contiguous()
        _lib.check(h.u2tok_decoder_decode_post(d["cfg_ref"], x2.data_ptr(), sc["qkv"].data_ptr(), K.data_ptr(), V.data_ptr(),
                                               K.shape[2], 0, *d["post"], out.data_ptr(), ws, nws, stream),
                   "u2tok_decoder_decode_post")
    return out


def _prefill_append_layer(cache, layer_idx: int, B: int, Hkv: int, S: int, d: int, like: torch.Tensor):
    """For a plain HF DynamicCache whose layer `layer_idx` is still empty: put an append-in-place layer there with room for S
    positions (and as many again for the decode steps) and return it; None for every other cache (its `update` is used)."""
    try:
        from transformers.cache_utils import DynamicCache, DynamicLayer
    except ImportError:  # (older transformers: no per-layer cache objects)
        return None
    layers = getattr(cache, "layers", None)
    if type(cache) is not DynamicCache or not isinstance(layers, list) or getattr(cache, "offloading", False):
        return None
    cls = _append_layer_class()
    if getattr(cache, "layer_class_to_replicate", None) is DynamicLayer:
        while len(layers) <= layer_idx:
            layers.append(DynamicLayer())
    if layer_idx >= len(layers) or type(layers[layer_idx]) not in (DynamicLayer, cls) or layers[layer_idx].get_seq_length() != 0:
        return None
    try:
        lay = cls()
        proto = torch.empty((B, Hkv, 0, d), dtype=like.dtype, device=like.device)
        lay.lazy_initialization(proto, proto)
        lay._room(S, torch.empty((B, Hkv, 1, d), dtype=like.dtype, device=like.device))
    except (TypeError, AttributeError):  # (another transformers version's layer protocol)
        return None
    layers[layer_idx] = lay
    return lay


def _mask_hook(module, args, kwargs):
    """Forward pre-hook of the decoder stack: the fused layers assume no padding (the path's prompts are left-aligned and the
    reference evaluates at batch 1, eval/mrg.py:74); a 2-D mask with zeros sends the whole call to the stock layers."""
    m = kwargs.get("attention_mask")
    ok = m is None or (torch.is_tensor(m) and m.dim() == 2 and bool(m.to(torch.bool).all()))
    module._u2_prefill_mask_ok = ok
    return None


_warned_protocol = [False]


def _layer_protocol_ok(layer) -> bool:
    """_layer_forward is written against the decoder-layer protocol of transformers >= 4.56 / 5.x: the cache arrives as the
    keyword `past_key_values`, rotary tables as `position_embeddings`, and the layer returns the hidden-state TENSOR.  From
    4.46 (the reference's pin) up to that change the keyword is `past_key_value` and layers return tuples: there the cache
    would never be updated and `layer_outputs[0]` would slice the batch.  Such layers are left stock."""
    import inspect
    import warnings
    try:
        sig = inspect.signature(layer.forward)
        ok = "past_key_values" in sig.parameters and "position_embeddings" in sig.parameters
        ret = sig.return_annotation
        if ok and ret is not inspect.Signature.empty and "tuple" in str(ret).lower():
            ok = False
    except (TypeError, ValueError):
        ok = False
    if not ok and not _warned_protocol[0]:
        _warned_protocol[0] = True
        import transformers
        warnings.warn(f"u2tokenizer_amd.prefill: decoder layer protocol of transformers {transformers.__version__} is not the one "
                      "the fused prefill is written for (past_key_values keyword, tensor return); layers stay stock")
    return ok


def enable_fused_prefill(model, decode: bool = True, strict: bool = True) -> int:
    """Patch the decoder layers of an HF Llama / Qwen3 causal LM (u2LlamaForCausalLM / u2Qwen3ForCausalLM included) for the
    fused prefill and (decode=True) the fused decode step.  Idempotent; returns the number of layers patched.  strict=False: a
    decoder layer of another layout is skipped instead of refused.
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
            if strict:
                raise RuntimeError(f"enable_fused_prefill: unsupported decoder layer {type(layer).__name__}")
            continue   # (another layer layout, e.g. Phi3's fused qkv_proj / gate_up_proj: stays stock)
        if not _layer_protocol_ok(layer):
            continue
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
    base.__dict__.pop("_u2_decode_scratch", None)
