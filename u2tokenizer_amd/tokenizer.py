"""u2Tokenizer (drop-in for /root/reference/src/model/u2tokenizer/{u2Tokenizer,svr,tta,rma,rope}.py).

The sub-modules keep the reference's attribute names, parameter shapes and initialisers, so state dicts are
interchangeable (including the never-read `linear_aggregator.wv/dense`, tta.py:47-48,62-65).  They only own
parameters: `u2Tokenizer.forward` passes their device pointers to `u2tok_tokenizer_forward`
(include/u2tok.h), which runs SVR (spatial + temporal relative attention x L, hard or differentiable token
selection, {1,2,4} multi-scale pooling with optional gates) and TTA (query self-attention, visual and text
cross-attention with post-LN x L, un-projected linear aggregation) as one launch sequence.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops

_ATTN_TYPES = {"rma": 0, "rope": 1}  # every other value: 2 = nn.MultiheadAttention, as in the reference


def _init_attn(m):
    for lin in (m.wq, m.wk, m.wv, m.dense):
        nn.init.xavier_uniform_(lin.weight)
        if lin.bias is not None:
            nn.init.zeros_(lin.bias)


class RelativeMultiheadAttention(nn.Module):
    """Parameters of rma.py:5-35 (relative bias table: (2*max_seq_len-1, heads))."""

    def __init__(self, d_model, num_heads, max_seq_len=512):
        super().__init__()
        assert d_model % num_heads == 0
        self.num_heads, self.d_model, self.depth, self.max_seq_len = num_heads, d_model, d_model // num_heads, max_seq_len
        self.wq = nn.Linear(d_model, d_model)
        self.wk = nn.Linear(d_model, d_model)
        self.wv = nn.Linear(d_model, d_model)
        self.dense = nn.Linear(d_model, d_model)
        self.relative_bias = nn.Parameter(torch.zeros(2 * max_seq_len - 1, num_heads))
        self._reset_parameters()

    def _reset_parameters(self):
        _init_attn(self)
        nn.init.zeros_(self.relative_bias)

    init_weights = _reset_parameters


class RotaryMultiheadAttention(nn.Module):
    """Parameters of rope.py:16-60 (cos/sin caches are non-persistent buffers there; recomputed on device here)."""

    def __init__(self, d_model, num_heads, max_seq_len=512):
        super().__init__()
        assert d_model % num_heads == 0, "d_model must be divisible by num_heads"
        self.num_heads, self.d_model, self.head_dim, self.max_seq_len = num_heads, d_model, d_model // num_heads, max_seq_len
        self.wq = nn.Linear(d_model, d_model)
        self.wk = nn.Linear(d_model, d_model)
        self.wv = nn.Linear(d_model, d_model)
        self.dense = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        _init_attn(self)


class MultiHeadCrossAttention(nn.Module):
    """Parameters of tta.py:7-40."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        assert d_model % num_heads == 0
        self.num_heads, self.d_model, self.depth = num_heads, d_model, d_model // num_heads
        self.wq = nn.Linear(d_model, d_model)
        self.wk = nn.Linear(d_model, d_model)
        self.wv = nn.Linear(d_model, d_model)
        self.dense = nn.Linear(d_model, d_model)
        _init_attn(self)


def _self_attention(embed_size, num_heads, attn_type):
    if attn_type == "rma":
        return RelativeMultiheadAttention(embed_size, num_heads)
    if attn_type == "rope":
        return RotaryMultiheadAttention(embed_size, num_heads)
    # svr.py:16-18 / tta.py:83-84: any other value builds the stock module (the "linvt" ablation checkpoints);
    # it only holds the parameters here (in_proj_weight (3E, E), in_proj_bias, out_proj.{weight,bias})
    return nn.MultiheadAttention(embed_size, num_heads)


class SpatioTemporalAttentionLayer(nn.Module):
    def __init__(self, embed_size, num_heads, attn_type="rma"):
        super().__init__()
        self.spatial_attention = _self_attention(embed_size, num_heads, attn_type)
        self.temporal_attention = _self_attention(embed_size, num_heads, attn_type)


class SpatioTemporalSignificanceScoring(nn.Module):
    def __init__(self, embed_size, num_heads, num_layers, attn_type="rma"):
        super().__init__()
        self.layers = nn.ModuleList(
            [SpatioTemporalAttentionLayer(embed_size, num_heads, attn_type) for _ in range(num_layers)])


class TokenSelection(nn.Module):
    def __init__(self, embed_size, top_k):
        super().__init__()
        self.score_net = nn.Linear(embed_size, 1)
        self.top_k = top_k
        self.score_net.bias.data.zero_()


class DifferentiableTokenSelection(nn.Module):
    def __init__(self, embed_size, top_k, tau=1.0):
        super().__init__()
        self.score_net = nn.Linear(embed_size, top_k)
        self.top_k = top_k
        self.tau = tau


class DynamicMultiScalePooling(nn.Module):
    def __init__(self, embed_size, scales=(1, 2, 4)):
        super().__init__()
        if list(scales) != [1, 2, 4]:
            raise ValueError("the HIP pooling kernel implements the reference's scales [1, 2, 4] (svr.py:120,177)")
        self.scales = list(scales)
        self.gate_fc = nn.Linear(embed_size, 1)


class SpatioTemporalVisualTokenRefinerModel(nn.Module):
    def __init__(self, embed_size, num_heads, num_layers, top_k, use_multi_scale, attn_type="rma",
                 enable_diffts=False, enable_dmtp=False):
        super().__init__()
        self.attention_network = SpatioTemporalSignificanceScoring(embed_size, num_heads, num_layers, attn_type)
        if enable_diffts:
            self.token_selection = DifferentiableTokenSelection(embed_size, top_k)
        else:
            self.token_selection = TokenSelection(embed_size, top_k)
        if enable_dmtp:
            self.dynamic_pool = DynamicMultiScalePooling(embed_size)
        self.enable_dmtp = enable_dmtp
        self.use_multi_scale = use_multi_scale


class TextConditionTokenAttMap(nn.Module):
    def __init__(self, d_model, num_heads, attn_type="rma"):
        super().__init__()
        self.visual_cross_attention = MultiHeadCrossAttention(d_model, num_heads)
        self.text_cross_attention = MultiHeadCrossAttention(d_model, num_heads)
        self.dropout_cross = nn.Identity()
        self.norm_cross_v = nn.LayerNorm(d_model)
        self.norm_cross_t = nn.LayerNorm(d_model)
        self.self_attention = _self_attention(d_model, num_heads, attn_type)
        self.dropout_self = nn.Identity()
        self.norm_self = nn.LayerNorm(d_model)


class LinearAggregation(nn.Module):
    def __init__(self, d_model, num_heads):
        super().__init__()
        self.linear_aggregator = MultiHeadCrossAttention(d_model, num_heads)


class TextConditionTokenAggregatorModel(nn.Module):
    def __init__(self, d_model, num_layers, num_heads, attn_type="rma"):
        super().__init__()
        self.layers_vt = nn.ModuleList([TextConditionTokenAttMap(d_model, num_heads, attn_type)
                                        for _ in range(num_layers)])
        self.layer_linagg = LinearAggregation(d_model, num_heads)


def _pack_qkv(m) -> None:
    if isinstance(m, nn.MultiheadAttention):
        return  # in_proj_weight is already q | k | v
    _pack_qkv_linear(m)


def _pack_qkv_linear(m) -> None:
    """Weight packing: lay wq | wk | wv (and biases) of one attention module back to back in HBM so
    the library can run the three projections as ONE GEMM (pipeline.hip: qkv_packed / kv_packed).  The
    nn.Parameters keep their names and shapes -- only their storage becomes a view of the packed buffer, so
    state_dict() / load_state_dict() are unaffected."""
    ws, bs = [m.wq.weight, m.wk.weight, m.wv.weight], [m.wq.bias, m.wk.bias, m.wv.bias]
    E = ws[0].shape[0]
    step_w, step_b = ws[0].numel() * ws[0].element_size(), bs[0].numel() * bs[0].element_size()
    if all(w.is_contiguous() for w in ws) and all(ws[i + 1].data_ptr() == ws[i].data_ptr() + step_w for i in (0, 1)) \
            and all(bs[i + 1].data_ptr() == bs[i].data_ptr() + step_b for i in (0, 1)):
        return
    W = torch.cat([w.data for w in ws], 0).contiguous()  # noqa: N806
    Bv = torch.cat([b.data for b in bs], 0).contiguous()
    for i in range(3):
        ws[i].data = W[i * E:(i + 1) * E]
        bs[i].data = Bv[i * E:(i + 1) * E]


def _att_ptrs(m, value_side: bool = True):
    """value_side = False: the value / output projections are not handed to the library (the un-projected aggregator)."""
    if not value_side:
        return [m.wq.weight, m.wq.bias, m.wk.weight, m.wk.bias, None, None, None, None, None]
    if isinstance(m, nn.MultiheadAttention):  # packed q | k | v rows of in_proj_weight; no relative bias
        E = m.embed_dim
        W, b = m.in_proj_weight, m.in_proj_bias
        return [W[:E], b[:E], W[E:2 * E], b[E:2 * E], W[2 * E:], b[2 * E:], m.out_proj.weight, m.out_proj.bias, None]
    return [m.wq.weight, m.wq.bias, m.wk.weight, m.wk.bias, m.wv.weight, m.wv.bias, m.dense.weight, m.dense.bias,
            getattr(m, "relative_bias", None)]


def _unshare_packed(module, state_dict, prefix, local_metadata):
    """state_dict hook: parameters that pack_weights() turned into views of one q | k | v buffer are handed out as
    unshared copies, so that safetensors / save_pretrained (which refuse tensors sharing storage) and any consumer that
    expects independent tensors see what the reference's checkpoint holds.  Unpacked (CPU) modules are untouched."""
    for k, v in list(state_dict.items()):
        if k.startswith(prefix) and torch.is_tensor(v) and \
                v.untyped_storage().nbytes() != v.numel() * v.element_size():  # (also the LAST view of a packed buffer)
            state_dict[k] = v.detach().clone() if not v.requires_grad else v.clone().detach()
    return state_dict


class u2Tokenizer(nn.Module):
    """Same constructor and forward as the reference (u2Tokenizer.py:6-47)."""

    def __init__(self, embed_size, num_heads, num_layers, top_k, use_multi_scale, num_3d_query_token, hidden_size,
                 attn_type="rma", enable_diffts=False, enable_dmtp=False):
        super().__init__()
        if embed_size != hidden_size:
            raise ValueError("embed_size must equal hidden_size (builder.py:5,11 pass config.hidden_size for both)")
        self.svt_module = SpatioTemporalVisualTokenRefinerModel(
            embed_size=embed_size, num_heads=num_heads, num_layers=num_layers, top_k=top_k,
            use_multi_scale=use_multi_scale, attn_type=attn_type, enable_diffts=enable_diffts,
            enable_dmtp=enable_dmtp)
        self.tta_module = TextConditionTokenAggregatorModel(d_model=embed_size, num_layers=num_layers,
                                                            num_heads=num_heads, attn_type=attn_type)
        self.query_tokens = nn.Parameter(torch.zeros(1, num_3d_query_token, hidden_size))
        self.query_tokens.data.normal_(mean=0.0, std=0.02)
        self.embed_size, self.num_heads, self.num_layers, self.top_k = embed_size, num_heads, num_layers, top_k
        self.use_multi_scale, self.num_query, self.attn_type = bool(use_multi_scale), num_3d_query_token, attn_type
        self.enable_diffts, self.enable_dmtp = bool(enable_diffts), bool(enable_dmtp)
        self._ws = ops._Workspace()
        self._packed_key = None
        self._dead_offloaded = False
        self._register_state_dict_hook(_unshare_packed)
        self.last_topk_indices = None  # (B, top_k) int64 -- set by forward() when hard top-k selection is on
        self.capture_svr_tokens = False  # True: forward() also keeps the refined tokens in last_svr_tokens
        self.last_svr_tokens = None

    def _weights(self):
        w = [self.query_tokens]
        for layer in self.svt_module.attention_network.layers:
            w += _att_ptrs(layer.spatial_attention) + _att_ptrs(layer.temporal_attention)
        sn = self.svt_module.token_selection.score_net
        w += [sn.weight, sn.bias]
        if self.enable_dmtp:
            w += [self.svt_module.dynamic_pool.gate_fc.weight, self.svt_module.dynamic_pool.gate_fc.bias]
        else:
            w += [None, None]
        for layer in self.tta_module.layers_vt:
            w += _att_ptrs(layer.self_attention) + _att_ptrs(layer.visual_cross_attention) \
                + _att_ptrs(layer.text_cross_attention)
            w += [layer.norm_self.weight, layer.norm_self.bias, layer.norm_cross_v.weight, layer.norm_cross_v.bias,
                  layer.norm_cross_t.weight, layer.norm_cross_t.bias]
        # LinearAggregation runs its cross attention with is_compress=True (tta.py:109-116): wv / dense are never read
        w += _att_ptrs(self.tta_module.layer_linagg.linear_aggregator, value_side=False)
        return w

    def pack_weights(self) -> None:
        """Idempotent.  Runs eagerly whenever the parameters move (.to() / .cuda() / .bfloat16() go through _apply) and
        as a safety net at the head of forward().  Packing re-points wq / wk / wv at freshly written buffers, so it ends
        with a device-wide synchronisation: every stream sees the packed weights, and the old storages are idle when
        the caching allocator takes them back (forward calls may be issued on several non-blocking streams)."""
        key = (self.query_tokens.data_ptr(), self.query_tokens.dtype)
        if self._packed_key == key:
            return
        with torch.no_grad():
            for layer in self.svt_module.attention_network.layers:
                _pack_qkv(layer.spatial_attention)
                _pack_qkv(layer.temporal_attention)
            for layer in self.tta_module.layers_vt:
                for m in (layer.self_attention, layer.visual_cross_attention, layer.text_cross_attention):
                    _pack_qkv(m)
        if self.query_tokens.is_cuda:
            torch.cuda.synchronize(self.query_tokens.device)
        self._packed_key = key

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._packed_key = None
        if self.query_tokens.is_cuda:
            self.pack_weights()
            if self._dead_offloaded:
                self.offload_dead_parameters()
        return r

    def dead_parameters(self):
        """linear_aggregator.wv / dense: constructed, initialised and checkpointed by the reference, never read by its
        forward (MultiHeadCrossAttention with is_compress=True skips both: tta.py:47-48,62-65,109-116)."""
        la = self.tta_module.layer_linagg.linear_aggregator
        return [la.wv.weight, la.wv.bias, la.dense.weight, la.dense.bias]

    def offload_dead_parameters(self) -> int:
        """Park the never-read aggregator projections in host memory (2 E^2 + 2 E elements -- 67 MB of HBM at
        E = 4096).  They stay nn.Parameters under their reference names, so state_dict() / save_checkpoint() /
        load_state_dict() are unchanged and lossless; they stop requiring gradients (they never receive any), so optimisers
        skip them.  Sticky: a later .to(device) parks them again.  Returns the device bytes released."""
        freed = 0
        for p in self.dead_parameters():
            if p.is_cuda:
                freed += p.numel() * p.element_size()
                p.data = p.data.to("cpu")
            p.requires_grad_(False)
        self._dead_offloaded = True
        return freed

    # The envelope of the HIP path, checked here so that a violation names the limit instead of a bare U2TOK_ERR_ARG.
    def _check_envelope(self, B, T, N, E, Lt):
        H = self.num_heads
        if E % H or (E // H) % 8:
            raise RuntimeError(f"head dim {E}/{H} must be a multiple of 8 (16-byte fragment loads)")
        if max(N, T, self.num_query) > 512:
            raise RuntimeError(f"sequence lengths N={N}, T={T}, queries={self.num_query} must be <= max_seq_len = 512 "
                               "(relative-bias table / RoPE cache of the reference, rma.py:6,64-68, rope.py:19)")
        if not self.enable_diffts:
            if self.top_k > T * N:
                raise RuntimeError(f"top_k={self.top_k} > T*N={T * N}: torch.topk would raise in the reference (svr.py:82)")
            if T * N > 8192:
                raise RuntimeError(f"hard top-k sorts T*N={T * N} scores in one workgroup's LDS: limit 8192")

    def forward_with_taps(self, v_token, t_token, svr_in=None, visual_in=None, tta_in=None):
        """Parity instrument (u2tok_tokenizer_forward_taps): the inference forward with any layer's input replaced
        ("teacher forcing": svr_in / tta_in are lists of num_layers tensors or None entries, visual_in replaces the tokens
        the aggregation stage attends to) and every layer's output copied out.  Returns (out, taps) with taps =
        {"svr_out": [...], "visual_out": tensor, "tta_out": [...]}."""
        with ops.on_device(v_token, elem=self.query_tokens.dtype):
            return self._forward_with_taps(v_token, t_token, svr_in, visual_in, tta_in)

    def _forward_with_taps(self, v_token, t_token, svr_in, visual_in, tta_in):
        h = _lib.load_library()
        self.pack_weights()
        v_token = ops._need(v_token, ops.ELEM, "v_token").contiguous()
        t_token = ops._need(t_token, ops.ELEM, "t_token").contiguous()
        (B, T, N, E) = v_token.size()
        self._check_envelope(B, T, N, E, t_token.shape[1])
        L, Q, dev = self.num_layers, self.num_query, v_token.device
        Lv = self.top_k + (self.top_k // 2 + self.top_k // 4 if self.use_multi_scale else 0)
        cfg = self._config(B, T, N, E, t_token.shape[1])
        table = ops.weight_table(self._weights())
        nbytes = h.u2tok_tokenizer_workspace_bytes(C.byref(cfg))
        keep = []

        def ptr_array(tensors, shape):
            arr = (C.c_void_p * L)()
            for i in range(L):
                t = tensors[i] if tensors is not None and i < len(tensors) else None
                if t is not None:
                    t = ops._need(t, ops.ELEM, "tap").contiguous()
                    assert tuple(t.shape) == shape, (tuple(t.shape), shape)
                    keep.append(t)
                    arr[i] = t.data_ptr()
            return arr

        svr_out = [torch.empty((B, T, N, E), dtype=ops.elem_dtype(), device=dev) for _ in range(L)]
        tta_out = [torch.empty((B, Q, E), dtype=ops.elem_dtype(), device=dev) for _ in range(L)]
        visual_out = torch.empty((B, Lv, E), dtype=ops.elem_dtype(), device=dev)
        if visual_in is not None:
            visual_in = ops._need(visual_in, ops.ELEM, "visual_in").contiguous()
            assert tuple(visual_in.shape) == (B, Lv, E)
        taps = _lib.TokTaps(svr_in=ptr_array(svr_in, (B, T, N, E)), svr_out=ptr_array(svr_out, (B, T, N, E)),
                            visual_in=None if visual_in is None else visual_in.data_ptr(), visual_out=visual_out.data_ptr(),
                            tta_in=ptr_array(tta_in, (B, Q, E)), tta_out=ptr_array(tta_out, (B, Q, E)))
        with ops.on_device(v_token) as (h, stream):
            ws = self._ws.get(nbytes, dev)
            out = torch.empty((B, Q, E), dtype=ops.elem_dtype(), device=dev)
            idx = None if self.enable_diffts else torch.empty((B, self.top_k), dtype=torch.int64, device=dev)
            _lib.check(h.u2tok_tokenizer_forward_taps(C.byref(cfg), table, v_token.data_ptr(), t_token.data_ptr(),
                                                      out.data_ptr(), None if idx is None else idx.data_ptr(),
                                                      C.byref(taps), ws.data_ptr(), ws.numel(), stream),
                       "u2tok_tokenizer_forward_taps")
        self.last_topk_indices = idx
        return out, {"svr_out": svr_out, "visual_out": visual_out, "tta_out": tta_out}

    def _config(self, B, T, N, E, Lt):
        return _lib.TokConfig(B=B, T=T, N=N, E=E, Lt=Lt, num_heads=self.num_heads,
                              num_layers=self.num_layers, top_k=self.top_k, num_query=self.num_query,
                              use_multi_scale=int(self.use_multi_scale), attn_type=_ATTN_TYPES.get(self.attn_type, 2),
                              enable_diffts=int(self.enable_diffts), enable_dmtp=int(self.enable_dmtp),
                              max_seq_len=512, diffts_tau=float(getattr(self.svt_module.token_selection, "tau", 1.0)),
                              ln_eps=1e-5)

    def forward(self, v_token, t_token):
        # parameters in bf16, or in fp16 for a model loaded in float16 (evalscipt/ourmodel_amos.py:33): the f16 build of the library
        with ops.on_device(v_token, elem=self.query_tokens.dtype):
            return self._forward(v_token, t_token)

    def _forward(self, v_token, t_token):
        if torch.is_grad_enabled() and (v_token.requires_grad or t_token.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            # training: the same kernels sequenced op by op behind torch.autograd.Function (autograd.py)
            from . import autograd as AG
            ops.training_needs_bf16(self.query_tokens.dtype, "u2Tokenizer")
            ops._need(v_token, ops.ELEM, "v_token"), ops._need(t_token, ops.ELEM, "t_token")
            (B, T, N, E) = v_token.size()
            self._check_envelope(B, T, N, E, t_token.shape[1])
            with ops.on_device(v_token):
                return AG.tokenizer_forward(self, v_token, t_token)
        h = _lib.load_library()
        if self.query_tokens.is_cuda:
            self.pack_weights()
        v_token = ops._need(v_token, ops.ELEM, "v_token").contiguous()
        t_token = ops._need(t_token, ops.ELEM, "t_token").contiguous()
        (B, T, N, E) = v_token.size()
        if E != self.embed_size or t_token.shape[0] != B or t_token.shape[2] != E:
            raise RuntimeError(f"shape mismatch: v_token {tuple(v_token.shape)}, t_token {tuple(t_token.shape)}")
        self._check_envelope(B, T, N, E, t_token.shape[1])
        cfg = self._config(B, T, N, E, t_token.shape[1])
        table = ops.weight_table(self._weights())
        nbytes = h.u2tok_tokenizer_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("u2tok_tokenizer_workspace_bytes rejected the configuration "
                               f"(B={B}, T={T}, N={N}, E={E}, heads={self.num_heads}, top_k={self.top_k})")
        idx = svr = None
        with ops.on_device(v_token) as (h, stream):
            ws = self._ws.get(nbytes, v_token.device)
            out = torch.empty((B, self.num_query, E), dtype=ops.elem_dtype(), device=v_token.device)
            if not self.enable_diffts:
                idx = torch.empty((B, self.top_k), dtype=torch.int64, device=v_token.device)
            if self.capture_svr_tokens:
                svr = torch.empty((B, T * N, E), dtype=ops.elem_dtype(), device=v_token.device)
            _lib.check(h.u2tok_tokenizer_forward(C.byref(cfg), table, v_token.data_ptr(), t_token.data_ptr(),
                                                 out.data_ptr(), None if idx is None else idx.data_ptr(),
                                                 None if svr is None else svr.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 stream), "u2tok_tokenizer_forward")
        self.last_topk_indices = idx
        self.last_svr_tokens = svr
        return out
