"""Training path of the drop-in modules: torch.autograd.Functions over the C-ABI building blocks.

Inference (torch.no_grad) runs each module as ONE fused launch sequence inside libu2tok_hip.so (pipeline.hip).  Under
autograd the same kernels are sequenced from here, op by op, so that every op can save what its backward needs:

  * GEMM-shaped work -- forward AND backward -- runs on the library's MFMA kernels: y = x W^T (+ b, + residual) through
    u2tok_gemm_bf16; dX = dY W and dW = dY^T X through the same kernel's K-major operand forms (LDS transpose reads: no
    transposed copy of W, dY or X in HBM); the tokenizer attention cores' dP = dO V^T, dV = P^T dO, dQ = dS K,
    dK = dS^T Q as NT products on operands transposed by u2tok_transpose_bf16, batched over (batch, head);
  * the ViT attention is flash both ways: the forward kernel keeps no S x S tensor, the backward is the fused kernel pair
    of csrc/attn_bwd.hip, which rebuilds the probabilities tile by tile from q, k and the saved output (the unfused chain
    -- scores GEMM + row softmax + batched products -- serves the tokenizer's d = E/8 cores and as the cross-check);
  * the non-GEMM pieces are the kernels of csrc/backward.hip (GELU, LayerNorm, softmax, relative-bias table, column
    sums);
  * torch itself only moves data (views, permutes, cat / split, residual adds, the scatter of the hard top-k gather)
    and differentiates those moves.

Reference being differentiated: src/model/multimodal_encoder/vit.py (MONAI blocks), multimodal_projector/
spatial_pooling_projector.py, u2tokenizer/{svr,tta,rma,rope}.py.  The gradient tests differentiate the CPU restatement
of those files with torch.autograd on the host and compare every parameter's gradient (tests/test_gpu_backward.py).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.autograd import Function

from . import ops

BF = torch.bfloat16


def _r8(n: int) -> int:
    return (n + 7) // 8 * 8


_scratch = {}


def ensure_gemm_scratch(device: torch.device) -> None:
    """Split-K scratch for the skinny products of the training path (dW of small layers has K = rows), one buffer per
    (device, stream), registered on the active context."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, id(ops.active_context(device)))
    if key not in _scratch:
        buf = torch.empty(64 << 20, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            ops.set_gemm_scratch(buf)
        _scratch[key] = buf


def _bind_context(cls):
    """Class decorator: remember the execution context the forward ran on (ops.active_context: the innermost
    `with ops.Context()` of the calling thread, else the device's default) and re-enter it in backward.  Function.backward
    runs on the autograd engine's device thread, whose thread-local context stack is empty: without this the backward
    kernels of a model used inside `with ops.Context():` would take the default context's options, miss the split-K scratch
    registered on the private context and land their profiling records elsewhere."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kwargs):
        t = next((a for a in args if torch.is_tensor(a) and a.is_cuda), None)
        ctx.u2ctx = ops.active_context(t.device) if t is not None else None
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        if ctx.u2ctx is None:
            return bwd(ctx, *grads)
        with ctx.u2ctx:
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


# ------------------------------------------------------------------------------------------------ Linear (+GELU, +residual)
@_bind_context
class LinearFn(Function):
    """y = x W^T (+ b) (-> GELU) (+ residual), bias / residual fused in the GEMM epilogue as in the inference path."""

    @staticmethod
    def forward(ctx, x, w, b, res, gelu: bool):
        K, N = x.shape[-1], w.shape[0]
        if N % 8 or K % 8:
            raise RuntimeError(f"LinearFn: in / out features must be multiples of 8 (got {K}, {N})")
        x2 = x.reshape(-1, K).contiguous()
        z = None
        if gelu:
            if res is not None:
                raise RuntimeError("LinearFn: GELU with a residual is not used by the path")
            z = ops.gemm(x2, w, bias=b)
            y = ops.gelu_fwd(z)
        else:
            y = ops.gemm(x2, w, bias=b, residual=None if res is None else res.reshape(-1, N))
        ctx.save_for_backward(x2, w, z)
        ctx.has_b, ctx.has_res, ctx.xshape = b is not None, res is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, z = ctx.saved_tensors
        N, K = w.shape
        dy2 = dy.reshape(-1, N).contiguous()
        dz = ops.gelu_bwd(z, dy2) if z is not None else dy2
        dx = dw = db = dres = None
        # K-major operand forms of the GEMM kernel (LDS transpose reads): no transposed copies of W, dY or X in HBM.
        # Exception: many rows against a small weight (the ViT: 16392 x {768, 2304, 3072}) -- transposing W costs ~5 us and
        # lets the product take the 256-wide-tile kernel (fc2's dX: 86 us against 133 us in the K-major form).
        if ctx.needs_input_grad[0]:
            if dz.shape[0] >= 4096 and w.numel() <= (1 << 23):
                dx = ops.gemm(dz, ops.transpose_ex(w, 1, N, K, K, 0)[0]).view(ctx.xshape)   # (M, N_out) (K_in, N_out)^T
            else:
                dx = ops.gemm_kmajor(dz, w, a_kmajor=False).view(ctx.xshape)                 # (M, N_out) @ (N_out, K_in)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_kmajor(dz, x2, a_kmajor=True)                      # (M, N_out)^T @ (M, K_in)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = ops.colsum(dz)
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None


def linear(x, w, b=None, res=None, gelu=False):
    return LinearFn.apply(x, w, b, res, gelu)


# ------------------------------------------------------------------------------------------------ LayerNorm (+residual)
@_bind_context
class LayerNormFn(Function):
    """y = LayerNorm(x (+ res)) * w + b  (tta.py:96,100,103; MONAI TransformerBlock norm1 / norm2; ViT.norm)."""

    @staticmethod
    def forward(ctx, x, res, w, b, eps: float):
        y = ops.layernorm(x, w, b, residual=res, eps=eps)
        ctx.save_for_backward(x, res, w)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, w = ctx.saved_tensors
        dv, dw, db = ops.layernorm_bwd(x, res, w, dy, ctx.eps)
        return dv, (dv if res is not None else None), dw.to(w.dtype), db.to(w.dtype), None


def layernorm(x, w, b, res=None, eps=1e-5):
    return LayerNormFn.apply(x, res, w, b, eps)


# ------------------------------------------------------------------------------------------------ attention cores
def _rowstride(t):  # (nb, S, E) view with unit feature stride -> (batch stride, row stride)
    assert t.stride(2) == 1, "attention operands must be feature-contiguous"
    return t.stride(0), t.stride(1)


def _attn_probs(q, k, H, scale, rel_bias, max_len):
    """softmax(q k^T * scale + bias) per (batch, head): q (nb, Sq, E), k (nb, Skv, E) views -> P (nb*H, Sq, ldp) bf16."""
    nb, Sq, E = q.shape
    Skv, d = k.shape[1], E // H
    qb, ql = _rowstride(q)
    kb, kl = _rowstride(k)
    S = torch.empty((nb * H, Sq, Skv), dtype=torch.float32, device=q.device)
    ops.gemm_strided(q, k, S, M=Sq, N=Skv, K=d, lda=ql, ldb=kl, ldc=Skv, nz=nb * H, nbh=H, sAb=qb, sAh=d, sBb=kb, sBh=d,
                     sCb=H * Sq * Skv, sCh=Sq * Skv, out_f32=True)
    return ops.softmax_rows(S, scale=scale, rel_bias=rel_bias, heads=H, max_len=max_len, ldp=_r8(Skv))


def _kmajor_ok(Skv: int, *views) -> bool:
    """The K-major operand forms of the GEMM need the key count (their contiguous dimension) and every stride in whole
    16-byte chunks; other shapes go through transposed copies."""
    return Skv % 8 == 0 and all(s % 8 == 0 for t in views for s in _rowstride(t))


def _attn_pv(P, v, H, Sq):
    """out (nb, Sq, E) = P V per (batch, head); v (nb, Skv, E) view, read in place as a K-major operand."""
    nb, Skv, E = v.shape
    d, ldp = E // H, P.shape[-1]
    vb, vl = _rowstride(v)
    out = torch.empty((nb, Sq, E), dtype=BF, device=v.device)
    if _kmajor_ok(Skv, v):
        ops.gemm_strided(P, v, out, M=Sq, N=d, K=Skv, lda=ldp, ldb=vl, ldc=E, nz=nb * H, nbh=H, sAb=H * Sq * ldp,
                         sAh=Sq * ldp, sBb=vb, sBh=d, sCb=Sq * E, sCh=d, b_kmajor=True)
        return out
    Vt = ops.transpose_ex(v, nb, Skv, E, vl, vb, ld_out=ldp)                      # (nb, E, ldp)
    ops.gemm_strided(P, Vt, out, M=Sq, N=d, K=ldp, lda=ldp, ldb=ldp, ldc=E, nz=nb * H, nbh=H, sAb=H * Sq * ldp,
                     sAh=Sq * ldp, sBb=E * ldp, sBh=d * ldp, sCb=Sq * E, sCh=d)
    return out


def _attn_backward(q, k, v, P, dO, H, scale, dq, dk, dv, dtable, max_len):
    """Gradients of out = softmax(q k^T scale + bias) v written into the (nb, S, E) views dq / dk / dv (any may be None);
    dtable: fp32 relative-bias gradient table to accumulate into, or None.  The products that contract over queries
    (dV = P^T dO, dK = dS^T Q) and over keys against a row-major operand (dQ = dS K) read P / dS / dO / Q / K in place through
    the K-major forms of the GEMM; key counts that are no multiple of 8 go through transposed copies."""
    nb, Sq, E = q.shape
    Skv, d, Z = k.shape[1], E // H, nb * H
    ldp, Sqp = P.shape[-1], _r8(Sq)
    dO = dO.contiguous()
    vb, vl = _rowstride(v)
    dP = torch.empty((Z, Sq, Skv), dtype=torch.float32, device=q.device)
    ops.gemm_strided(dO, v, dP, M=Sq, N=Skv, K=d, lda=E, ldb=vl, ldc=Skv, nz=Z, nbh=H, sAb=Sq * E, sAh=d, sBb=vb, sBh=d,
                     sCb=H * Sq * Skv, sCh=Sq * Skv, out_f32=True)
    dS = ops.softmax_bwd(P, dP, Skv)                                              # (Z, Sq, ldp), pad columns zero
    del dP
    if dtable is not None:
        ops.relbias_grad(dS, dtable, Sq, H, max_len)
    km = _kmajor_ok(Skv, q, k)
    pz = dict(lda=ldp, nz=Z, nbh=H, sAb=H * Sq * ldp, sAh=Sq * ldp)             # P / dS as the A operand
    if dv is not None:
        b_, l_ = _rowstride(dv)
        if km:   # dV (Skv, d) = P^T (Skv, Sq) dO (Sq, d): both operands K-major, K = Sq
            ops.gemm_strided(P, dO, dv, M=Skv, N=d, K=Sq, ldb=E, ldc=l_, sBb=Sq * E, sBh=d, sCb=b_, sCh=d,
                             a_kmajor=True, b_kmajor=True, **pz)
        else:
            Pt = ops.transpose_ex(P, Z, Sq, Skv, ldp, Sq * ldp, ld_out=Sqp)       # (Z, Skv, Sqp)
            dOt = ops.transpose_ex(dO, nb, Sq, E, E, Sq * E, ld_out=Sqp)          # (nb, E, Sqp)
            ops.gemm_strided(Pt, dOt, dv, M=Skv, N=d, K=Sqp, lda=Sqp, ldb=Sqp, ldc=l_, nz=Z, nbh=H, sAb=H * Skv * Sqp,
                             sAh=Skv * Sqp, sBb=E * Sqp, sBh=d * Sqp, sCb=b_, sCh=d)
            del Pt, dOt
    if dq is not None:
        kb, kl = _rowstride(k)
        b_, l_ = _rowstride(dq)
        if km:   # dQ (Sq, d) = dS (Sq, Skv) K (Skv, d): B K-major
            ops.gemm_strided(dS, k, dq, M=Sq, N=d, K=Skv, ldb=kl, ldc=l_, sBb=kb, sBh=d, sCb=b_, sCh=d, alpha=scale,
                             b_kmajor=True, **pz)
        else:
            Kt = ops.transpose_ex(k, nb, Skv, E, kl, kb, ld_out=ldp)              # (nb, E, ldp)
            ops.gemm_strided(dS, Kt, dq, M=Sq, N=d, K=ldp, ldb=ldp, ldc=l_, sBb=E * ldp, sBh=d * ldp, sCb=b_, sCh=d,
                             alpha=scale, **pz)
    if dk is not None:
        qb, ql = _rowstride(q)
        b_, l_ = _rowstride(dk)
        if km:   # dK (Skv, d) = dS^T (Skv, Sq) Q (Sq, d): both operands K-major, K = Sq
            ops.gemm_strided(dS, q, dk, M=Skv, N=d, K=Sq, ldb=ql, ldc=l_, sBb=qb, sBh=d, sCb=b_, sCh=d, alpha=scale,
                             a_kmajor=True, b_kmajor=True, **pz)
        else:
            dSt = ops.transpose_ex(dS, Z, Sq, Skv, ldp, Sq * ldp, ld_out=Sqp)     # (Z, Skv, Sqp)
            Qt = ops.transpose_ex(q, nb, Sq, E, ql, qb, ld_out=Sqp)               # (nb, E, Sqp)
            ops.gemm_strided(dSt, Qt, dk, M=Skv, N=d, K=Sqp, lda=Sqp, ldb=Sqp, ldc=l_, nz=Z, nbh=H, sAb=H * Skv * Sqp,
                             sAh=Skv * Sqp, sBb=E * Sqp, sBh=d * Sqp, sCb=b_, sCh=d, alpha=scale)


@_bind_context
class SelfAttnFn(Function):
    """Self-attention core on a packed q | k | v buffer (nb, S, 3E) -> (nb, S, E).  rel_bias: (2 max_len - 1, H) table of
    RelativeMultiheadAttention (rma.py:64-70) or None.  flash (head dim 64, no bias): the forward is the ViT flash
    kernel with the LAST row of every batch as its "extra" row; flash = 2 (True): the backward is the fused kernel pair of
    attn_bwd.hip (probabilities rebuilt tile by tile from q, k and the saved output, which the out-projection keeps alive
    anyway); flash = 1: the backward rebuilds the probabilities in HBM and runs the unfused chain (kept as the cross-check
    of the fused kernels, tests/test_gpu_backward.py)."""

    @staticmethod
    def forward(ctx, qkv, rel_bias, H: int, scale: float, max_len: int, flash):
        nb, S, E3 = qkv.shape
        E = E3 // 3
        qkv = qkv.contiguous()
        flash = 2 if flash is True else int(flash)
        P = out_saved = lse = None
        if flash == 2:
            out, lse = ops.flash_attention_d64(qkv, H, scale, extra_last=S > 1, return_lse=True)
            out_saved = out
        elif flash:
            out = ops.flash_attention_d64(qkv, H, scale, extra_last=S > 1)
        else:
            P = _attn_probs(qkv[..., :E], qkv[..., E:2 * E], H, scale, rel_bias, max_len)
            out = _attn_pv(P, qkv[..., 2 * E:], H, S)
        ctx.save_for_backward(qkv, rel_bias, P, out_saved, lse)
        ctx.cfg = (H, scale, max_len)
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, rel_bias, P, out, lse = ctx.saved_tensors
        H, scale, max_len = ctx.cfg
        E = qkv.shape[-1] // 3
        if out is not None:
            return ops.flash_attention_d64_bwd(qkv, out, dO, H, scale, lse=lse), None, None, None, None, None
        q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
        if P is None:
            P = _attn_probs(q, k, H, scale, rel_bias, max_len)
        dqkv = torch.empty_like(qkv)
        dtable = None
        if rel_bias is not None and ctx.needs_input_grad[1]:
            dtable = torch.zeros(rel_bias.shape, dtype=torch.float32, device=qkv.device)
        _attn_backward(q, k, v, P, dO, H, scale, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:], dtable, max_len)
        return dqkv, (None if dtable is None else dtable.to(rel_bias.dtype)), None, None, None, None


@_bind_context
class CrossAttnFn(Function):
    """MultiHeadCrossAttention core (tta.py:55-61): q (nb, Sq, E), packed k | v (nb, Skv, 2E) -> (nb, Sq, E)."""

    @staticmethod
    def forward(ctx, q, kv, H: int, scale: float):
        E = q.shape[-1]
        q, kv = q.contiguous(), kv.contiguous()
        P = _attn_probs(q, kv[..., :E], H, scale, None, 0)
        out = _attn_pv(P, kv[..., E:], H, q.shape[1])
        ctx.save_for_backward(q, kv, P)
        ctx.cfg = (H, scale)
        return out

    @staticmethod
    def backward(ctx, dO):
        q, kv, P = ctx.saved_tensors
        H, scale = ctx.cfg
        E = q.shape[-1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        _attn_backward(q, kv[..., :E], kv[..., E:], P, dO, H, scale, dq, dkv[..., :E], dkv[..., E:], None, 0)
        return dq, dkv, None, None


@_bind_context
class AttnFn(Function):
    """Attention core on separate q, k, v (nb, S, E) -- the un-projected aggregation of LinearAggregation (tta.py:109-116)."""

    @staticmethod
    def forward(ctx, q, k, v, H: int, scale: float):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        P = _attn_probs(q, k, H, scale, None, 0)
        out = _attn_pv(P, v, H, q.shape[1])
        ctx.save_for_backward(q, k, v, P)
        ctx.cfg = (H, scale)
        return out

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P = ctx.saved_tensors
        H, scale = ctx.cfg
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _attn_backward(q, k, v, P, dO, H, scale, dq, dk, dv, None, 0)
        return dq, dk, dv, None, None


@_bind_context
class RopeFn(Function):
    """rotate-half RoPE on the q and k thirds of a packed (nb, S, 3E) buffer (rope.py:77-80); position = row index."""

    @staticmethod
    def forward(ctx, qkv, H: int, max_len: int):
        nb, S, E3 = qkv.shape
        E = E3 // 3
        out = qkv.clone()
        ops.rope_apply(out[..., :E], nb, S, 1, H, E // H, max_len)
        ops.rope_apply(out[..., E:2 * E], nb, S, 1, H, E // H, max_len)
        ctx.cfg = (H, max_len)
        return out

    @staticmethod
    def backward(ctx, d):
        H, max_len = ctx.cfg
        nb, S, E3 = d.shape
        E = E3 // 3
        g = d.clone()
        ops.rope_apply(g[..., :E], nb, S, 1, H, E // H, max_len, inverse=True)
        ops.rope_apply(g[..., E:2 * E], nb, S, 1, H, E // H, max_len, inverse=True)
        return g, None, None


# ------------------------------------------------------------------------------------------------ selection / pooling
@_bind_context
class DiffTSFn(Function):
    """DifferentiableTokenSelection (svr.py:101-117): out[r] = sum_tok softmax_tok(score_net(x) / tau)[tok, r] x[tok],
    computed operand-swapped as in the inference path: raw^T = W X^T + b (k x TN), row softmax, P X."""

    @staticmethod
    def forward(ctx, x, w, b, tau: float):
        B, TN, E = x.shape
        k = w.shape[0]
        x = x.contiguous()
        raw = torch.empty((B, k, TN), dtype=torch.float32, device=x.device)
        ops.gemm_strided(w, x, raw, M=k, N=TN, K=E, lda=E, ldb=E, ldc=TN, nz=B, sBb=TN * E, sCb=k * TN, out_f32=True,
                         bias=b, bias_m=True)
        P = ops.softmax_rows(raw, scale=1.0 / tau, ldp=_r8(TN))                   # (B, k, ldp)
        ldp = P.shape[-1]
        Xt = ops.transpose_ex(x, B, TN, E, E, TN * E, ld_out=ldp)                 # (B, E, ldp)
        out = torch.empty((B, k, E), dtype=BF, device=x.device)
        ops.gemm_strided(P, Xt, out, M=k, N=E, K=ldp, lda=ldp, ldb=ldp, ldc=E, nz=B, sAb=k * ldp, sBb=E * ldp, sCb=k * E)
        ctx.save_for_backward(x, w, P, Xt)
        ctx.tau = tau
        return out

    @staticmethod
    def backward(ctx, dsel):
        x, w, P, Xt = ctx.saved_tensors
        B, TN, E = x.shape
        k, ldp = w.shape[0], P.shape[-1]
        dsel = dsel.contiguous()
        dP = torch.empty((B, k, TN), dtype=torch.float32, device=x.device)
        ops.gemm_strided(dsel, x, dP, M=k, N=TN, K=E, lda=E, ldb=E, ldc=TN, nz=B, sAb=k * E, sBb=TN * E, sCb=k * TN,
                         out_f32=True)
        dS = ops.softmax_bwd(P, dP, TN)                                            # d logits; d raw = dS / tau
        inv = 1.0 / ctx.tau
        kp = _r8(k)
        # dX = P^T dsel + (dS^T W) / tau
        Pt = ops.transpose_ex(P, B, k, TN, ldp, k * ldp, ld_out=kp)               # (B, TN, kp)
        dselT = ops.transpose_ex(dsel, B, k, E, E, k * E, ld_out=kp)              # (B, E, kp)
        dX = torch.empty_like(x)
        ops.gemm_strided(Pt, dselT, dX, M=TN, N=E, K=kp, lda=kp, ldb=kp, ldc=E, nz=B, sAb=TN * kp, sBb=E * kp, sCb=TN * E)
        dSt = ops.transpose_ex(dS, B, k, TN, ldp, k * ldp, ld_out=kp)             # (B, TN, kp)
        Wt = ops.transpose_ex(w, 1, k, E, E, 0, ld_out=kp)                        # (1, E, kp)
        dX2 = torch.empty_like(x)
        ops.gemm_strided(dSt, Wt, dX2, M=TN, N=E, K=kp, lda=kp, ldb=kp, ldc=E, nz=B, sAb=TN * kp, sCb=TN * E, alpha=inv)
        dX = dX + dX2
        # dW = sum_b dS_b X_b / tau : (k, E) = [dS_0 | dS_1 | ...] (k, B*ldp) [Xt_0 | Xt_1 | ...]^T
        dS_cat = dS.permute(1, 0, 2).reshape(k, B * ldp).contiguous()
        Xt_cat = Xt.permute(1, 0, 2).reshape(E, B * ldp).contiguous()
        dW = ops.gemm(dS_cat, Xt_cat, alpha=inv)
        db = (dS.float().sum(dim=(0, 2)) * inv).to(w.dtype)
        return dX, dW, db, None


@_bind_context
class MultiScalePoolFn(Function):
    """{1,2,4} average pooling along the token axis (svr.py:176-184), optionally gated by DynamicMultiScalePooling
    (svr.py:126-151).  Forward: the HIP kernel of the inference path; backward: the few thousand-element gate algebra
    and the pooling transposes in torch (fp32)."""

    @staticmethod
    def forward(ctx, x, gate_w, gate_b):
        out = ops.multiscale_pool(x, gate_w, gate_b)
        ctx.save_for_backward(x, gate_w, gate_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gw, gb = ctx.saved_tensors
        B, k, E = x.shape
        xf, d = x.float(), dout.float()
        sizes = [k, k // 2, k // 4]
        pools = [xf, xf[:, :2 * sizes[1]].view(B, sizes[1], 2, E).mean(2), xf[:, :4 * sizes[2]].view(B, sizes[2], 4, E).mean(2)]
        douts = list(d.split(sizes, dim=1))
        dpools, dgw, dgb = douts, None, None
        if gw is not None:
            means = torch.stack([p.mean(1) for p in pools], 1)                     # (B, 3, E)
            logits = means @ gw.float().reshape(-1) + gb.float().reshape(())       # (B, 3)
            g = torch.softmax(logits, dim=1)
            dg = torch.stack([(douts[i] * pools[i]).sum((1, 2)) for i in range(3)], 1)   # (B, 3)
            dlog = g * (dg - (g * dg).sum(1, keepdim=True))
            dgw = torch.einsum("bs,bse->e", dlog, means).reshape(gw.shape).to(gw.dtype)
            dgb = dlog.sum().reshape(gb.shape).to(gb.dtype)
            dmeans = dlog.unsqueeze(-1) * gw.float().reshape(1, 1, E)              # (B, 3, E)
            dpools = [g[:, i, None, None] * douts[i] + dmeans[:, i, None, :] / sizes[i] for i in range(3)]
        dx = dpools[0].clone()
        dx[:, :2 * sizes[1]] += dpools[1].repeat_interleave(2, dim=1) / 2
        dx[:, :4 * sizes[2]] += dpools[2].repeat_interleave(4, dim=1) / 4
        return dx.to(x.dtype), dgw, dgb


@_bind_context
class HardTopKFn(Function):
    """TokenSelection (svr.py:75-91): indices are not differentiable (score_net receives no gradient, exactly as in the
    reference -- hence its find_unused_parameters=True, train_stage1.py:21-22); the gather scatters its gradient back."""

    @staticmethod
    def forward(ctx, x, w, b, k: int):
        B, TN, E = x.shape
        scores = ops.score_gemv(x.contiguous(), w, b)
        idx = ops.topk_sorted(scores.view(B, TN), k)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        ctx.mark_non_differentiable(idx)
        return ops.gather_rows(x.contiguous(), idx), idx

    @staticmethod
    def backward(ctx, dsel, _didx):
        (idx,) = ctx.saved_tensors
        B, TN, E = ctx.shape
        dx = torch.zeros((B, TN, E), dtype=dsel.dtype, device=dsel.device)
        dx.scatter_(1, idx.unsqueeze(-1).expand(-1, -1, E), dsel.contiguous())     # top-k indices are distinct
        return dx, None, None, None


@_bind_context
class AvgPool3dFn(Function):
    """SpatialPoolingProjector pooling (spatial_pooling_projector.py:38-41) over the (g1, g2, g3) token grid."""

    @staticmethod
    def forward(ctx, x, grid, window):
        ctx.grid, ctx.window = tuple(grid), tuple(window)
        return ops.avgpool3d_tokens(x, grid, window)

    @staticmethod
    def backward(ctx, dy):
        (g1, g2, g3), (w1, w2, w3) = ctx.grid, ctx.window
        p1, p2, p3 = g1 // w1, g2 // w2, g3 // w3
        nb, _, C = dy.shape
        d = (dy / (w1 * w2 * w3)).view(nb, p1, 1, p2, 1, p3, 1, C).expand(nb, p1, w1, p2, w2, p3, w3, C)
        dx = torch.zeros((nb, g1, g2, g3, C), dtype=dy.dtype, device=dy.device)
        dx[:, :p1 * w1, :p2 * w2, :p3 * w3] = d.reshape(nb, p1 * w1, p2 * w2, p3 * w3, C)
        return dx.view(nb, g1 * g2 * g3, C), None, None


# ================================================================================================ module forwards
class _AliasCatFn(Function):
    """torch.cat(parts, 0) for parameters that ALREADY lie back to back in one storage (u2Tokenizer.pack_weights() points
    wq | wk | wv at slices of one packed buffer): the forward is a zero-copy view of that storage, the backward hands every
    part its rows of the gradient.  (torch.cat copied 100 MB per attention module and step at E = 4096.)"""

    @staticmethod
    def forward(ctx, *parts):
        ctx.rows = [p.shape[0] for p in parts]
        base = parts[0].detach()
        shape = (sum(ctx.rows),) + tuple(base.shape[1:])
        return base.as_strided(shape, base.stride())

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


def _cat_rows(parts):
    """Rows of `parts` stacked: a view when they are contiguous neighbours in one storage, torch.cat otherwise."""
    p0 = parts[0]
    adjacent = all(p.is_contiguous() and p.dtype == p0.dtype and p.device == p0.device and p.shape[1:] == p0.shape[1:]
                   for p in parts)
    if adjacent:
        end = p0.data_ptr() + p0.numel() * p0.element_size()
        for p in parts[1:]:
            adjacent = adjacent and p.data_ptr() == end
            end = p.data_ptr() + p.numel() * p.element_size()
        if adjacent:
            st = p0.untyped_storage()
            adjacent = end <= st.data_ptr() + st.nbytes() and all(
                p.untyped_storage().data_ptr() == st.data_ptr() for p in parts[1:])
    return _AliasCatFn.apply(*parts) if adjacent else torch.cat(list(parts), 0)


def _qkv_params(m):
    """(W (3E, E), b (3E,)) of an attention module: wq | wk | wv stacked -- autograd routes the gradient back."""
    return _cat_rows((m.wq.weight, m.wk.weight, m.wv.weight)), _cat_rows((m.wq.bias, m.wk.bias, m.wv.bias))


def _self_attention(m, x, attn_type: str, H: int):
    """RelativeMultiheadAttention / RotaryMultiheadAttention forward on x (nb, S, E) (rma.py:46-83, rope.py:62-91)."""
    E = x.shape[-1]
    W, b = _qkv_params(m)
    qkv = linear(x, W, b)
    scale = 1.0 / math.sqrt(E // H)
    if attn_type == "rope":
        qkv = RopeFn.apply(qkv, H, m.max_seq_len)
        ctxv = SelfAttnFn.apply(qkv, None, H, scale, 0, False)
    else:
        ctxv = SelfAttnFn.apply(qkv, m.relative_bias, H, scale, m.max_seq_len, False)
    return linear(ctxv, m.dense.weight, m.dense.bias)


def _mha_attention(m, x, H: int):
    """nn.MultiheadAttention(embed, heads) called as m(x, x, x) with its default batch_first=False -- the "linvt" ablation
    (svr.py:16-18,29,35; tta.py:83-84,94; script/amos_mm_stage1/amos_mm_linvt_stage1.sh:46 `--enable_rpe False`): dim 0
    of x is read as the SEQUENCE and dim 1 as the batch, so the "spatial" call attends across (batch, chunk) pairs, the
    "temporal" call across (batch, token) pairs and the TTA self-attention across batch entries.  The kernels are
    batch-first: swap the two axes around the same packed-projection + attention-core + out-projection sequence."""
    E = x.shape[-1]
    xb = x.permute(1, 0, 2).contiguous()                       # (batch = dim 1, sequence = dim 0, E)
    qkv = linear(xb, m.in_proj_weight, m.in_proj_bias)
    ctxv = SelfAttnFn.apply(qkv, None, H, 1.0 / math.sqrt(E // H), 0, False)
    return linear(ctxv, m.out_proj.weight, m.out_proj.bias).permute(1, 0, 2)


def _self_attention_any(m, x, attn_type: str, H: int):
    if attn_type in ("rma", "rope"):
        return _self_attention(m, x, attn_type, H)
    return _mha_attention(m, x, H).contiguous()


def _cross_attention(m, query, value, H: int):
    """MultiHeadCrossAttention forward (tta.py:42-69), is_compress = False."""
    E = query.shape[-1]
    q = linear(query, m.wq.weight, m.wq.bias)
    kv = linear(value, _cat_rows((m.wk.weight, m.wv.weight)), _cat_rows((m.wk.bias, m.wv.bias)))
    ctxv = CrossAttnFn.apply(q, kv, H, 1.0 / math.sqrt(E // H))
    return linear(ctxv, m.dense.weight, m.dense.bias)


def tokenizer_forward(tok, v_token: torch.Tensor, t_token: torch.Tensor) -> torch.Tensor:
    """u2Tokenizer.forward under autograd (u2Tokenizer.py:40-47 = svr.py:166-188 + tta.py:126-140)."""
    B, T, N, E = v_token.shape
    H = tok.num_heads
    ensure_gemm_scratch(v_token.device)
    x = v_token.to(BF)
    t_token = t_token.to(BF)
    # ---- SVR: x = attn(x), spatial then temporal, no residual / norm (svr.py:23-40)
    for layer in tok.svt_module.attention_network.layers:
        xs = _self_attention_any(layer.spatial_attention, x.reshape(B * T, N, E), tok.attn_type, H)
        xt = xs.view(B, T, N, E).permute(0, 2, 1, 3).reshape(B * N, T, E)
        xt = _self_attention_any(layer.temporal_attention, xt, tok.attn_type, H)
        x = xt.view(B, N, T, E).permute(0, 2, 1, 3).contiguous()
    flat = x.reshape(B, T * N, E)
    sel_m = tok.svt_module.token_selection
    if tok.enable_diffts:
        sel = DiffTSFn.apply(flat, sel_m.score_net.weight, sel_m.score_net.bias, float(sel_m.tau))
    else:
        sel, idx = HardTopKFn.apply(flat, sel_m.score_net.weight, sel_m.score_net.bias, tok.top_k)
        tok.last_topk_indices = idx
    if tok.use_multi_scale:
        if tok.enable_dmtp:
            g = tok.svt_module.dynamic_pool.gate_fc
            V = MultiScalePoolFn.apply(sel, g.weight, g.bias)
        else:
            V = MultiScalePoolFn.apply(sel, None, None)
    else:
        V = sel
    # ---- TTA (tta.py:93-107,126-140)
    q = tok.query_tokens.expand(B, -1, -1)
    for layer in tok.tta_module.layers_vt:
        so = _self_attention_any(layer.self_attention, q, tok.attn_type, H)
        q1 = layernorm(q.contiguous(), layer.norm_self.weight, layer.norm_self.bias, res=so)
        co = _cross_attention(layer.visual_cross_attention, q1, V, H)
        q2 = layernorm(q1, layer.norm_cross_v.weight, layer.norm_cross_v.bias, res=co)
        ct = _cross_attention(layer.text_cross_attention, q2, t_token, H)
        q = layernorm(q2, layer.norm_cross_t.weight, layer.norm_cross_t.bias, res=ct)
    la = tok.tta_module.layer_linagg.linear_aggregator
    qq = linear(q, la.wq.weight, la.wq.bias)
    kk = linear(V, la.wk.weight, la.wk.bias)
    return AttnFn.apply(qq, kk, V, H, 1.0 / math.sqrt(E // H))


def spp_forward(m, x: torch.Tensor) -> torch.Tensor:
    """SpatialPoolingProjector.forward under autograd (spatial_pooling_projector.py:34-52)."""
    import torch.nn as nn
    ensure_gemm_scratch(x.device)
    g = m.num_patches_pre
    ps = m.pooling_size
    x = x.to(BF).contiguous()
    if m.pooling_type == "spatial":
        x = AvgPool3dFn.apply(x, tuple(g), (ps, ps, ps))
    else:
        x = AvgPool3dFn.apply(x, (1, 1, g[0] * g[1] * g[2]), (1, 1, ps ** 3))
    lins = [l for l in m.projector if isinstance(l, nn.Linear)]
    for i, lin in enumerate(lins):
        last = i == len(lins) - 1
        x = linear(x, lin.weight, lin.bias, gelu=(not last and m.layer_type == "mlp"))
    return x


def vit_forward(vit, images: torch.Tensor, keep_cls: bool) -> torch.Tensor:
    """ViT.forward + feature selection under autograd (vit.py:114-126,148-164, MONAI blocks).  Rows of a chunk are kept
    as [patches | cls] (the attention kernel's "extra row" is the last one); the reference order is restored on exit."""
    ensure_gemm_scratch(images.device)
    pe = vit.patch_embedding
    nc = images.shape[0]
    Hd, heads = vit.hidden_size, vit.num_heads
    patches = ops.im2col(images.contiguous(), tuple(vit.patch_size))              # (nc, ntok, p1 p2 p3); not differentiated
    lin = pe.patch_embeddings[1]
    x = linear(patches, lin.weight, lin.bias) + pe.position_embeddings.to(BF)
    x = torch.cat((x, vit.cls_token.to(BF).expand(nc, -1, -1)), dim=1).contiguous()   # (nc, ntok + 1, Hd)
    ntok = pe.n_patches
    scale = 1.0 / math.sqrt(Hd // heads)
    for blk in vit.blocks:
        h = layernorm(x, blk.norm1.weight, blk.norm1.bias, eps=blk.norm1.eps)
        qkv = linear(h, blk.attn.qkv.weight, None)
        att = SelfAttnFn.apply(qkv, None, heads, scale, 0, True)
        x = linear(att, blk.attn.out_proj.weight, blk.attn.out_proj.bias, res=x)
        h = layernorm(x, blk.norm2.weight, blk.norm2.bias, eps=blk.norm2.eps)
        h = linear(h, blk.mlp.linear1.weight, blk.mlp.linear1.bias, gelu=True)
        x = linear(h, blk.mlp.linear2.weight, blk.mlp.linear2.bias, res=x)
    x = layernorm(x, vit.norm.weight, vit.norm.bias, eps=vit.norm.eps)
    if keep_cls:
        return torch.cat((x[:, ntok:], x[:, :ntok]), dim=1)
    return x[:, :ntok]
