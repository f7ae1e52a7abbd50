"""CPU oracle for the u2Tokenizer forward path -- TEST INFRASTRUCTURE ONLY.

This file is a functional restatement (plain torch ops on CPU tensors, parameters read from a
state dict) of the reference's hot path `prepare_inputs_for_multimodal`
(/root/reference/src/model/u2_arch.py:96-117).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product (u2tokenizer_amd/) never does.

Every function cites the reference lines it follows and performs the same torch ops in the same
order, so that running it with bf16 tensors rounds at the same points as the reference does.

PINNING STATUS
  * tokenizer / projector / splice: pinned -- tests/golden/*.npz were produced by importing the
    reference's own modules (src/model/u2tokenizer/*, multimodal_projector/*) in the build container
    (tests/golden/make_golden.py) and `tests/test_oracle_golden.py` checks this file against them:
    forward in fp32 (and float64 to 1e-9 on the "lively" sets), and the BACKWARD of the tokenizer --
    torch.autograd over this file against the reference modules' own float64 gradients
    (tokenizer_*_grads.npz) to 1e-9, since the training-path tests differentiate this file.
  * ViT blocks: PARITY UNPINNED.  vit.py:19-20 imports MONAI 1.3.0 (requirements.txt:52), which is
    not installed and not vendored in /root/reference; `patch_embedding_block`, `sa_block`,
    `mlp_block` and `transformer_block` below restate MONAI's published semantics
    (monai/networks/blocks/{patchembedding,selfattention,mlp,transformerblock}.py @1.3.0).  The golden
    ViT vectors run the reference's own `ViT`/`ViT3DTower` classes on top of that same restatement,
    so they pin the composition (cls token, block loop, final norm, cls drop) but not the MONAI blocks.
    tests/test_oracle_independent.py cross-checks the restated blocks against independent torch
    formulations (nn.MultiheadAttention with permuted qkv weights for SABlock, nn.TransformerEncoderLayer
    for the block, Conv3d(kernel = stride = patch) for the perceptron patch embedding).
  * hard top-k: the reference's tie order is torch.topk's unspecified one and its scores depend on
    the BLAS summation order; the canonical definition used here (and by the HIP kernels) is
    score = fp32(exact dot product), order = descending score, ties by ascending index.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- configuration
@dataclass
class PathConfig:
    """Hyper-parameters of the path; defaults = shipped config
    (/root/reference/base_model_tokenizers/Llama-3.2-1B-Instruct/config.json:9-40)."""

    image_size: Sequence[int] = (32, 256, 256)
    patch_size: Sequence[int] = (4, 16, 16)
    image_channel: int = 1
    vit_hidden: int = 768
    vit_mlp: int = 3072
    vit_layers: int = 12
    vit_heads: int = 12
    vision_select_feature: str = "patch"
    proj_layer_type: str = "mlp"
    proj_layer_num: int = 2
    proj_pooling_type: str = "spatial"
    proj_pooling_size: int = 2
    hidden_size: int = 2048
    u2t_num_heads: int = 8
    u2t_num_layers: int = 4
    u2t_top_k: int = 1024
    use_multi_scale: bool = True
    num_3d_query_token: int = 256
    attn_type: str = "rma"
    enable_diffts: bool = True
    enable_dmtp: bool = True
    max_seq_len: int = 512
    enable_u2tokenizer: bool = True
    # True: DiffTS sums its weighted tokens the way svr.py:112-115 does (see diff_token_selection) -- only matters below fp32
    diffts_loop_form: bool = False

    @property
    def grid(self) -> List[int]:
        return [i // p for i, p in zip(self.image_size, self.patch_size)]

    @property
    def n_patches(self) -> int:
        g = self.grid
        return g[0] * g[1] * g[2]

    @property
    def proj_out_num(self) -> int:
        g = self.grid
        if self.proj_pooling_type == "spatial":
            return math.prod(n // self.proj_pooling_size for n in g)
        return self.n_patches // self.proj_pooling_size ** 3


def _lin(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# --------------------------------------------------------------------------- MONAI 1.3.0 blocks (restated)
def patch_embedding_block(sd: SD, p: str, x: torch.Tensor, patch: Sequence[int]) -> torch.Tensor:
    """MONAI PatchEmbeddingBlock(pos_embed="perceptron", spatial_dims=3), as built at vit.py:90-99:
    Rearrange("b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)") -> Linear -> + position_embeddings."""
    b, c, D, H, W = x.shape
    p1, p2, p3 = patch
    h, w, d = D // p1, H // p2, W // p3
    x = x.reshape(b, c, h, p1, w, p2, d, p3).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(b, h * w * d, p1 * p2 * p3 * c)
    x = _lin(x, sd, p + ".patch_embeddings.1")
    return x + sd[p + ".position_embeddings"]


def sa_block(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """MONAI SABlock (qkv_bias=False): qkv Linear, "b h (qkv l d) -> qkv b l h d", softmax(q k^T * d^-0.5) v,
    "b h l d -> b l (h d)", out_proj."""
    b, s, hd = x.shape
    dh = hd // heads
    qkv = _lin(x, sd, p + ".qkv").reshape(b, s, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (torch.einsum("blxd,blyd->blxy", q, k) * (dh ** -0.5)).softmax(dim=-1)
    o = torch.einsum("bhxy,bhyd->bhxd", att, v)
    o = o.permute(0, 2, 1, 3).reshape(b, s, hd)
    return _lin(o, sd, p + ".out_proj")


def mlp_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """MONAI MLPBlock: linear1 -> GELU (exact erf) -> linear2 (dropout 0)."""
    return _lin(F.gelu(_lin(x, sd, p + ".linear1")), sd, p + ".linear2")


def transformer_block(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """MONAI TransformerBlock: x = x + attn(norm1(x)); x = x + mlp(norm2(x))."""
    hd = x.shape[-1]
    x = x + sa_block(sd, p + ".attn", F.layer_norm(x, (hd,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]), heads)
    x = x + mlp_block(sd, p + ".mlp", F.layer_norm(x, (hd,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]))
    return x


# --------------------------------------------------------------------------- ViT3DTower
def vit_tower_forward(sd: SD, p: str, images: torch.Tensor, cfg: PathConfig) -> torch.Tensor:
    """ViT.forward (vit.py:114-126) + ViT3DTower.forward (vit.py:148-164); p = "...vision_tower.vision_tower"."""
    x = patch_embedding_block(sd, p + ".patch_embedding", images, cfg.patch_size)
    cls = sd[p + ".cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg.vit_layers):
        x = transformer_block(sd, f"{p}.blocks.{i}", x, cfg.vit_heads)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"])
    if cfg.vision_select_feature == "patch":
        x = x[:, 1:]
    elif cfg.vision_select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {cfg.vision_select_feature}")
    return x


# --------------------------------------------------------------------------- SpatialPoolingProjector
def spp_forward(sd: SD, p: str, x: torch.Tensor, cfg: PathConfig) -> torch.Tensor:
    """spatial_pooling_projector.py:34-52; p = "...mm_projector"."""
    B, n, dim = x.shape
    ps = cfg.proj_pooling_size
    if cfg.proj_pooling_type == "spatial":
        g = cfg.grid
        x = x.reshape(B, g[0], g[1], g[2], dim).permute(0, 4, 1, 2, 3)
        if x.dtype in (torch.bfloat16, torch.float16):  # CPU torch has no 16-bit avg_pool3d kernel: fp32 accumulate, one rounding
            x = F.avg_pool3d(x.float(), kernel_size=ps, stride=ps).to(x.dtype)
        else:
            x = F.avg_pool3d(x, kernel_size=ps, stride=ps)
        x = x.permute(0, 2, 3, 4, 1).reshape(B, -1, dim)
    elif cfg.proj_pooling_type == "sequence":
        x = F.avg_pool1d(x.permute(0, 2, 1), kernel_size=ps ** 3, stride=ps ** 3).permute(0, 2, 1)
    n2 = x.shape[1]
    x = x.reshape(B * n2, dim)
    for i in range(cfg.proj_layer_num):
        idx = 2 * i if cfg.proj_layer_type == "mlp" else i
        if i > 0 and cfg.proj_layer_type == "mlp":
            x = F.gelu(x)
        x = _lin(x, sd, f"{p}.projector.{idx}")
    return x.reshape(B, n2, -1)


# --------------------------------------------------------------------------- attention modules
def _split_heads(x: torch.Tensor, heads: int) -> torch.Tensor:
    b, s, e = x.shape
    return x.view(b, s, heads, e // heads).permute(0, 2, 1, 3)


def rma_attention(sd: SD, p: str, query, key, value, heads: int, max_seq_len: int = 512, is_compress=False):
    """RelativeMultiheadAttention.forward (rma.py:46-83).  Returns (output, attention_weights)."""
    b, s, e = query.shape
    depth = e // heads
    q = _split_heads(_lin(query, sd, p + ".wq"), heads)
    k = _split_heads(_lin(key, sd, p + ".wk"), heads)
    v = _split_heads(value if is_compress else _lin(value, sd, p + ".wv"), heads)
    scaling = torch.sqrt(torch.tensor(depth, dtype=q.dtype))                      # rma.py:60 (rounded to dtype)
    scores = torch.matmul(q, k.transpose(-2, -1)) / scaling
    pos = torch.arange(s)
    rel = pos[None, :] - pos[:, None] + max_seq_len - 1                          # rma.py:64-66
    bias = sd[p + ".relative_bias"][rel].permute(2, 0, 1).unsqueeze(0)            # rma.py:68-69
    scores = scores + bias
    w = F.softmax(scores, dim=-1)
    ctx = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(b, s, e)
    return (ctx if is_compress else _lin(ctx, sd, p + ".dense")), w


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rope_attention(sd: SD, p: str, query, key, value, heads: int, max_seq_len: int = 512):
    """RotaryMultiheadAttention.forward (rope.py:62-91); cos/sin cache as rope.py:33-40."""
    b, s, e = query.shape
    dh = e // heads
    q = _split_heads(_lin(query, sd, p + ".wq"), heads)
    k = _split_heads(_lin(key, sd, p + ".wk"), heads)
    v = _split_heads(_lin(value, sd, p + ".wv"), heads)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    t = torch.arange(max_seq_len, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos = emb.cos()[None, None, :s, :].to(q.dtype)
    sin = emb.sin()[None, None, :s, :].to(q.dtype)
    q = (q * cos) + (_rotate_half(q) * sin)
    k = (k * cos) + (_rotate_half(k) * sin)
    scores = torch.matmul(q, k.transpose(-2, -1)) / (dh ** 0.5)
    w = F.softmax(scores, dim=-1)
    ctx = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(b, s, e)
    return _lin(ctx, sd, p + ".dense"), w


def mha_attention(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """torch.nn.MultiheadAttention(embed, heads) called as m(x, x, x) with its default batch_first=False
    (svr.py:17-18,29,35; tta.py:84,94): dim 0 of x is the SEQUENCE, dim 1 the batch -- so the "spatial" call attends
    across chunks and the "temporal" call across tokens.  torch's own functional is the reference's
    dependency, not reference code: call it directly."""
    out, _ = F.multi_head_attention_forward(
        x, x, x, x.shape[-1], heads, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"], None, None, False, 0.0,
        sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"], training=False, need_weights=False)
    return out


def self_attention(sd, p, x, heads, attn_type, max_seq_len):
    if attn_type == "rma":
        return rma_attention(sd, p, x, x, x, heads, max_seq_len)[0]
    if attn_type == "rope":
        return rope_attention(sd, p, x, x, x, heads, max_seq_len)[0]
    return mha_attention(sd, p, x, heads)  # every other value builds nn.MultiheadAttention (svr.py:16-18, tta.py:83-84)


def cross_attention(sd: SD, p: str, query, value, heads: int, is_compress=False):
    """MultiHeadCrossAttention.forward (tta.py:42-69)."""
    b = query.shape[0]
    e = query.shape[-1]
    depth = e // heads
    q = _split_heads(_lin(query, sd, p + ".wq"), heads)
    k = _split_heads(_lin(value, sd, p + ".wk"), heads)
    v = _split_heads(value if is_compress else _lin(value, sd, p + ".wv"), heads)
    scaling = torch.sqrt(torch.tensor(depth, dtype=q.dtype))                      # tta.py:55
    scores = torch.matmul(q, k.transpose(-2, -1)) / scaling
    w = F.softmax(scores, dim=-1)
    ctx = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(b, -1, e)
    return ctx if is_compress else _lin(ctx, sd, p + ".dense")


# --------------------------------------------------------------------------- SVR
def st_attention_layer(sd: SD, p: str, x: torch.Tensor, cfg: PathConfig) -> torch.Tensor:
    """SpatioTemporalAttentionLayer.forward (svr.py:23-40): no residual, no norm."""
    b, t, n, e = x.shape
    x = x.reshape(b * t, n, e)
    x = self_attention(sd, p + ".spatial_attention", x, cfg.u2t_num_heads, cfg.attn_type, cfg.max_seq_len)
    x = x.view(b, t, n, e).permute(0, 2, 1, 3).contiguous().view(b * n, t, e)
    x = self_attention(sd, p + ".temporal_attention", x, cfg.u2t_num_heads, cfg.attn_type, cfg.max_seq_len)
    return x.view(b, n, t, e).permute(0, 2, 1, 3).contiguous()


def exact_scores(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """score_net = Linear(E, 1) (svr.py:67,78) with the canonical arithmetic: fp64 dot, one rounding to fp32."""
    s = x.double() @ w.double().reshape(-1)
    if bias is not None:
        s = s + bias.double().reshape(())
    return s.float()


def canonical_topk(scores: torch.Tensor, k: int) -> torch.Tensor:
    """torch.topk(..., sorted) (svr.py:82) with the canonical tie rule: ties by ascending index; -0.0 == +0.0."""
    s = scores + 0.0  # -0.0 -> +0.0
    return torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]


def token_selection(sd: SD, p: str, x: torch.Tensor, top_k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """TokenSelection.forward (svr.py:75-91). Returns (tokens (b,k,e), flat indices (b,k) int64)."""
    b, t, n, e = x.shape
    scores = exact_scores(x.reshape(b, t * n, e), sd[p + ".score_net.weight"], sd.get(p + ".score_net.bias"))
    idx = canonical_topk(scores, top_k)
    tok = x[torch.arange(b).unsqueeze(1), idx // n, idx % n]
    return tok, idx


def diff_token_selection(sd: SD, p: str, x: torch.Tensor, tau: float = 1.0, loop_form: bool = False) -> torch.Tensor:
    """DifferentiableTokenSelection.forward (svr.py:101-117).  The python loop over selection heads (svr.py:112-115:
    `token_r = torch.sum(w * x_flat, dim=1)` for r in range(top_k)) is, in exact arithmetic, the matrix product
    weights^T @ x_flat -- the default here, pinned against the reference module to 1e-9 in fp32 / float64.

    In bf16 the two are NOT the same arithmetic: the reference rounds every product `w * x_flat` to bf16 (an elementwise op on bf16
    tensors) before `torch.sum` adds them (fp32 accumulator inside the kernel, one rounding at the end), whereas a bf16 matmul keeps
    its products exact in the fp32 accumulator.  The matmul form is therefore MORE accurate than the reference's own bf16 run at this
    stage, and a bf16 yardstick built on it slightly understates the reference's bf16 distance downstream of the selection
    (VERDICT r5 weak #6).  loop_form=True performs the reference's op sequence (heads in blocks of 16 instead of one by one: the
    same elementwise products and the same per-(head, feature) sums over the tokens); tests/test_gpu_configs.py reports both
    distances at the sizes where the loop is affordable and gates against the smaller (stricter) one."""
    b, t, n, e = x.shape
    scores = _lin(x, sd, p + ".score_net").view(b, t * n, -1)
    weights = F.softmax(scores / tau, dim=1)
    x_flat = x.reshape(b, t * n, e)
    if not loop_form:
        return torch.matmul(weights.transpose(1, 2), x_flat)
    out, step = [], 16
    for r0 in range(0, weights.shape[2], step):
        w = weights[:, :, r0:r0 + step].unsqueeze(-1)            # (b, t*n, R, 1)
        out.append(torch.sum(w * x_flat.unsqueeze(2), dim=1))    # (b, R, e): products rounded in x's dtype, then summed over tokens
    return torch.cat(out, dim=1)


def multi_scale_pool(sd: SD, p: Optional[str], x: torch.Tensor, scales=(1, 2, 4)) -> torch.Tensor:
    """Fixed pooling (svr.py:176-184) when p is None, DynamicMultiScalePooling.forward (svr.py:126-151) otherwise."""
    pooled, gates = [], []
    for s in scales:
        if x.size(1) >= s:
            pl = F.avg_pool1d(x.permute(0, 2, 1), kernel_size=s, stride=s).permute(0, 2, 1)
            pooled.append(pl)
            if p is not None:
                gates.append(_lin(pl.mean(dim=1), sd, p + ".gate_fc"))
    if p is None:
        return torch.cat(pooled, dim=1)
    w = F.softmax(torch.cat(gates, dim=1), dim=1)
    return torch.cat([pl * w[:, i].unsqueeze(1).unsqueeze(2) for i, pl in enumerate(pooled)], dim=1)


def svr_forward(sd: SD, p: str, x: torch.Tensor, cfg: PathConfig):
    """SpatioTemporalVisualTokenRefinerModel.forward (svr.py:166-188). Returns (tokens, topk_idx | None)."""
    for l in range(cfg.u2t_num_layers):
        x = st_attention_layer(sd, f"{p}.attention_network.layers.{l}", x, cfg)
    idx = None
    if cfg.enable_diffts:
        x = diff_token_selection(sd, p + ".token_selection", x, loop_form=cfg.diffts_loop_form)
    else:
        x, idx = token_selection(sd, p + ".token_selection", x, cfg.u2t_top_k)
    if cfg.use_multi_scale:
        x = multi_scale_pool(sd, p + ".dynamic_pool" if cfg.enable_dmtp else None, x)
    return x, idx


# --------------------------------------------------------------------------- TTA
def tta_layer(sd: SD, lp: str, query, visual, text, cfg: PathConfig) -> torch.Tensor:
    """TextConditionTokenAttMap.forward (tta.py:93-107); lp = "...tta_module.layers_vt.N"."""
    H = cfg.u2t_num_heads
    e = query.shape[-1]
    so = self_attention(sd, lp + ".self_attention", query, H, cfg.attn_type, cfg.max_seq_len)
    so = F.layer_norm(query + so, (e,), sd[lp + ".norm_self.weight"], sd[lp + ".norm_self.bias"])
    co = cross_attention(sd, lp + ".visual_cross_attention", so, visual, H)
    cv = F.layer_norm(so + co, (e,), sd[lp + ".norm_cross_v.weight"], sd[lp + ".norm_cross_v.bias"])
    ct = cross_attention(sd, lp + ".text_cross_attention", cv, text, H)
    return F.layer_norm(cv + ct, (e,), sd[lp + ".norm_cross_t.weight"], sd[lp + ".norm_cross_t.bias"])


def tta_forward(sd: SD, p: str, query, visual, text, cfg: PathConfig) -> torch.Tensor:
    """TextConditionTokenAggregatorModel.forward (tta.py:126-140) with TextConditionTokenAttMap.forward
    (tta.py:93-107) and LinearAggregation.forward (tta.py:114-116)."""
    for l in range(cfg.u2t_num_layers):
        query = tta_layer(sd, f"{p}.layers_vt.{l}", query, visual, text, cfg)
    return cross_attention(sd, p + ".layer_linagg.linear_aggregator", query, visual, cfg.u2t_num_heads, is_compress=True)


def tokenizer_forward(sd: SD, p: str, v_token: torch.Tensor, t_token: torch.Tensor, cfg: PathConfig):
    """u2Tokenizer.forward (u2Tokenizer.py:40-47); p = "...u2tokenizer". Returns (aligned, topk_idx | None)."""
    B = v_token.shape[0]
    q = sd[p + ".query_tokens"].expand(B, -1, -1)
    v, idx = svr_forward(sd, p + ".svt_module", v_token, cfg)
    return tta_forward(sd, p + ".tta_module", q, v, t_token, cfg), idx


# --------------------------------------------------------------------------- the whole path
def prepare_inputs_for_multimodal(sd: SD, embed_w: torch.Tensor, input_ids, images, question_ids, cfg: PathConfig,
                                  prefix: str = "model."):
    """u2MetaForCausalLM.prepare_inputs_for_multimodal (u2_arch.py:96-117) -> inputs_embeds (and the
    hard top-k indices, when that selection mode is on).  `embed_w` = get_model().embed_tokens.weight."""
    if cfg.enable_u2tokenizer:
        B, C, D, H, W = images.shape
        feats = vit_tower_forward(sd, prefix + "vision_tower.vision_tower", images.view(B * C, 1, D, H, W), cfg)
        feats = spp_forward(sd, prefix + "mm_projector", feats, cfg)
        v_tokens = feats.view(B, C, feats.shape[-2], feats.shape[-1])
        t_tokens = F.embedding(question_ids, embed_w)
        feats, idx = tokenizer_forward(sd, prefix + "u2tokenizer", v_tokens, t_tokens, cfg)
    else:
        feats = vit_tower_forward(sd, prefix + "vision_tower.vision_tower", images, cfg)
        feats = spp_forward(sd, prefix + "mm_projector", feats, cfg)
        idx = None
    emb = F.embedding(input_ids, embed_w)
    emb = torch.cat((emb[:, :1, :], feats, emb[:, feats.shape[1] + 1:, :]), dim=1)
    return emb, idx
