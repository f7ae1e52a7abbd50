"""GPU: every HIP building block against an fp32 CPU computation of the same bf16 inputs (through the C ABI).

Tolerances: a bf16 result is allowed TWO bf16 roundings of the exact value (|err| <= 2 * 2^-8 * |ref|, plus
2^-8 of the tensor's max for values near zero); fp32 results 1e-5 of the tensor's max; integer / index /
data-movement results are bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import u2_oracle as O

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from u2tokenizer_amd import ops as _ops
    _ops.device_check()
    return _ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed * 7919 + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(bf)


ULP = 2.0 ** -8   # of the element type under test (tests/test_gpu_f16.py re-runs this module's cases with bf = float16, 2^-11)


def close_bf16(got, ref, rounds=2):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all()
    tol = rounds * ULP * ref.abs() + ULP * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{bad.sum().item()} elements off; worst {(got - ref).abs().max().item():.3e}"


def close_f32(got, ref):
    got, ref = got.float().cpu(), ref.float()
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-7


D = "cuda"


@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_shapes_and_epilogues(ops, tile):
    ops.set_option("gemm_tile", tile)
    try:
        for (M, N, K) in [(128, 128, 64), (300, 200, 136), (77, 520, 72), (1, 8, 8), (2049, 768, 768)]:
            a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
            close_bf16(ops.gemm(a.to(D), b.to(D)), a.float() @ b.float().t())
        M, N, K = 300, 264, 200
        a, b, bias, res, bm = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6), rnd(M, seed=7)
        base = a.float() @ b.float().t()
        close_bf16(ops.gemm(a.to(D), b.to(D), bias=bias.to(D), residual=res.to(D), gelu=True),
                   F.gelu(base + bias.float()) + res.float(), rounds=3)
        close_f32(ops.gemm(a.to(D), b.to(D), bias=bias.to(D), out_f32=True, alpha=0.5), 0.5 * base + bias.float())
        close_f32(ops.gemm(a.to(D), b.to(D), bias=bm.to(D), bias_m=True, out_f32=True), base + bm.float()[:, None])
        b2, bias2 = rnd(203, K, seed=8), rnd(203, seed=9)  # N % 4 != 0 -> scalar epilogue
        close_bf16(ops.gemm(a.to(D), b2.to(D), bias=bias2.to(D)), a.float() @ b2.float().t() + bias2.float())
        a3, b3 = rnd(6, 100, 64, seed=10), rnd(6, 90, 64, seed=11)
        close_f32(ops.gemm(a3.to(D), b3.to(D), out_f32=True), torch.einsum("zmk,znk->zmn", a3.float(), b3.float()))
    finally:
        ops.set_option("gemm_tile", 0)


def test_gemm_full_size_is_exact_on_integer_data(ops):
    """ViT QKV shape of BASELINE config 3 (M = 8 x 2049): small-integer operands make every partial sum exactly
    representable, so the fp32 result must equal the CPU product bit for bit (catches any dropped/duplicated k)."""
    M, N, K = 8 * 2049, 2304, 768
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-4, 5, (M, K), generator=g).to(bf)
    b = torch.randint(-4, 5, (N, K), generator=g).to(bf)
    got = ops.gemm(a.to(D), b.to(D), out_f32=True).cpu()
    assert torch.equal(got, a.float() @ b.float().t())


def test_gemm_rejects_bad_arguments(ops):
    a, b = rnd(16, 12).to(D), rnd(16, 12).to(D)  # K % 8 != 0
    with pytest.raises(RuntimeError, match="U2TOK_ERR_ARG"):
        ops.gemm(a, b)


@pytest.mark.parametrize("C_", [512, 768, 2048, 4096])
def test_layernorm(ops, C_):
    x, r, w, b = rnd(37, C_, seed=1), rnd(37, C_, seed=2), rnd(C_, seed=3), rnd(C_, seed=4)
    close_bf16(ops.layernorm(x.to(D), w.to(D), b.to(D)), F.layer_norm(x.float(), (C_,), w.float(), b.float()))
    close_bf16(ops.layernorm(x.to(D), w.to(D), b.to(D), residual=r.to(D)),
               F.layer_norm(x.float() + r.float(), (C_,), w.float(), b.float()))


def test_softmax_rows_scale_bias_and_padding(ops):
    for (Z, R, n) in [(8, 40, 40), (3, 17, 1792), (4, 9, 13), (1, 1, 1)]:
        s = torch.randn(Z, R, n, generator=torch.Generator().manual_seed(n)) * 3
        got = ops.softmax_rows(s.to(D), scale=0.7, elem=bf)
        close_bf16(got[:, :, :n], F.softmax(s * 0.7, dim=-1))
        assert got.shape[2] % 8 == 0 and (got[:, :, n:] == 0).all()
        close_f32(got.float().sum(-1), torch.ones(Z, R)) if False else None
    H, L = 4, 512
    tbl = rnd(2 * L - 1, H, scale=0.5, seed=5)
    s = torch.randn(2 * H, 40, 40, generator=torch.Generator().manual_seed(7))
    got = ops.softmax_rows(s.to(D), scale=0.5, rel_bias=tbl.to(D), heads=H, max_len=L, elem=bf)
    pos = torch.arange(40)
    bias = tbl.float()[pos[None, :] - pos[:, None] + L - 1].permute(2, 0, 1)  # rma.py:64-69
    close_bf16(got[:, :, :40], F.softmax(s.view(2, H, 40, 40) * 0.5 + bias[None], dim=-1).view(2 * H, 40, 40))
    with pytest.raises(RuntimeError, match="U2TOK_ERR_ARG"):  # seq_len > max_seq_len cannot index the bias table
        ops.softmax_rows(torch.zeros(H, 600, 600, device=D), rel_bias=tbl.to(D), heads=H, max_len=L, elem=bf)


def test_transpose_and_data_movement_are_bit_exact(ops):
    x = rnd(3, 70, 130, seed=8)
    got = ops.transpose(x.to(D), ld_out=72).cpu()
    assert torch.equal(got[:, :, :70], x.transpose(1, 2)) and (got[:, :, 70:] == 0).all()
    x = rnd(2, 203, 192, seed=81)  # strides % 4 == 0: the 8-byte vector kernel; ragged rows, several tiles
    got = ops.transpose(x.to(D), ld_out=208).cpu()
    assert torch.equal(got[:, :, :203], x.transpose(1, 2)) and (got[:, :, 203:] == 0).all()
    got = ops.transpose(x.to(D), ld_out=208, perm16=True).cpu()
    want = torch.zeros(2, 192, 208, dtype=bf)
    want[:, :, :203] = x.transpose(1, 2)
    order = [j for g in range(13) for j in (list(range(16 * g, 16 * g + 4)) + list(range(16 * g + 8, 16 * g + 12)) +
                                             list(range(16 * g + 4, 16 * g + 8)) + list(range(16 * g + 12, 16 * g + 16)))]
    assert torch.equal(got, want[:, :, order])
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        vol = torch.rand(2, 1, 8, 32, 32, generator=torch.Generator().manual_seed(3)).to(dt)
        ref = vol.to(bf).reshape(2, 1, 2, 4, 2, 16, 2, 16).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(2, 8, 1024)
        assert torch.equal(ops.im2col(vol.to(D), (4, 16, 16), elem=bf).cpu(), ref)
    table = rnd(100, 64, seed=10)
    ids = torch.randint(0, 100, (2, 12), generator=torch.Generator().manual_seed(1))
    feats = rnd(2, 5, 64, seed=11)
    emb = table[ids]
    assert torch.equal(ops.embed_splice(table.to(D), ids.to(D), feats.to(D)).cpu(),
                       torch.cat((emb[:, :1], feats, emb[:, 6:]), 1))  # u2_arch.py:115-116
    assert torch.equal(ops.embed_splice(table.to(D), ids.to(D)).cpu(), emb)
    x = rnd(2, 50, 64, seed=15)
    idx = torch.randint(0, 50, (2, 20), generator=torch.Generator().manual_seed(2))
    assert torch.equal(ops.gather_rows(x.to(D), idx.to(D)).cpu(), x[torch.arange(2)[:, None], idx])


def test_im2col_full_volume_is_the_reference_permutation(ops):
    """256^3 fp16 volume = 8 chunks of (32,256,256): exact equality with the einops pattern of vit.py:90-99."""
    vol = torch.rand(8, 1, 32, 256, 256, generator=torch.Generator().manual_seed(9)).half()
    ref = vol.to(bf).reshape(8, 1, 8, 4, 16, 16, 16, 16).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(8, 2048, 1024)
    assert torch.equal(ops.im2col(vol.to(D), (4, 16, 16), elem=bf).cpu(), ref)


def test_avgpool(ops):
    x = rnd(2, 8 * 4 * 4, 768, seed=9)
    ref = F.avg_pool3d(x.float().view(2, 8, 4, 4, 768).permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1).reshape(2, -1, 768)
    close_bf16(ops.avgpool3d_tokens(x.to(D), (8, 4, 4), (2, 2, 2)), ref, rounds=1)
    close_bf16(ops.avgpool3d_tokens(x.to(D), (1, 1, 128), (1, 1, 8)),
               F.avg_pool1d(x.float().permute(0, 2, 1), 8, 8).permute(0, 2, 1), rounds=1)


def test_selection_stage_is_bit_exact(ops):
    """score -> top-k -> gather on identical inputs == oracle, at BASELINE size (2048 tokens, E=4096, k=1024) and
    with heavy exact ties / signed zeros."""
    x, w, b = rnd(2, 2048, 4096, seed=12), rnd(1, 4096, scale=0.02, seed=13), rnd(1, seed=14)
    sc = ops.score_gemv(x.to(D), w.to(D), b.to(D)).cpu()
    assert torch.equal(sc, O.exact_scores(x, w, b))
    idx = ops.topk_sorted(sc.to(D), 1024).cpu()
    assert torch.equal(idx, O.canonical_topk(sc, 1024))
    tok, oidx = O.token_selection({"p.score_net.weight": w, "p.score_net.bias": b}, "p", x.view(2, 8, 256, 4096), 1024)
    assert torch.equal(idx, oidx) and torch.equal(ops.gather_rows(x.to(D), idx.to(D)).cpu(), tok)
    picked = sc.gather(1, idx)
    assert (picked[:, :-1] >= picked[:, 1:]).all() and all(len(set(r.tolist())) == 1024 for r in idx)
    for (B, n, k) in [(3, 200, 50), (1, 32, 16), (2, 33, 33), (1, 1, 1), (2, 4096, 7)]:
        s = (torch.randn(B, n, generator=torch.Generator().manual_seed(n)) * 4).round() / 4
        if n > 2:
            s[0, 1], s[0, 2] = -0.0, 0.0
        assert torch.equal(ops.topk_sorted(s.to(D), k).cpu(), O.canonical_topk(s, k))


def test_multiscale_pool_fixed_and_gated(ops):
    for k in (64, 30, 31, 3, 1):   # (31, 30, 3: the tokens past the last group of four)
        x = rnd(2, k, 512, seed=16)
        close_bf16(ops.multiscale_pool(x.to(D)), O.multi_scale_pool({}, None, x.float()), rounds=1)
        gw, gb = rnd(1, 512, scale=0.3, seed=17), rnd(1, seed=18)
        sd = {"p.gate_fc.weight": gw.float(), "p.gate_fc.bias": gb.float()}
        close_bf16(ops.multiscale_pool(x.to(D), gw.to(D), gb.to(D)), O.multi_scale_pool(sd, "p", x.float()))


def test_rope(ops):
    x = rnd(2 * 3 * 5, 768, seed=19)
    xd = x.to(D).clone()
    ops.rope_apply(xd[:, :256], 2, 3, 5, 4, 64)  # rows (b t n), position = t, 4 heads x 64
    xx = x.float()[:, :256].view(2, 3, 5, 4, 64).permute(0, 2, 3, 1, 4)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    emb = torch.cat((torch.einsum("i,j->ij", torch.arange(512.0), inv),) * 2, -1)
    cos, sin = emb.cos()[:3].to(bf).float(), emb.sin()[:3].to(bf).float()
    close_bf16(xd[:, :256].float().cpu().view(2, 3, 5, 4, 64).permute(0, 2, 3, 1, 4), xx * cos + O._rotate_half(xx) * sin)
    assert torch.equal(xd[:, 256:].cpu(), x[:, 256:])


@pytest.mark.parametrize("B,T,N,H,d", [(1, 8, 6, 8, 512), (2, 4, 5, 8, 256), (1, 2, 16, 8, 64), (1, 3, 7, 4, 128),
                                       (1, 16, 3, 2, 64), (1, 1, 4, 2, 64)])  # T > 16 / other head dims: pipeline path
def test_temporal_attention(ops, B, T, N, H, d):
    E = H * d
    qkv, tbl = rnd(B * T * N, 3 * E, seed=d), rnd(1023, H, scale=0.5, seed=3)
    qd = qkv.to(D)
    got = ops.temporal_attention(qd[:, :E], qd[:, E:2 * E], qd[:, 2 * E:], B, T, N, H, 1 / math.sqrt(d), tbl.to(D), 512)
    x = qkv.float().view(B, T, N, 3, H, d).permute(3, 0, 2, 4, 1, 5)
    pos = torch.arange(T)
    bias = tbl.float()[pos[None, :] - pos[:, None] + 511].permute(2, 0, 1)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) / math.sqrt(d) + bias[None, None], dim=-1)
    close_bf16(got, (p @ x[2]).permute(0, 3, 1, 2, 4).reshape(B * T * N, E))


def _sdpa_ref(qkv, nb, S, H):
    x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    return (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, H * 64)


@pytest.mark.parametrize("nb,S,H,scale", [(1, 64, 1, 1.0), (2, 129, 3, 1.0), (1, 513, 12, 1.0), (1, 2049, 2, 1.0),
                                          (3, 100, 12, 1.0), (1, 300, 2, 3.0), (1, 1, 1, 1.0)])
@pytest.mark.parametrize("mode", [1, 7])
def test_flash_attention(ops, nb, S, H, scale, mode):
    """scale 3.0 makes the logits spiky so the online-softmax rescale branch does real work.  mode 1 = plain 128-row
    units, mode 7 = the double pipeline (256-row units, generated asm KV loop)."""
    qkv = rnd(nb, S, 3 * H * 64, scale=scale, seed=S)
    ops.set_option("flash_mode", mode)
    try:
        got = ops.flash_attention_d64(qkv.to(D), H, 0.125)
    finally:
        ops.set_option("flash_mode", 0)
    close_bf16(got, _sdpa_ref(qkv, nb, S, H))


@pytest.mark.parametrize("nb,S,H,scale,mode", [(2, 129, 3, 1.0, 1), (1, 513, 12, 1.0, 1), (3, 257, 2, 1.0, 1),
                                               (1, 2049, 3, 1.0, 0), (1, 2049, 3, 1.0, 1),
                                               (3, 101, 12, 3.0, 0), (1, 2, 1, 1.0, 0),
                                               (1, 2049, 3, 1.0, 7), (2, 321, 3, 3.0, 7), (1, 66, 2, 1.0, 7), (1, 2, 1, 1.0, 7),
                                               (2, 577, 3, 1.0, 7), (1, 34, 1, 3.0, 7), (1, 97, 2, 3.0, 7), (1, 1025, 2, 1.0, 7)])
def test_flash_attention_extra_row(ops, nb, S, H, scale, mode):
    """The ViT path: S - 1 tiled main rows + one "extra" row per batch (the cls token) as key AND query.  The result
    must equal plain attention over all S rows."""
    qkv = rnd(nb, S, 3 * H * 64, scale=scale, seed=S + 1)
    ops.set_option("flash_mode", mode)
    try:
        got = ops.flash_attention_d64(qkv.to(D), H, 0.125, extra_last=True)
    finally:
        ops.set_option("flash_mode", 0)
    close_bf16(got, _sdpa_ref(qkv, nb, S, H))


def test_flash_attention_extra_key_forces_rescale(ops):
    """The extra key dominates every query: the rescale branch of the post-loop VALU step must fire."""
    nb, S, H = 1, 257, 3
    qkv = rnd(nb, S, 3 * H * 64, seed=91)
    for h in range(H):
        qkv[0, S - 1, 64 * H + 64 * h: 64 * H + 64 * (h + 1)] = (qkv[0, :, 64 * h:64 * (h + 1)].float().mean(0) * 60).to(bf)
    got = ops.flash_attention_d64(qkv.to(D), H, 0.125, extra_last=True)
    close_bf16(got, _sdpa_ref(qkv, nb, S, H))


def test_flash_attention_forced_rescale(ops):
    """One key row spiked against every query at a late KV tile: the running max must jump there (rule 26 of the
    CDNA guide: a rare data-dependent branch needs an input that forces it)."""
    nb, S, H = 1, 400, 1
    qkv = rnd(nb, S, 192, seed=77)
    qkv[0, 333, 64:128] = (qkv[0, :, :64].float().mean(0) * 40).to(bf)  # k row 333 aligned with the mean query
    got = ops.flash_attention_d64(qkv.to(D), H, 0.125)
    x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    close_bf16(got, (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, 64))


C_LOG2E = 0.125 * 1.4426950408889634


def _prescale_q(qkv, H):
    """what the ViT's q|k|v product leaves in the q columns: q * scale * log2 e, rounded to bf16 once"""
    out = qkv.clone()
    out[..., :64 * H] = (qkv[..., :64 * H].float() * C_LOG2E).to(bf)
    return out


def _sdpa_ref_prescaled(qkv, nb, S, H):
    x = qkv.double().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * math.log(2.0), dim=-1)      # the kernel computes exp2(q' . k)
    return (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, H * 64).float()


@pytest.mark.parametrize("nb,S,H,extra,gain", [(1, 513, 12, True, 0.0), (1, 2049, 2, True, 0.0), (2, 640, 3, False, 0.0),
                                                (1, 66, 2, True, 0.0), (1, 1100, 2, False, 12.0), (1, 1537, 2, True, 30.0)])
def test_flash_attention_prescaled_queries(ops, nb, S, H, extra, gain):
    """Option flash_q_prescaled: the q handed over already carries scale * log2 e (the ViT pipeline's q|k|v product scales
    its q columns from the fp32 accumulator) and the loop drops its per-score multiply.  Exact against a float64 reference
    on the SAME pre-scaled inputs -- plain, ragged, short, and with keys that force the out-of-line rescale."""
    qkv = rnd(nb, S, 3 * H * 64, seed=S)
    if gain:
        n = S - 1 if extra else S
        for h in range(H):
            for i, (kj, qi) in enumerate([(70, 5), (100, 40), (700, 300), (n - 3, 77), (n - 40, 250), (333, 250), (900, 3)]):
                q = qkv[0, qi, 64 * h:64 * (h + 1)].float()
                qkv[0, kj, 64 * H + 64 * h: 64 * H + 64 * (h + 1)] = (q * gain * (1 + 0.5 * (i % 3))).to(bf)
    qp = _prescale_q(qkv, H)
    ops.set_option("flash_q_prescaled", 1)
    try:
        got = ops.flash_attention_d64(qp.to(D), H, 0.125, extra_last=extra)
    finally:
        ops.set_option("flash_q_prescaled", 0)
    close_bf16(got, _sdpa_ref_prescaled(qp, nb, S, H))


def _sdpa_ref64(qkv, nb, S, H):
    x = qkv.double().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    return (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, H * 64).float()


@pytest.mark.parametrize("S,extra", [(1536, False), (1537, True), (1100, False)])
@pytest.mark.parametrize("gain", [4.0, 12.0, 30.0])
def test_flash_mode7_out_of_line_rescale(ops, S, extra, gain):
    """Mode 7 keeps its running max only within 2^64 of the true one and tests a row-sum piece instead of taking row
    maxima; the exact max, the rescale of O / l / the C tuple and the redone exponentials live out of line.  Inputs that
    force that path at chosen places: key j = gain x (a query row), so that row's score jumps by ~8 gain natural units
    (gain 12: 2^138 over the first keys' max -- above the 2^64 test; gain 30: 2^346, v_exp_f32 returns +inf first),
    at keys in the first / second half tile, in both 32-row blocks of a wave, in late tiles and in the last, partial one,
    twice in a row with growing size; gain 4 stays under the test (stale max, no rescale).  Checked against float64."""
    nb, H = 1, 2
    qkv = rnd(nb, S, 3 * H * 64, seed=S + int(gain))
    n = S - 1 if extra else S
    spikes = [(70, 5), (100, 40), (700, 300), (n - 3, 77), (n - 40, 250), (333, 250), (900, 3)]  # (key row, query row)
    for h in range(H):
        for i, (kj, qi) in enumerate(spikes):
            q = qkv[0, qi, 64 * h:64 * (h + 1)].float()
            qkv[0, kj, 64 * H + 64 * h: 64 * H + 64 * (h + 1)] = (q * gain * (1 + 0.5 * (i % 3))).to(bf)
    ops.set_option("flash_mode", 7)
    try:
        got = ops.flash_attention_d64(qkv.to(D), H, 0.125, extra_last=extra)
    finally:
        ops.set_option("flash_mode", 0)
    assert torch.isfinite(got.float()).all()
    close_bf16(got, _sdpa_ref64(qkv, nb, S, H))


def test_flash_mode7_first_keys_dominate(ops):
    """The other direction: the first 32 keys hold a row's maximum by a wide margin, every later p underflows to 0."""
    nb, S, H = 1, 1024, 1
    qkv = rnd(nb, S, 192, seed=5)
    for qi in (3, 40, 200, 777):
        qkv[0, qi % 32, 64:128] = (qkv[0, qi, :64].float() * 25).to(bf)
    ops.set_option("flash_mode", 7)
    try:
        got = ops.flash_attention_d64(qkv.to(D), H, 0.125)
    finally:
        ops.set_option("flash_mode", 0)
    close_bf16(got, _sdpa_ref64(qkv, nb, S, H))


def test_flash_mode7_log_sum_exp(ops):
    """The row statistics the fused backward takes: log2 sum_k exp2(s_k scale log2 e), against float math."""
    nb, S, H = 2, 1281, 3
    qkv = rnd(nb, S, 3 * H * 64, seed=11).to(D)
    out, lse = ops.flash_attention_d64(qkv, H, 0.125, extra_last=True, return_lse=True)
    ref = torch.logsumexp((lambda x: x[0] @ x[1].transpose(-1, -2) * 0.125)(
        qkv.float().cpu().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)), dim=-1) * 1.4426950408889634
    assert (lse.cpu()[:, :S] - ref.reshape(nb * H, S)).abs().max().item() < 2e-3


def test_flash_wide_and_narrow_stores_agree(ops):
    """16-byte output stores need 16-byte aligned rows; an output view at an 8-byte offset takes the 8-byte form."""
    from u2tokenizer_amd import _lib
    h = _lib.load_library(ops.ELEM_OF[bf])
    nb, S, H = 1, 640, 2
    Hd = 64 * H
    qkv = rnd(nb, S, 3 * Hd, seed=4).to(D)
    want = ops.flash_attention_d64(qkv, H, 0.125)
    vt = torch.empty((nb, Hd, S), dtype=bf, device=D)
    _lib.check(h.u2tok_transpose_bf16(qkv[:, :, 2 * Hd:].data_ptr(), vt.data_ptr(), nb, S, Hd, 3 * Hd, S, S * 3 * Hd, Hd * S, 1,
                                      None), "transpose")
    buf = torch.zeros(nb * S * (Hd + 4) + 8, dtype=bf, device=D)
    out = buf[4:4 + nb * S * (Hd + 4)].view(nb, S, Hd + 4)[:, :, :Hd]          # rows at 8-byte, not 16-byte, alignment
    es = 2
    _lib.check(h.u2tok_flash_attention_d64(qkv.data_ptr(), qkv.data_ptr() + Hd * es, vt.data_ptr(), out.data_ptr(), nb, S, H,
                                           3 * Hd, S * 3 * Hd, Hd + 4, S * (Hd + 4), S, 0.125, None, None, None, None, 0, 0, 0,
                                           None), "flash")
    torch.cuda.synchronize()
    assert torch.equal(out.contiguous(), want)


def _tok_attn_ref(q, k, v, H, scale, tbl=None, L=512):
    nb, Sq, E = q.shape
    Skv, d = k.shape[1], E // H
    qh, kh, vh = (t.float().view(nb, -1, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if tbl is not None:
        s = s + tbl.float()[torch.arange(Skv)[None, :] - torch.arange(Sq)[:, None] + L - 1].permute(2, 0, 1)[None]
    return (F.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(nb, Sq, E)


@pytest.mark.parametrize("nb,Sq,Skv,H,d,bias,splits", [
    (8, 256, 256, 8, 512, True, 0),     # SVR spatial attention of a 256^3 volume at E = 4096 (rma.py:60-75)
    (1, 256, 256, 8, 512, True, 0),     # TTA self attention (key splits by heuristic)
    (1, 256, 1792, 8, 512, False, 0),   # TTA visual cross attention / linear aggregation (tta.py:55-61)
    (1, 256, 1792, 8, 512, False, 1),   # ... unsplit
    (1, 256, 1792, 8, 512, False, 5),   # ... an uneven split (56 tiles over 5 -> 12, 12, 12, 12, 8)
    (1, 256, 1024, 8, 256, False, 0),   # text cross attention at E = 2048
    (4, 64, 64, 8, 256, True, 0),       # configuration-2 sized spatial attention
    (2, 37, 53, 4, 64, True, 0),        # tails everywhere: partial query block, partial key tile, 128-byte tile rows
    (2, 37, 53, 4, 64, True, 2),
    (3, 100, 70, 2, 128, False, 3),
    (1, 1, 1, 1, 64, True, 0), (1, 16, 33, 1, 128, False, 0), (1, 512, 512, 2, 64, True, 4),
    # wide heads (8-wave form: a wave pair per 16-query block) with tails everywhere, one-tile and odd tile counts, splits
    (2, 37, 53, 2, 256, True, 0), (2, 37, 53, 2, 256, True, 2), (1, 100, 70, 1, 512, False, 3), (1, 1, 1, 1, 512, True, 0),
    (1, 130, 97, 2, 512, True, 0), (3, 64, 32, 1, 256, False, 0), (1, 16, 33, 1, 512, False, 2), (1, 300, 480, 2, 512, True, 5),
    # the merge kernel's compile-time split counts 6, 7 and the run-time form behind 8 (round 6)
    (1, 128, 384, 2, 128, False, 6), (1, 64, 448, 2, 256, False, 7), (1, 64, 640, 1, 512, False, 10)])
def test_tok_attention(ops, nb, Sq, Skv, H, d, bias, splits):
    """The fused attention core of the tokenizer (u2tok_tok_attention) against an fp32 softmax(q k^T scale + bias) v of the
    same bf16 inputs; q / k / v are column slices of packed buffers, as the pipeline passes them; three launches must
    agree bit for bit (the key-split merge has a fixed order)."""
    E = H * d
    qkv = rnd(nb, Sq, 3 * E, seed=Sq + d) if Sq == Skv else None
    if qkv is not None:
        q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
        dq = qkv.to(D)
        dq, dk, dv = dq[..., :E], dq[..., E:2 * E], dq[..., 2 * E:]
    else:
        q, kv = rnd(nb, Sq, E, seed=Sq), rnd(nb, Skv, 2 * E, seed=Skv)
        k, v = kv[..., :E], kv[..., E:]
        dq, dkv = q.to(D), kv.to(D)
        dk, dv = dkv[..., :E], dkv[..., E:]
    tbl = rnd(1023, H, scale=0.5, seed=3) if bias else None
    scale = 2.0 / math.sqrt(d)  # a little spikier than 1 / sqrt(d): the softmax is not flat
    got = [ops.tok_attention(dq, dk, dv, H, scale, None if tbl is None else tbl.to(D), 512, splits) for _ in range(3)]
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])
    close_bf16(got[0], _tok_attn_ref(q, k, v, H, scale, tbl))
    if d >= 256:  # the 4-wave form of the same head dims (option tok_wide = 0) stays a tested path; and the 8-wave form's FLAT-encoded
        # DMA (tok_wide = 1; the default 2 issues buffer_load ... lds) moves the same bytes: same bits
        try:
            ops.set_option("tok_wide", 1)
            assert torch.equal(ops.tok_attention(dq, dk, dv, H, scale, None if tbl is None else tbl.to(D), 512, splits), got[0])
            ops.set_option("tok_wide", 0)
            close_bf16(ops.tok_attention(dq, dk, dv, H, scale, None if tbl is None else tbl.to(D), 512, splits),
                       _tok_attn_ref(q, k, v, H, scale, tbl))
        finally:
            ops.set_option("tok_wide", 2)


def test_tok_attention_forced_rescale_and_split_merge(ops):
    """One key row aligned with every query at a late tile (the running max jumps there: CDNA guide rule 26), and in the
    split form that key sits in the last split, so the merge weights of all other splits are ~2^-100."""
    nb, Sq, Skv, H, d = 1, 128, 480, 2, 256
    q, kv = rnd(nb, Sq, H * d, seed=5), rnd(nb, Skv, 2 * H * d, seed=6)
    for h in range(H):
        kv[0, 433, h * d:(h + 1) * d] = (q[0, :, h * d:(h + 1) * d].float().mean(0) * 40 + 3 * q[0, 7, h * d:(h + 1) * d].float()).to(bf)
    k, v = kv[..., :H * d], kv[..., H * d:]
    ref = _tok_attn_ref(q, k, v, H, 1 / math.sqrt(d))
    dkv = kv.to(D)
    for splits in (1, 3, 5):
        close_bf16(ops.tok_attention(q.to(D), dkv[..., :H * d], dkv[..., H * d:], H, 1 / math.sqrt(d), None, 512, splits), ref)


def test_tok_attention_rejects_what_it_does_not_take(ops):
    q = rnd(1, 8, 96, seed=1).to(D)  # head dim 48
    with pytest.raises(RuntimeError):
        ops.tok_attention(q, q, q, 2, 1.0)
    big = rnd(1, 600, 64, seed=2).to(D)  # relative bias beyond the table
    with pytest.raises(RuntimeError):
        ops.tok_attention(big, big, big, 1, 1.0, rnd(1023, 1, seed=3).to(D), 512)


@pytest.mark.parametrize("split", [0, 2, 3, 8])
def test_gemm_split_k(ops, split):
    """Skinny products (few output tiles, long K) cut along K into fp32 partial sums + a reduce kernel that applies the
    epilogue: every epilogue, row / column tails, K slices that do not divide, bit-repeatable (fixed summation order).
    split = 0 is the heuristic (takes M = 256, N = K = 2048 and leaves the 1000 x 1000 product alone)."""
    scratch = torch.empty(48 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)
    ops.set_option("gemm_splitk", split)
    try:
        for (M, N, K) in [(256, 2048, 2048), (77, 520, 1160), (256, 1000, 4096), (1000, 1000, 1024)]:
            a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
            ad, bd = a.to(D), b.to(D)
            outs = [ops.gemm(ad, bd).clone() for _ in range(3)]
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
            close_bf16(outs[0], a.float() @ b.float().t())
        M, N, K = 200, 264, 2048
        a, b, bias, res = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
        base = a.float() @ b.float().t()
        ad, bd, biasd, resd = a.to(D), b.to(D), bias.to(D), res.to(D)
        close_bf16(ops.gemm(ad, bd, bias=biasd), base + bias.float())
        close_bf16(ops.gemm(ad, bd, bias=biasd, gelu=True), F.gelu(base + bias.float()))
        close_bf16(ops.gemm(ad, bd, bias=biasd, residual=resd), base + bias.float() + res.float())
        close_f32(ops.gemm(ad, bd, bias=biasd, out_f32=True, alpha=0.5), 0.5 * base + bias.float())
        bm = rnd(M, seed=7)
        close_bf16(ops.gemm(ad, bd, bias=bm.to(D), bias_m=True), base + bm.float()[:, None])
    finally:
        ops.set_option("gemm_splitk", 0)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_kmajor_operands(ops, tile):
    """K-major operand forms (ds_read_b64_tr_b16 fragments): C = A B with B (K, N) and C = A^T B with A (K, M), B (K, N) --
    the dX / dW products of the training path -- on random (transpose-detecting) data: row / column tails of both tile
    sizes, K that is no multiple of the K tile (and, with both operands K-major, of nothing), K slices, bit-repeatable."""
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)
    ops.set_option("gemm_tile", tile)
    try:
        for sk in (-1, 0, 3):
            ops.set_option("gemm_splitk", sk)
            for (M, N, K) in [(128, 128, 64), (200, 72, 136), (8, 520, 1160), (264, 8, 72), (2049, 768, 768), (256, 1000, 2048)]:
                a, bk = rnd(M, K, seed=1), rnd(K, N, seed=2)
                got = ops.gemm_kmajor(a.to(D), bk.to(D), a_kmajor=False)
                close_bf16(got, a.float() @ bk.float())
                assert torch.equal(got, ops.gemm_kmajor(a.to(D), bk.to(D), a_kmajor=False))
            for (M, N, K) in [(128, 128, 64), (72, 200, 131), (520, 8, 1157), (8, 264, 77), (768, 2304, 2049), (1000, 256, 4100),
                              (768, 768, 16392)]:
                ak, bk = rnd(K, M, seed=3), rnd(K, N, seed=4)
                got = ops.gemm_kmajor(ak.to(D), bk.to(D), a_kmajor=True)
                close_bf16(got, ak.float().t() @ bk.float())
                assert torch.equal(got, ops.gemm_kmajor(ak.to(D), bk.to(D), a_kmajor=True))
        ak, bk = rnd(300, 264, seed=5), rnd(300, 200, seed=6)
        close_f32(ops.gemm_kmajor(ak.to(D), bk.to(D), a_kmajor=True, alpha=0.5, out_f32=True), 0.5 * ak.float().t() @ bk.float())
        with pytest.raises(RuntimeError):
            ops.gemm_kmajor(rnd(64, 64, seed=7).to(D), rnd(64, 12, seed=8).to(D), a_kmajor=False)   # N % 8 != 0
    finally:
        ops.set_option("gemm_tile", 0)
        ops.set_option("gemm_splitk", 0)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


def test_gemm_small_tile_flat_and_mubuf_pieces_agree(ops):
    """Round 6: the small-tile kernel's LDS-DMA pieces leave as buffer_load ... lds (option gemm_mubuf, default 1; the chunk that does
    not exist -- K tail, columns past M / N of a K-major operand -- takes an out-of-range lane offset and lands as zeros) or as the
    FLAT-encoded global_load_lds of rounds 1-5 (0): the same bits on ragged shapes of every operand form, batched and K-tile-major."""
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)

    def both(fn):
        out = []
        for mode in (0, 1):
            ops.set_option("gemm_mubuf", mode)
            out.append(fn().clone())
        assert torch.equal(out[0], out[1])

    try:
        for tile in (64, 128):
            ops.set_option("gemm_tile", tile)
            for (M, N, K) in [(300, 200, 136), (77, 520, 72), (2049, 768, 776)]:
                a, b, bias = rnd(M, K, seed=1).to(D), rnd(N, K, seed=2).to(D), rnd(N, seed=3).to(D)
                both(lambda: ops.gemm(a, b, bias=bias))
            a3, b3 = rnd(6, 100, 72, seed=10).to(D), rnd(6, 90, 72, seed=11).to(D)
            both(lambda: ops.gemm(a3, b3, out_f32=True))
            for (M, N, K) in [(200, 72, 136), (264, 8, 72), (256, 1000, 2048)]:
                a, bk = rnd(M, K, seed=4).to(D), rnd(K, N, seed=5).to(D)
                both(lambda: ops.gemm_kmajor(a, bk, a_kmajor=False))
            for (M, N, K) in [(72, 200, 131), (520, 8, 1157), (768, 2304, 2049)]:
                ak, bk = rnd(K, M, seed=6).to(D), rnd(K, N, seed=7).to(D)
                both(lambda: ops.gemm_kmajor(ak, bk, a_kmajor=True))
        ops.set_option("gemm_tile", 0)
        a, w = rnd(77, 128, seed=8).to(D), rnd(200, 128, seed=9).to(D)
        wt = ops.pack_ktile_major(w)
        both(lambda: ops.gemm(a, wt, b_ktile=True))
    finally:
        ops.set_option("gemm_tile", 0)
        ops.set_option("gemm_mubuf", 1)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


def test_gemm_ktile_major_weights(ops):
    """B handed over K-tile-major ([K/64][N][64], ops.pack_ktile_major): same products as the row-major weight, with and
    without split-K, tails in M and N."""
    scratch = torch.empty(16 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)
    try:
        for (M, N, K) in [(256, 1024, 2048), (77, 200, 128), (300, 520, 1024)]:
            a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
            wt = ops.pack_ktile_major(w.to(D))
            ref = a.float() @ w.float().t() + bias.float()
            for sk in (-1, 0, 4):
                ops.set_option("gemm_splitk", sk)
                got = ops.gemm(a.to(D), wt, bias=bias.to(D), b_ktile=True)
                close_bf16(got, ref)
                assert torch.equal(got, ops.gemm(a.to(D), wt, bias=bias.to(D), b_ktile=True))
    finally:
        ops.set_option("gemm_splitk", 0)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


@pytest.mark.parametrize("variant", [20, 21, 22, 24, 26])
def test_gemm_big_tile_variants(ops, variant):
    """gemm_bt.hip (persistent 256 x 256 / 256 x 192 big-tile kernel, asm K loop; 22 = the 256 x 128 three-stage ring form;
    24 / 26 = the deep forms of the 192- and 256-wide tiles: three stages for B, two for A)
    forced on shapes with row / column tails, several tiles per workgroup (K loops of 4 and 5 K tiles chained from tile to
    tile: the ring form then enters at each of its three stages), every fused epilogue, batches; each product is launched 3
    times and must repeat bit for bit (a race between the LDS-DMA ring and the fragment reads would not).  Shapes the
    kernel cannot take (K % 64 != 0, K < 128) fall through to the small-tile kernel."""
    ops.set_option("gemm_big", variant)
    try:
        for (M, N, K) in [(256, 256, 128), (300, 200, 136), (77, 520, 192), (1000, 768, 1024), (2049, 768, 768),
                          (5000, 1536, 256), (9000, 2304, 256), (4100, 3000, 320), (9000, 1100, 128),
                          (9000, 2304, 448), (9000, 2304, 512), (9000, 2304, 576)]:   # (K tiles mod 6 = 1, 2, 3: every entry of the deep forms' unrolled loop)
            a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
            ad, bd = a.to(D), b.to(D)
            outs = [ops.gemm(ad, bd).clone() for _ in range(3)]
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
            close_bf16(outs[0], a.float() @ b.float().t())
        M, N, K = 515, 264, 192
        a, b, bias, res = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
        base = a.float() @ b.float().t()
        ad, bd, biasd, resd = a.to(D), b.to(D), bias.to(D), res.to(D)
        close_bf16(ops.gemm(ad, bd, bias=biasd), base + bias.float())
        close_bf16(ops.gemm(ad, bd, bias=biasd, gelu=True), F.gelu(base + bias.float()))
        close_bf16(ops.gemm(ad, bd, bias=biasd, residual=resd), base + bias.float() + res.float())
        close_bf16(ops.gemm(ad, bd, residual=resd), base + res.float())
        close_f32(ops.gemm(ad, bd, bias=biasd, out_f32=True, alpha=0.5), 0.5 * base + bias.float())
        close_f32(ops.gemm(ad, bd, out_f32=True), base)
        a3, b3 = rnd(6, 300, 128, seed=10), rnd(6, 96, 128, seed=11)
        close_f32(ops.gemm(a3.to(D), b3.to(D), out_f32=True), torch.einsum("zmk,znk->zmn", a3.float(), b3.float()))
    finally:
        ops.set_option("gemm_big", 0)


def test_gelu_fwd_over_all_bf16_inputs(ops):
    """u2tok_gelu_fwd (= the GEMM epilogues' gelu_fast, csrc/common.h: the logistic-polynomial form of round 6) on EVERY finite input of the
    element type with |x| <= 30 against float64 x Phi(x): the output is the correctly rounded value or its neighbour, the neighbour for at
    most 0.3 % of the inputs with |y| >= 0.01 (CPU restatement: 0.09 % in bf16; the hardware's exp / rcp add their ulp), and exact x / -0
    beyond saturation."""
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(bf)
    x = x[torch.isfinite(x.float()) & (x.float().abs() <= 30)]
    pad = (-x.numel()) % 8
    x = torch.cat([x, x[:pad]])
    y = ops.gelu_fwd(x.to(D)).cpu()
    xd = x.double()
    ref = xd * 0.5 * (1 + torch.erf(xd / math.sqrt(2.0)))
    rb = ref.to(bf)
    big = ref.abs() >= 1e-2
    ulp = torch.exp2(torch.floor(torch.log2(rb[big].double().abs())) - (7 if bf == torch.bfloat16 else 10))
    assert ((y[big].double() - rb[big].double()).abs() <= ulp).all()
    frac = (y[big] != rb[big]).double().mean().item()
    assert frac <= (3e-3 if bf == torch.bfloat16 else 3e-2), frac
    assert ((y.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-5).all()
    far = torch.tensor([40.0, 1e4, -40.0, -1e4, 0.0, 0.0, 0.0, 0.0]).to(bf)
    out = ops.gelu_fwd(far.to(D)).cpu().float()
    assert out[:2].tolist() == [40.0, float(far[1])] and (out[2:] == 0).all()


@pytest.mark.parametrize("M,N,K,grid", [(1024, 768, 768, 4), (512, 1536, 1536, 4), (2048, 384, 768, 3), (768, 1152, 2304, 2)])
def test_gemm_big_tile_drain_form(ops, M, N, K, grid):
    """Round 6, gemm_bt_drain_kernel (variant 27): tile i's epilogue under tile i + 1's K loop -- CONVERT packs the accumulators into 96
    registers at the start of the next tile's asm statement, the K loop's first twelve iterations store them (and, in the GELU form,
    evaluate the GELU on the held values from the MFMA slots).  A small persistent grid gives every workgroup 2-5 tiles at test size
    (first tile: plain deep loop; middle tiles: CONVERT + drain loop, K = 768 exits behind the twelfth drain body, K = 1536 / 2304 run on
    into the plain bodies; last tile: the exposed epilogue in HIP).
    Plain / bias / alpha forms: the arithmetic of the deep 256 x 192 form (variant 24) -- BIT-identical outputs.
    GELU form: the pre-activation is rounded to the element type first (the reference's own rounding point: nn.Linear hands nn.GELU a
    bf16 tensor) -- bit-identical to `gelu_fwd(gemm(..., bias))`, i.e. the library's GELU kernel on the stored pre-activation."""
    a, b, bias = rnd(M, K, seed=51), rnd(N, K, scale=0.05, seed=52), rnd(N, seed=53)
    ad, bd, biasd = a.to(D), b.to(D), bias.to(D)
    base = a.float() @ b.float().t()
    ops.set_option("gemm_big_grid", grid)
    try:
        got, ref = {}, {}
        for variant, res in ((24, ref), (27, got)):
            ops.set_option("gemm_big", variant)
            res["plain"] = [ops.gemm(ad, bd).clone() for _ in range(3)]
            res["bias"] = [ops.gemm(ad, bd, bias=biasd).clone() for _ in range(3)]
            res["alpha"] = [ops.gemm(ad, bd, alpha=0.37).clone()]
            res["alpha_bias"] = [ops.gemm(ad, bd, bias=biasd, alpha=0.37).clone()]
            if variant == 27:
                res["gelu"] = [ops.gemm(ad, bd, bias=biasd, gelu=True).clone() for _ in range(3)]
        for k, outs in got.items():
            assert all(torch.equal(outs[0], o) for o in outs[1:]), k            # repeatable
        for k in ("plain", "bias", "alpha", "alpha_bias"):
            assert torch.equal(got[k][0], ref[k][0]), (k, (got[k][0].float() - ref[k][0].float()).abs().max().item())
        close_bf16(got["plain"][0], base)
        close_bf16(got["alpha_bias"][0], 0.37 * base + bias.float())
        assert torch.equal(got["gelu"][0], ops.gelu_fwd(ref["bias"][0]))
        close_bf16(got["gelu"][0], F.gelu(base + bias.float()), rounds=4)
    finally:
        ops.set_option("gemm_big", 0)
        ops.set_option("gemm_big_grid", 256)


def test_gemm_drain_form_by_heuristic_with_cls_rows(ops):
    """The heuristic's own choice at the ViT's row count shape (rows = k 256 + 8 cls rows, K = 768): with gemm_big_drain on, the many-row
    part runs the drain form and the 8 tail rows ride in the same launch; outputs equal the drain-less launch bit for bit (plain and
    bias epilogues), the GELU product equals the library's GELU kernel on the stored pre-activation in its many rows."""
    M, N, K = 8 * 256 + 8, 1536, 768
    a, b, bias = rnd(M, K, seed=61), rnd(N, K, scale=0.05, seed=62), rnd(N, seed=63)
    ad, bd, biasd = a.to(D), b.to(D), bias.to(D)
    ops.set_option("gemm_big_grid", 16)        # 8 x 8 tiles on 16 workgroups: four tiles each
    try:
        res = {}
        for drain in (0, 1):
            ops.set_option("gemm_big_drain", drain)
            res[drain] = (ops.gemm(ad, bd).clone(), ops.gemm(ad, bd, bias=biasd).clone(), ops.gemm(ad, bd, bias=biasd, gelu=True).clone())
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        close_bf16(res[1][1], a.float() @ b.float().t() + bias.float())
        assert torch.equal(res[1][2][:M - 8], ops.gelu_fwd(res[0][1])[:M - 8])
        close_bf16(res[1][2], F.gelu(a.float() @ b.float().t() + bias.float()), rounds=4)
        assert torch.equal(res[1][2][M - 8:], res[0][2][M - 8:])      # the cls rows: the few-rows arithmetic either way
    finally:
        ops.set_option("gemm_big_drain", 1)
        ops.set_option("gemm_big_grid", 256)


@pytest.mark.parametrize("variant,slices", [(21, 2), (21, 8), (20, 5), (21, 3), (20, 2), (22, 8), (22, 3)])
def test_gemm_big_tile_k_slices(ops, variant, slices):
    """The big-tile kernel with its K range cut into slices (fp32 partial sums in the stream's scratch, epilogue applied by the
    reduce kernel in a fixed order): uneven last slices, several tiles per workgroup with slices of different lengths chained in
    one K loop, row / column tails, every epilogue; repeatable bit for bit."""
    scratch = torch.empty(96 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)
    ops.set_option("gemm_big", variant)
    ops.set_option("gemm_big_splitk", slices)
    try:
        for (M, N, K) in [(256, 4096, 4096), (200, 520, 1024), (1024, 1032, 2176), (2049, 384, 704), (129, 136, 640)]:
            a, b, bias, res = rnd(M, K, seed=41), rnd(N, K, seed=42), rnd(N, seed=43), rnd(M, N, seed=44)
            ad, bd, biasd, resd = a.to(D), b.to(D), bias.to(D), res.to(D)
            base = a.float() @ b.float().t()
            outs = [ops.gemm(ad, bd, bias=biasd, residual=resd).clone() for _ in range(3)]
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
            close_bf16(outs[0], base + bias.float() + res.float())
            close_bf16(ops.gemm(ad, bd), base)
            close_f32(ops.gemm(ad, bd, bias=biasd, out_f32=True, alpha=0.5), 0.5 * base + bias.float())
    finally:
        ops.set_option("gemm_big", 0)
        ops.set_option("gemm_big_splitk", 0)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(256, 4096, 4096), (200, 2048, 2048), (129, 4096, 1024), (256, 2048, 8192), (65, 3072, 1536)])
def test_gemm_skinny_unsplit_form(ops, M, N, K):
    """Round 6, gemm_skinny.hip (option gemm_skinny, default 2): 64 < M <= 256 rows against a 2048 .. 4096-column weight run 64 x 64 tiles
    over the WHOLE K in 128-wide K tiles (256-byte LDS rows swizzled by row & 15, three stages, one counted wait per tile) -- no K
    slices, no partial sums, no reduce launch.  Every epilogue; rows past M; bit-repeatable; with the option off the round-5 path (64 x 64
    tiles x K slices + reduce, needs the scratch) gives the same values up to summation order (that the new kernel RUNS is what the bench line's
    kernel table and tools/skinny_probe.py show: 21 gemm_skinny64_kernel launches per volume, 21 reduce launches fewer)."""
    a, b, bias, res = rnd(M, K, seed=71), rnd(N, K, scale=0.05, seed=72), rnd(N, seed=73), rnd(M, N, seed=74)
    ad, bd, biasd, resd = a.to(D), b.to(D), bias.to(D), res.to(D)
    base = a.float() @ b.float().t()
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=D)
    ops.set_gemm_scratch(scratch)

    try:
        ops.set_option("gemm_skinny", 1)           # the FLAT-encoded pieces (A/B form): the same bits as the default's buffer_load ... lds
        flat = ops.gemm(ad, bd, bias=biasd).clone()
        ops.set_option("gemm_skinny", 2)
        outs = [ops.gemm(ad, bd, bias=biasd).clone() for _ in range(4)]
        assert torch.equal(flat, outs[0])
        ops.set_option("gemm_skinny", 0)
        old = [ops.gemm(ad, bd, bias=biasd).clone()]
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        close_bf16(outs[0], base + bias.float())
        close_bf16(old[0], base + bias.float())
        assert (old[0].float() - outs[0].float()).abs().max() <= 2 * ULP * (base + bias.float()).abs().max()
        ops.set_option("gemm_skinny", 2)
        close_bf16(ops.gemm(ad, bd), base)
        close_bf16(ops.gemm(ad, bd, bias=biasd, residual=resd), base + bias.float() + res.float())
        close_bf16(ops.gemm(ad, bd, bias=biasd, gelu=True), F.gelu(base + bias.float()), rounds=4)
        close_f32(ops.gemm(ad, bd, bias=biasd, out_f32=True, alpha=0.5), 0.5 * base + bias.float())
        ops.set_gemm_scratch(None)                 # the unsplit form needs no scratch
        assert torch.equal(ops.gemm(ad, bd, bias=biasd), outs[0])
    finally:
        ops.set_option("gemm_skinny", 2)
        ops.set_option("profile", 0)
        ops.set_gemm_scratch(None)
        torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(8, 2304, 768), (8, 768, 3072), (1, 40, 64), (16, 4096, 4096), (5, 24, 96), (8, 768, 768)])
def test_gemm_few_rows(ops, M, N, K):
    """M <= 16 (the ViT's 8 cls rows behind the big-tile launches): gemm_rows16_kernel -- waves split K, partial tiles are
    added through LDS in a fixed order; every epilogue; three launches agree bit for bit."""
    a, b, bias, res = rnd(M, K, seed=31), rnd(N, K, seed=32), rnd(N, seed=33), rnd(M, N, seed=34)
    base = a.float() @ b.float().t()
    got = [ops.gemm(a.to(D), b.to(D), bias=bias.to(D), residual=res.to(D)) for _ in range(3)]
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])
    close_bf16(got[0], base + bias.float() + res.float())
    close_bf16(ops.gemm(a.to(D), b.to(D)), base)
    close_bf16(ops.gemm(a.to(D), b.to(D), bias=bias.to(D), gelu=True), F.gelu(base + bias.float()), rounds=3)
    close_f32(ops.gemm(a.to(D), b.to(D), bias=bias.to(D), out_f32=True, alpha=0.5), 0.5 * base + bias.float())


def test_gemm_heuristic_split_rows(ops):
    """M = 8 * 2049 (the ViT's row count): the launcher sends 16384 rows to the big-tile kernel and the 8 leftover
    rows to the small-tile kernel; the seam must be invisible."""
    M, N, K = 16392, 768, 768
    a, b, bias, res = rnd(M, K, seed=21), rnd(N, K, seed=22), rnd(N, seed=23), rnd(M, N, seed=24)
    got = ops.gemm(a.to(D), b.to(D), bias=bias.to(D), residual=res.to(D))
    close_bf16(got, a.float() @ b.float().t() + bias.float() + res.float())
