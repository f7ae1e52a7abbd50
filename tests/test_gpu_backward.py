"""GPU: the training path (u2tokenizer_amd/autograd.py -- HIP forward AND backward behind torch.autograd.Function) against
torch.autograd over the CPU oracle.

Every op-level test differentiates a plain fp32 torch expression of the same bf16 inputs; the module-level tests
differentiate oracle/u2_oracle.py (the restatement of the reference's modules) with name-seeded parameters and compare
the gradient of EVERY parameter.  Tolerance: a gradient may be no further from the fp32 reference gradient than 1.5x the
distance of the reference's own bf16 run (same oracle, bf16 tensors, CPU autograd) plus 2 % of that tensor's gradient
scale; gradients that are tiny relative to the largest one of the module are compared against that largest scale.
"""
import math
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

from helpers import module_sd
from oracle import u2_oracle as O
from u2tokenizer_amd import synth

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
D = "cuda"


@pytest.fixture(autouse=True)
def _grad_on():
    assert torch.cuda.is_available()
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    yield
    torch.set_grad_enabled(prev)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed * 7919 + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(bf)


def rel(a, b, floor=0.0):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + floor)).item()


def leaf(t, dev=None):
    t = t.detach().clone()
    if dev:
        t = t.to(dev)
    return t.requires_grad_(True)


def check_grads(got: dict, ref32: dict, ref16: dict = None, what=""):
    """got / ref32 / ref16: name -> gradient tensor."""
    top = max(v.double().pow(2).mean().sqrt().item() for v in ref32.values())
    worst = {}
    for k, g32 in ref32.items():
        assert k in got and got[k] is not None, f"{what}: no gradient for {k}"
        assert torch.isfinite(got[k].float()).all(), (what, k)
        floor = 2e-3 * top
        e = rel(got[k].float(), g32, floor)
        bar = 2e-2 + (1.5 * rel(ref16[k].float(), g32, floor) if ref16 is not None else 1e-2)
        worst[k] = (e, bar)
    bad = {k: v for k, v in worst.items() if v[0] > v[1]}
    assert not bad, (what, bad)
    # all gradients as one vector: the direction of the step must agree with the fp32 reference as well as the
    # reference's own bf16 run does (parameters whose exact gradient is zero -- e.g. key biases -- are pure rounding noise
    # in any bf16 run and only matter here through their tiny norm)
    keys = sorted(ref32)

    def cos(a, b):
        va = torch.cat([a[k].detach().double().cpu().flatten() for k in keys])
        vb = torch.cat([b[k].detach().double().cpu().flatten() for k in keys])
        return (va @ vb / (va.norm() * vb.norm()).clamp_min(1e-30)).item()

    c_hip = cos(got, ref32)
    c_ref = cos(ref16, ref32) if ref16 is not None else 0.999
    assert c_hip >= c_ref - 0.02, (what, c_hip, c_ref)
    worst["cosine_vs_fp32"] = (c_hip, c_ref)
    return worst


# ------------------------------------------------------------------------------------------------ op level
@pytest.mark.parametrize("gelu,res,bias", [(False, False, True), (True, False, True), (False, True, True), (False, False, False)])
def test_linear_fn(gelu, res, bias):
    from u2tokenizer_amd import autograd as AG
    M, N, K = 300, 264, 200
    x, w, b, r, g = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    xr, wr, br, rr = leaf(x.float()), leaf(w.float()), leaf(b.float()), leaf(r.float())
    y = F.linear(xr, wr, br if bias else None)
    y = F.gelu(y) if gelu else y
    y = y + rr if res else y
    (y * g.float()).sum().backward()
    xd, wd, bd, rd = leaf(x, D), leaf(w, D), leaf(b, D), leaf(r, D)
    yd = AG.linear(xd, wd, bd if bias else None, rd if res else None, gelu)
    assert rel(yd.float(), y) < 1e-2
    (yd.float() * g.to(D).float()).sum().backward()
    got = {"x": xd.grad, "w": wd.grad}
    ref = {"x": xr.grad, "w": wr.grad}
    if bias:
        got["b"], ref["b"] = bd.grad, br.grad
    if res:
        got["r"], ref["r"] = rd.grad, rr.grad
    check_grads(got, ref, what=f"linear gelu={gelu} res={res}")


@pytest.mark.parametrize("C,res", [(768, False), (768, True), (4096, True), (512, False)])
def test_layernorm_fn(C, res):
    from u2tokenizer_amd import autograd as AG
    R = 150
    x, r, w, b, g = rnd(R, C, seed=1), rnd(R, C, seed=2), (1 + 0.1 * rnd(C, seed=3).float()).to(bf), rnd(C, seed=4), rnd(R, C, seed=5)
    xr, rr, wr, br = leaf(x.float()), leaf(r.float()), leaf(w.float()), leaf(b.float())
    y = F.layer_norm(xr + rr if res else xr, (C,), wr, br)
    (y * g.float()).sum().backward()
    xd, rd, wd, bd = leaf(x, D), leaf(r, D), leaf(w, D), leaf(b, D)
    yd = AG.layernorm(xd, wd, bd, rd if res else None)
    (yd.float() * g.to(D).float()).sum().backward()
    got, ref = {"x": xd.grad, "w": wd.grad, "b": bd.grad}, {"x": xr.grad, "w": wr.grad, "b": br.grad}
    if res:
        got["r"], ref["r"] = rd.grad, rr.grad
    check_grads(got, ref, what=f"layernorm C={C}")


def _attn_ref(q, k, v, H, scale, bias_tbl=None, L=512):
    nb, Sq, E = q.shape
    d = E // H

    def heads(t):
        return t.view(nb, -1, H, d).permute(0, 2, 1, 3)

    s = heads(q) @ heads(k).transpose(-1, -2) * scale
    if bias_tbl is not None:
        pos = torch.arange(Sq)
        s = s + bias_tbl[pos[None, :] - pos[:, None] + L - 1].permute(2, 0, 1).unsqueeze(0)
    return (s.softmax(-1) @ heads(v)).permute(0, 2, 1, 3).reshape(nb, Sq, E)


@pytest.mark.parametrize("nb,S,H,d,bias", [(3, 40, 4, 64, True), (2, 37, 8, 32, True), (2, 24, 2, 128, False)])
def test_self_attention_fn(nb, S, H, d, bias):
    from u2tokenizer_amd import autograd as AG
    E, L = H * d, 512
    qkv, tbl, g = rnd(nb, S, 3 * E, seed=1), rnd(2 * L - 1, H, scale=0.5, seed=2), rnd(nb, S, E, seed=3)
    qr, tr = leaf(qkv.float()), leaf(tbl.float())
    y = _attn_ref(qr[..., :E], qr[..., E:2 * E], qr[..., 2 * E:], H, d ** -0.5, tr if bias else None, L)
    (y * g.float()).sum().backward()
    qd, td = leaf(qkv, D), leaf(tbl, D)
    yd = AG.SelfAttnFn.apply(qd, td if bias else None, H, d ** -0.5, L, False)
    assert rel(yd.float(), y) < 1.5e-2
    (yd.float() * g.to(D).float()).sum().backward()
    got, ref = {"qkv": qd.grad}, {"qkv": qr.grad}
    if bias:
        got["table"], ref["table"] = td.grad, tr.grad
    check_grads(got, ref, what="self attention")


def test_flash_self_attention_fn():
    """ViT form: flash forward (last row = the kernel's extra row), default backward (fused kernels since round 2)."""
    from u2tokenizer_amd import autograd as AG
    nb, S, H, d = 2, 129, 3, 64
    E = H * d
    qkv, g = rnd(nb, S, 3 * E, seed=7), rnd(nb, S, E, seed=8)
    qr = leaf(qkv.float())
    y = _attn_ref(qr[..., :E], qr[..., E:2 * E], qr[..., 2 * E:], H, 0.125)
    (y * g.float()).sum().backward()
    qd = leaf(qkv, D)
    yd = AG.SelfAttnFn.apply(qd, None, H, 0.125, 0, True)
    assert rel(yd.float(), y) < 1.5e-2
    (yd.float() * g.to(D).float()).sum().backward()
    check_grads({"qkv": qd.grad}, {"qkv": qr.grad}, what="flash self attention")


@pytest.mark.parametrize("nb,S,H,gain", [(2, 129, 3, 1.0), (1, 64, 2, 1.0), (1, 1, 1, 1.0), (2, 200, 4, 2.0), (1, 2049, 2, 1.0),
                                         (3, 127, 1, 1.0), (1, 448, 12, 0.5)])
def test_flash_attention_backward_kernels(nb, S, H, gain):
    """u2tok_flash_attention_d64_bwd (attn_bwd.hip) per component against fp32 torch.autograd, against the unfused chain
    (flash = 1: probabilities rebuilt in HBM), and bit-repeatable.  S covers ragged / single-tile / multi-workgroup sizes and
    the ViT's own 2049."""
    from u2tokenizer_amd import autograd as AG
    E = H * 64
    qkv, g = rnd(nb, S, 3 * E, scale=gain, seed=11), rnd(nb, S, E, seed=12)
    qr = leaf(qkv.float())
    y = _attn_ref(qr[..., :E], qr[..., E:2 * E], qr[..., 2 * E:], H, 0.125)
    (y * g.float()).sum().backward()
    grads = {}
    for mode in (2, 1, 2):
        qd = leaf(qkv, D)
        yd = AG.SelfAttnFn.apply(qd, None, H, 0.125, 0, mode)
        (yd.float() * g.to(D).float()).sum().backward()
        grads.setdefault(mode, []).append(qd.grad)
    assert torch.equal(grads[2][0], grads[2][1]), "fused attention backward is not bit-repeatable"
    assert torch.isfinite(grads[2][0].float()).all()
    top = qr.grad.double().pow(2).mean().sqrt().item()
    report = {}
    for i, name in enumerate(("dq", "dk", "dv")):
        ref = qr.grad[..., i * E:(i + 1) * E]
        fused, unfused = grads[2][0][..., i * E:(i + 1) * E].float(), grads[1][0][..., i * E:(i + 1) * E].float()
        e_f, e_u = rel(fused, ref, 2e-3 * top), rel(unfused, ref, 2e-3 * top)
        report[name] = (e_f, e_u)
        # bf16 gradients: the unfused chain rounds P and dS to bf16 exactly as the fused kernels do
        assert e_f <= max(1.5 * e_u, 0.0) + 1e-2, (name, report)
    check_grads({"qkv": grads[2][0]}, {"qkv": qr.grad}, what="fused flash attention backward")
    # the forward's row statistics: log2-sum-exp2 of the scaled scores, and the backward with / without them
    from u2tokenizer_amd import ops
    out, lse = ops.flash_attention_d64(qkv.to(D), H, 0.125, extra_last=S > 1, return_lse=True)
    x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = torch.logsumexp(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1) / math.log(2.0)       # (nb, H, S)
    assert (lse[:, :S].cpu().view(nb, H, S) - want).abs().max() < 2e-3
    with_lse = ops.flash_attention_d64_bwd(qkv.to(D), out, g.to(D), H, 0.125, lse=lse)
    without = ops.flash_attention_d64_bwd(qkv.to(D), out, g.to(D), H, 0.125)
    assert rel(with_lse.float(), without.float()) < 2e-3


def test_cross_and_plain_attention_fns():
    from u2tokenizer_amd import autograd as AG
    nb, Sq, Skv, H, d = 2, 24, 50, 4, 64
    E = H * d
    q, kv, g = rnd(nb, Sq, E, seed=1), rnd(nb, Skv, 2 * E, seed=2), rnd(nb, Sq, E, seed=3)
    qr, kvr = leaf(q.float()), leaf(kv.float())
    y = _attn_ref(qr, kvr[..., :E], kvr[..., E:], H, d ** -0.5)
    (y * g.float()).sum().backward()
    qd, kvd = leaf(q, D), leaf(kv, D)
    yd = AG.CrossAttnFn.apply(qd, kvd, H, d ** -0.5)
    (yd.float() * g.to(D).float()).sum().backward()
    check_grads({"q": qd.grad, "kv": kvd.grad}, {"q": qr.grad, "kv": kvr.grad}, what="cross attention")
    k, v = kv[..., :E].contiguous(), kv[..., E:].contiguous()
    q2, k2, v2 = leaf(q, D), leaf(k, D), leaf(v, D)
    y2 = AG.AttnFn.apply(q2, k2, v2, H, d ** -0.5)
    (y2.float() * g.to(D).float()).sum().backward()
    check_grads({"q": q2.grad, "k": k2.grad, "v": v2.grad},
                {"q": qr.grad, "k": kvr.grad[..., :E], "v": kvr.grad[..., E:]}, what="plain attention")


def test_diffts_fn():
    from u2tokenizer_amd import autograd as AG
    B, TN, E, k = 2, 96, 256, 40
    x, w, b, g = rnd(B, TN, E, seed=1), rnd(k, E, scale=4 * E ** -0.5, seed=2), rnd(k, scale=0.1, seed=3), rnd(B, k, E, seed=4)
    xr, wr, br = leaf(x.float()), leaf(w.float()), leaf(b.float())
    sc = F.linear(xr, wr, br)                                   # (B, TN, k)
    y = torch.softmax(sc / 0.7, dim=1).transpose(1, 2) @ xr     # svr.py:105-117
    (y * g.float()).sum().backward()
    xd, wd, bd = leaf(x, D), leaf(w, D), leaf(b, D)
    yd = AG.DiffTSFn.apply(xd, wd, bd, 0.7)
    assert rel(yd.float(), y) < 3e-2, rel(yd.float(), y)
    (yd.float() * g.to(D).float()).sum().backward()
    check_grads({"x": xd.grad, "w": wd.grad}, {"x": xr.grad, "w": wr.grad}, what="DiffTS")
    # the softmax runs over TOKENS, so a per-head constant cancels: the exact gradient of the score bias is zero and any
    # bf16 run produces rounding noise there (the fp32 reference itself: ~1e-9) -- it only has to be small
    assert br.grad.abs().max() < 1e-5 * wr.grad.abs().max()
    assert bd.grad.float().abs().max().item() < 0.05 * wr.grad.abs().max().item()


def test_multiscale_pool_fn():
    from u2tokenizer_amd import autograd as AG
    for gated in (False, True):
        xs, gw, gb = rnd(2, 30, 256, seed=5), rnd(1, 256, scale=0.3, seed=6), rnd(1, seed=7)
        go = rnd(2, 30 + 15 + 7, 256, seed=8)
        xr, gwr, gbr = leaf(xs.float()), leaf(gw.float()), leaf(gb.float())
        sd = {"p.gate_fc.weight": gwr, "p.gate_fc.bias": gbr}
        y = O.multi_scale_pool(sd, "p" if gated else None, xr)
        (y * go.float()).sum().backward()
        xd, gwd, gbd = leaf(xs, D), leaf(gw, D), leaf(gb, D)
        yd = AG.MultiScalePoolFn.apply(xd, gwd if gated else None, gbd if gated else None)
        (yd.float() * go.to(D).float()).sum().backward()
        got, ref = {"x": xd.grad}, {"x": xr.grad}
        if gated:
            got.update(w=gwd.grad, b=gbd.grad)
            ref.update(w=gwr.grad, b=gbr.grad)
        check_grads(got, ref, what=f"multi-scale pool gated={gated}")


def test_hard_topk_fn_scatters_its_gradient():
    """hard top-k: the gather's gradient lands on the selected rows only"""
    from u2tokenizer_amd import autograd as AG
    x, w, b = rnd(2, 64, 256, seed=9), rnd(1, 256, scale=0.1, seed=10), rnd(1, seed=11)
    xd = leaf(x, D)
    sel, idx = AG.HardTopKFn.apply(xd, w.to(D), b.to(D), 16)
    go = rnd(2, 16, 256, seed=12).to(D)
    (sel.float() * go.float()).sum().backward()
    want = torch.zeros(2, 64, 256)
    want[torch.arange(2)[:, None], idx.cpu()] = go.float().cpu()
    assert torch.equal(xd.grad.float().cpu(), want)


# ------------------------------------------------------------------------------------------------ module level
def _oracle_grads(sd32, dt, fn):
    """gradients of sum(out * G) w.r.t. every floating tensor of the state dict, oracle on the host in dtype dt."""
    sd = {k: leaf(v.to(dt)) for k, v in sd32.items()}
    out, G = fn(sd)
    (out.float() * G).sum().backward()
    return out.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}


TOK_TRAIN_CASES = {
    "mu2_2l": dict(E=512, layers=2, B=2, T=4, N=32, Lt=40, Q=32, top_k=64, ms=True, attn="rma", diffts=True, dmtp=True,
                   lively=False, seed=81),
    "mu2_1l_live": dict(E=512, layers=1, B=2, T=4, N=32, Lt=24, Q=16, top_k=64, ms=True, attn="rma", diffts=True, dmtp=True,
                        lively=True, seed=82),
    "hard_rope_1l": dict(E=512, layers=1, B=1, T=3, N=24, Lt=16, Q=16, top_k=32, ms=True, attn="rope", diffts=False,
                         dmtp=False, lively=False, seed=83),
    # the nn.MultiheadAttention ("linvt") ablation, a shipped stage-1 recipe (script/amos_mm_stage1/amos_mm_linvt_stage1.sh:46):
    # attention across batch entries -- B = 2 makes that visible
    "linvt_2l_live": dict(E=512, layers=2, B=2, T=4, N=24, Lt=20, Q=16, top_k=32, ms=True, attn="linvt", diffts=True,
                          dmtp=True, lively=True, seed=84),
}


@pytest.mark.parametrize("name", list(TOK_TRAIN_CASES))
def test_tokenizer_gradients_vs_oracle(name):
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    c = TOK_TRAIN_CASES[name]
    E = c["E"]
    tok = u2Tokenizer(E, 8, c["layers"], c["top_k"], c["ms"], c["Q"], E, c["attn"], c["diffts"], c["dmtp"])
    sd32 = module_sd(tok, "u2tokenizer.", c["seed"])
    if c["lively"]:
        for n, t in sd32.items():
            synth.lively_(n, t, qk_gain=2.0)
    tok.load_state_dict({k[len("u2tokenizer."):]: v for k, v in sd32.items()})
    v = synth.synth_tensor("v_token", (c["B"], c["T"], c["N"], E), c["seed"]).to(bf)
    t = (0.25 * synth.synth_tensor("t_token", (c["B"], c["Lt"], E), c["seed"])).to(bf)
    G = synth.synth_tensor("grad_out", (c["B"], c["Q"], E), c["seed"])
    oc = O.PathConfig(hidden_size=E, u2t_num_layers=c["layers"], u2t_top_k=c["top_k"], use_multi_scale=c["ms"],
                      num_3d_query_token=c["Q"], attn_type=c["attn"], enable_diffts=c["diffts"], enable_dmtp=c["dmtp"])
    grads = {}
    for dt in (torch.float32, bf):
        vin, tin = leaf(v.to(dt)), leaf(t.to(dt))

        def run(sd, vin=vin, tin=tin):
            return O.tokenizer_forward(sd, "u2tokenizer", vin, tin, oc)[0], G

        out, g = _oracle_grads(sd32, dt, run)
        g["v_token"], g["t_token"] = vin.grad, tin.grad
        grads[dt] = (out, g)
    tok = tok.to(bf).to(D).train()
    vd, td = leaf(v, D), leaf(t, D)
    got_out = tok(v_token=vd, t_token=td)
    assert rel(got_out.float(), grads[torch.float32][0]) <= 1.5 * rel(grads[bf][0].float(), grads[torch.float32][0]) + 1e-3
    (got_out.float() * G.to(D)).sum().backward()
    got = {"u2tokenizer." + k: p.grad for k, p in tok.named_parameters() if p.grad is not None}
    got["v_token"], got["t_token"] = vd.grad, td.grad
    ref32, ref16 = grads[torch.float32][1], grads[bf][1]
    # parameters the reference never reaches (linear_aggregator.wv / dense: tta.py:47-48,62-65; score_net of the hard
    # top-k) have no gradient on either side
    assert set(got) == set(ref32), set(got) ^ set(ref32)
    check_grads(got, ref32, ref16, what=name)


@pytest.mark.parametrize("case", ["mu2_2l", "linvt_b2_live"])
def test_tokenizer_gradients_vs_reference_fixture(case):
    """HIP backward against the REFERENCE modules' own backward (tests/golden/tokenizer_*_grads.npz, float64, written by
    make_golden.py grads): norm and name-seeded projection of every parameter gradient, column sample of the input
    gradients.  (tests/test_oracle_golden.py pins the oracle's autograd to the same fixtures to 1e-9; this closes the loop
    without the oracle in between, on the shipped rma + DiffTS + DMTP flavour and on the nn.MultiheadAttention ablation.)"""
    from cases import TOKENIZER_CASES, tokenizer_inputs
    from helpers import load_golden
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    c = TOKENIZER_CASES[case]
    g = load_golden(f"tokenizer_{case}_grads")
    tok = u2Tokenizer(c["E"], c["heads"], c["layers"], c["top_k"], c["use_multi_scale"], c["Q"], c["E"], c["attn_type"],
                      c["enable_diffts"], c["enable_dmtp"])
    sd32 = module_sd(tok, "u2tokenizer.", c["seed"], lively=c.get("lively", False))
    tok.load_state_dict({k[len("u2tokenizer."):]: v for k, v in sd32.items()})
    tok = tok.to(bf).to(D).train()
    v, t = tokenizer_inputs(c)
    G = synth.synth_tensor("grad_out", (c["B"], c["Q"], c["E"]), c["seed"])
    vd, td = leaf(v.to(bf), D), leaf(t.to(bf), D)
    out = tok(v_token=vd, t_token=td)
    (out.float() * G.to(D)).sum().backward()
    got = {"u2tokenizer." + k: p.grad.double().cpu() for k, p in tok.named_parameters() if p.grad is not None}
    names = [str(n) for n in g["names"]]
    assert set(got) == set(names), set(got) ^ set(names)
    top = float(g["norms"].max())
    bad = []
    tol = 0.35 if c.get("lively") else 6e-2  # (selective attention amplifies bf16 rounding -- 0.29 measured on the two-layer
    # gain-4 set; test_tokenizer_gradients_vs_oracle carries the bf16 reference's own distance as the yardstick)
    for k, n_ref, p_ref in zip(names, g["norms"], g["probes"]):
        n_ref, p_ref = float(n_ref), float(p_ref)
        probe = synth.synth_tensor(k + "/probe", tuple(got[k].shape), c["seed"]).double()
        scale = max(n_ref, 2e-3 * top)            # gradients that are tiny next to the largest: absolute floor
        e_norm = abs(got[k].norm().item() - n_ref) / scale
        e_probe = abs((got[k] * probe).sum().item() - p_ref) / (scale * probe.norm().item())
        if e_norm > tol or e_probe > tol:
            bad.append((k, round(e_norm, 4), round(e_probe, 4), n_ref / top))
    assert not bad, sorted(bad, key=lambda b: -max(b[1], b[2]))[:8]
    if c.get("lively"):
        # the INPUT gradients of a selective two-layer set have passed both residual-free attention layers backwards: the
        # reference's own bf16 run is ~100 % away from its float64 run there (chaotic regime) -- they are
        # compared with that yardstick in test_tokenizer_gradients_vs_oracle[linvt_2l_live], not against float64 here
        return
    ref_v = g["d_v_token_s8"].double()
    got_v = vd.grad.double().cpu()[..., ::8]
    e_v = (got_v - ref_v).norm().item() / ref_v.norm().item()
    assert e_v <= 0.15, e_v
    # In exact arithmetic the TEXT input of this parameter set has no influence on the output (the reference's float64
    # gradient is 1e-12 of the others).  Any bf16 run -- the reference's as well, see the bar of
    # test_tokenizer_gradients_vs_oracle -- breaks that cancellation with its rounding and leaves noise; it has to stay small
    # next to the real gradients.
    assert float(g["d_t_token_norm"]) <= 1e-6 * top and td.grad.double().norm().item() <= 0.1 * top


def test_vit_and_projector_gradients_vs_oracle():
    from u2tokenizer_amd.projector import SpatialPoolingProjector
    from u2tokenizer_amd.vit import ViT3DTower
    img, E, seed = [32, 64, 64], 256, 84
    vit = ViT3DTower(NS(vision_select_layer=-1, vision_select_feature="patch", image_channel=1, image_size=img,
                        patch_size=[4, 16, 16]))
    spp = SpatialPoolingProjector(img, [4, 16, 16], 768, E, "mlp", 2, "spatial", 2)
    sd32 = {**module_sd(vit, "vision_tower.", seed), **module_sd(spp, "mm_projector.", seed)}
    vit.load_state_dict({k[len("vision_tower."):]: v for k, v in sd32.items() if k.startswith("vision_tower.")})
    spp.load_state_dict({k[len("mm_projector."):]: v for k, v in sd32.items() if k.startswith("mm_projector.")})
    vol = synth.synth_volume(1, 2, img, seed=seed, dtype=torch.float16).view(2, 1, *img)
    G = synth.synth_tensor("grad_out", (2, 16, E), seed)
    oc = O.PathConfig(image_size=img, hidden_size=E)
    grads = {}
    for dt in (torch.float32, bf):
        def run(sd, dt=dt):
            f = O.vit_tower_forward(sd, "vision_tower.vision_tower", vol.to(dt), oc)
            return O.spp_forward(sd, "mm_projector", f, oc), G
        grads[dt] = _oracle_grads(sd32, dt, run)
    vit, spp = vit.to(bf).to(D).train(), spp.to(bf).to(D).train()
    out = spp(vit(vol.to(D)))
    assert rel(out.float(), grads[torch.float32][0]) <= 1.5 * rel(grads[bf][0].float(), grads[torch.float32][0]) + 1e-3
    (out.float() * G.to(D)).sum().backward()
    got = {"vision_tower." + k: p.grad for k, p in vit.named_parameters()}
    got.update({"mm_projector." + k: p.grad for k, p in spp.named_parameters()})
    assert set(got) == set(grads[torch.float32][1])
    check_grads(got, grads[torch.float32][1], grads[bf][1], what="vit + spp")


@pytest.mark.host_heavy(39)   # nominal seconds, mostly host (tests/suite_budget.py)
def test_gradients_at_full_width():
    """The training path at the real WIDTH of BASELINE configs 3 / 4 (reduced depth so that torch.autograd over the CPU oracle
    stays within seconds): two ViT blocks on two 32 x 256 x 256 chunks (2049 tokens per chunk: the fused attention backward,
    the K-major weight-gradient products with K = 4098 token rows, LayerNorm / bias reductions over them) and one
    SVR + DiffTS + DMTP + TTA layer of the tokenizer at E = 4096 with 8 x 256 visual and 1024 text tokens.  Every
    parameter's gradient against the fp32 oracle, with the bf16 oracle's own distance as the bar."""
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    from u2tokenizer_amd.vit import ViT
    img, seed = [32, 256, 256], 91
    vit = ViT(1, img, [4, 16, 16], num_layers=2, pos_embed="perceptron", classification=True)
    sd32 = module_sd(vit, "vision_tower.vision_tower.", seed)
    vit.load_state_dict({k[len("vision_tower.vision_tower."):]: v for k, v in sd32.items()})
    vol = synth.synth_volume(1, 2, img, seed=seed, dtype=torch.float16).view(2, 1, *img)
    G = synth.synth_tensor("grad_out", (2, 2048, 768), seed)
    oc = O.PathConfig(image_size=img, vit_layers=2)
    grads = {}
    for dt in (torch.float32, bf):
        def run(sd, dt=dt):
            return O.vit_tower_forward(sd, "vision_tower.vision_tower", vol.to(dt), oc), G
        grads[dt] = _oracle_grads(sd32, dt, run)
    vit = vit.to(bf).to(D).train()
    out = vit.forward_features(vol.to(D), keep_cls=False)
    assert rel(out.float(), grads[torch.float32][0]) <= 1.5 * rel(grads[bf][0].float(), grads[torch.float32][0]) + 1e-3
    (out.float() * G.to(D)).sum().backward()
    got = {"vision_tower.vision_tower." + k: p.grad for k, p in vit.named_parameters()}
    assert set(got) == set(grads[torch.float32][1])
    check_grads(got, grads[torch.float32][1], grads[bf][1], what="2-block ViT at 2049 tokens")
    del vit, got, grads, out

    E, seed = 4096, 92
    args = (E, 8, 1, 1024, True, 256, E, "rma", True, True)
    sd32 = module_sd(u2Tokenizer(*args), "u2tokenizer.", seed)
    for n, t in sd32.items():
        synth.lively_(n, t, qk_gain=2.0)
    v = synth.synth_tensor("v_token", (1, 8, 256, E), seed).to(bf)
    t = (0.25 * synth.synth_tensor("t_token", (1, 1024, E), seed)).to(bf)
    G = synth.synth_tensor("grad_out", (1, 256, E), seed)
    oc = O.PathConfig(hidden_size=E, u2t_num_layers=1)
    grads = {}
    for dt in (torch.float32, bf):
        vin, tin = leaf(v.to(dt)), leaf(t.to(dt))

        def run(sd, vin=vin, tin=tin):
            return O.tokenizer_forward(sd, "u2tokenizer", vin, tin, oc)[0], G

        out, g = _oracle_grads(sd32, dt, run)
        g["v_token"], g["t_token"] = vin.grad, tin.grad
        grads[dt] = (out, g)
    with torch.device("meta"):
        tok = u2Tokenizer(*args)
    tok = tok.to(bf).to_empty(device=D)
    tok.load_state_dict({k[len("u2tokenizer."):]: val.to(bf) for k, val in sd32.items()})
    tok.train()
    vd, td = leaf(v, D), leaf(t, D)
    got_out = tok(v_token=vd, t_token=td)
    assert rel(got_out.float(), grads[torch.float32][0]) <= 1.5 * rel(grads[bf][0].float(), grads[torch.float32][0]) + 1e-3
    (got_out.float() * G.to(D)).sum().backward()
    got = {"u2tokenizer." + k: p.grad for k, p in tok.named_parameters() if p.grad is not None}
    got["v_token"], got["t_token"] = vd.grad, td.grad
    assert set(got) == set(grads[torch.float32][1]), set(got) ^ set(grads[torch.float32][1])
    check_grads(got, grads[torch.float32][1], grads[bf][1], what="one tokenizer layer at E = 4096")


def test_training_step_through_the_hf_model():
    """stage-1 style step (train_stage1.py:244-251): model(images, input_ids, labels, attention_mask, question_ids) on the
    GPU with HIP forward + backward; loss and the gradients of the path's parameters (and of the embedding table rows)
    against oracle + the same HF decoder on the host in fp32."""
    from cases import FULL_CASES
    from test_oracle_golden import _full_model, full_path_cfg
    torch.set_grad_enabled(True)  # (test_oracle_golden switches autograd off at import)
    c = FULL_CASES["cfg1"]
    m, cfg = _full_model(c)
    m.train()
    vol = synth.synth_volume(c["B"], c["C"], c["mm"]["image_size"], seed=c["seed"], dtype=torch.float16)
    ids = synth.synth_ids(c["B"], c["S"], c["n_real"], cfg.vocab_size, seed=c["seed"], name="input_ids")
    qids = synth.synth_ids(c["B"], c["Lt"], c["n_q"], cfg.vocab_size, seed=c["seed"], name="question_ids")
    labels = ids.clone()
    labels[:, :20] = -100
    # host reference: oracle path (differentiated) -> the HF decoder in fp32
    sd = {k: leaf(v) for k, v in m.state_dict().items() if v.is_floating_point()}
    emb, _ = O.prepare_inputs_for_multimodal(sd, sd["model.embed_tokens.weight"], ids, vol.float(), qids, full_path_cfg(c))
    dec = {k: v for k, v in sd.items() if not any(s in k for s in ("vision_tower", "mm_projector", "u2tokenizer"))}
    loss32 = torch.func.functional_call(m, {k: v for k, v in dec.items()}, args=(),
                                        kwargs=dict(inputs_embeds=emb, labels=labels)).loss
    loss32.backward()
    ref = {k: v.grad for k, v in sd.items() if v.grad is not None
           and any(s in k for s in ("vision_tower", "mm_projector", "u2tokenizer", "embed_tokens"))}
    # the model's embed_tokens is nn.Embedding(padding_idx=pad_token_id): its backward leaves the pad row without a
    # gradient (the reference looks tokens up through that module, u2_arch.py:109,113); the oracle's F.embedding has no
    # padding_idx -- same forward, so drop the pad row from the reference gradient
    ref["model.embed_tokens.weight"][cfg.pad_token_id] = 0
    mg = m.to(bf).to(D)
    for p in mg.parameters():
        p.requires_grad_(True)
    out = mg(images=vol.to(D), input_ids=ids.to(D), labels=labels.to(D), question_ids=qids.to(D))
    assert abs(out.loss.item() - loss32.item()) < 3e-2 * abs(loss32.item()), (out.loss.item(), loss32.item())
    out.loss.backward()
    got = {k: p.grad for k, p in mg.named_parameters() if k in ref}
    check_grads(got, ref, what="training step")


@pytest.mark.parametrize("n,off", [(8 * 4096, 0), (8 * 4096 + 5, 0), (4099, 3), (7, 0), (2048 * 3 + 1, 8)])
def test_adamw_step_kernel_matches_torch_adamw(n, off):
    """u2tok_adamw_step (fused ZeRO-1 shard update) against torch.optim.AdamW on the same fp32 state: the 8-wide form, its
    tail, an unaligned piece (offset slices of larger buffers) and two parameter groups with a device-side clip coefficient."""
    from u2tokenizer_amd import ops
    g = torch.Generator().manual_seed(n + off)
    tot = n + off
    master = torch.randn(tot, generator=g)
    grads = [(torch.randn(tot, generator=g) * 0.1).to(bf) for _ in range(3)]
    group = (torch.arange(tot) % 3 == 0).to(torch.uint8)            # group 1: no decay, its own lr
    lrs, wds, coef = [1e-2, 3e-2], [0.1, 0.0], 0.37
    pa = master[off:][group[off:] == 0].clone().requires_grad_(True)
    pb = master[off:][group[off:] == 1].clone().requires_grad_(True)
    ref = torch.optim.AdamW([{"params": [pa], "lr": lrs[0], "weight_decay": wds[0]},
                             {"params": [pb], "lr": lrs[1], "weight_decay": wds[1]}], betas=(0.9, 0.95), eps=1e-8)
    md, m1, m2 = master.to(D), torch.zeros(tot, device=D), torch.zeros(tot, device=D)
    out = torch.zeros(tot, dtype=bf, device=D)
    cf = torch.tensor([coef], device=D)
    for step, gr in enumerate(grads, 1):
        gf = gr.float()[off:] * coef * 0.5
        pa.grad, pb.grad = gf[group[off:] == 0].clone(), gf[group[off:] == 1].clone()
        ref.step()
        ops.adamw_step(md[off:], m1[off:], m2[off:], gr.to(D)[off:], out[off:], step, lrs, wds, betas=(0.9, 0.95), eps=1e-8,
                       grad_scale=0.5, grad_coef=cf, group=group.to(D)[off:])
    want = torch.empty(n)
    want[group[off:] == 0], want[group[off:] == 1] = pa.detach(), pb.detach()
    got = md[off:].cpu()
    assert torch.allclose(got, want, rtol=2e-5, atol=1e-6), (got - want).abs().max()
    assert torch.equal(out[off:].cpu(), got.to(bf))
    if off:
        assert torch.equal(md[:off].cpu(), master[:off]) and not out[:off].any()


def test_zero1_adamw_on_the_gpu_keeps_packed_weights_in_sync():
    """Two optimiser steps of dp.Zero1AdamW (one rank: no collective) on a GPU tokenizer trained through the HIP path: the
    parameters follow torch.optim.AdamW on an fp32 copy fed with the same gradients, the in-place updates land in the packed
    q | k | v buffers (the fused inference forward reads them), and the loss goes down."""
    from u2tokenizer_amd import dp
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    E, seed = 512, 93
    tok = u2Tokenizer(E, 8, 1, 64, True, 16, E, "rma", True, True)
    sd32 = module_sd(tok, "u2tokenizer.", seed)
    tok.load_state_dict({k[len("u2tokenizer."):]: v for k, v in sd32.items()})
    tok = tok.to(bf).to(D).train()
    ref = {n: p.detach().float().clone().requires_grad_(True) for n, p in tok.named_parameters()}
    opt = dp.Zero1AdamW(tok.parameters(), lr=2e-3, weight_decay=0.01)
    ropt = torch.optim.AdamW(list(ref.values()), lr=2e-3, weight_decay=0.01)
    v = synth.synth_tensor("v_token", (2, 4, 32, E), seed).to(bf).to(D)
    t = (0.25 * synth.synth_tensor("t_token", (2, 24, E), seed)).to(bf).to(D)
    target = synth.synth_tensor("target", (2, 16, E), seed).to(D)
    losses = []
    for _ in range(3):
        out = tok(v_token=v, t_token=t)
        loss = (out.float() - target).pow(2).mean()
        losses.append(loss.item())
        loss.backward()
        for n, p in tok.named_parameters():
            ref[n].grad = None if p.grad is None else p.grad.detach().float().clone()
        opt.step()
        ropt.step()
        opt.zero_grad()
        for n, p in tok.named_parameters():
            if ref[n].grad is None:
                continue
            # Zero1AdamW keeps an fp32 master and writes its bf16 rounding; the reference's bf16 rounding may differ by an ulp
            assert torch.allclose(p.detach().float(), ref[n].detach().to(bf).float(), rtol=2 ** -7, atol=1e-6), n
        m = tok.svt_module.attention_network.layers[0].spatial_attention
        es = m.wq.weight.element_size()
        assert m.wk.weight.data_ptr() == m.wq.weight.data_ptr() + m.wq.weight.numel() * es, "packing lost by the update"
        with torch.no_grad():
            fused = tok(v_token=v, t_token=t)            # inference path: reads the packed buffers inside the library
        again = tok(v_token=v, t_token=t)                # autograd path: reads the parameters
        assert rel(fused.float(), again.detach().float()) < 2e-2
    assert losses[-1] < losses[0], losses


def test_dpo_duplicate_image_batch_is_deduplicated():
    """cat([images, images]) with the same question (dpo_u2trainer.py:160-162): the path runs once per distinct
    (image, question); results and gradients equal the plain run."""
    from cases import FULL_CASES
    from test_oracle_golden import _full_model
    torch.set_grad_enabled(True)  # (test_oracle_golden switches autograd off at import)
    c = FULL_CASES["cfg1"]
    m, cfg = _full_model(c)
    mg = m.to(bf).to(D).train()
    vol = synth.synth_volume(1, c["C"], c["mm"]["image_size"], seed=c["seed"], dtype=torch.float16).to(D)
    ids = synth.synth_ids(2, c["S"], c["n_real"], cfg.vocab_size, seed=c["seed"], name="input_ids").to(D)
    qids = synth.synth_ids(1, c["Lt"], c["n_q"], cfg.vocab_size, seed=c["seed"], name="question_ids").to(D)
    vol2, qids2 = torch.cat([vol, vol]), torch.cat([qids, qids])
    res = {}
    for flag in (True, False):
        mg.config.u2_dedup_duplicate_images = flag
        mg.zero_grad(set_to_none=True)
        emb = mg.prepare_inputs_for_multimodal(ids, None, None, None, None, vol2, qids2)[4]
        emb.float().square().sum().backward()
        res[flag] = (emb.detach(), {k: p.grad.clone() for k, p in mg.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][0][0, 1:17], res[True][0][1, 1:17])
    assert rel(res[True][0].float(), res[False][0].float()) < 1e-3
    for k, g in res[False][1].items():
        # (a key bias shifts every logit of a row alike: its exact gradient is zero, what a bf16 run leaves there is rounding
        # noise of the other gradients' size -- compared against that size, not against itself)
        floor = 1e-3 * g.float().abs().max().item()
        if k.endswith(".wk.bias"):
            floor = res[False][1][k.replace(".wk.bias", ".wq.bias")].float().abs().max().item()
        assert rel(res[True][1][k].float(), g.float(), floor) < 5e-2, k


def test_config4_full_depth_full_width_gradients():
    """BASELINE configs[3], the stage-1 training step of the path at its REAL size (train_stage1.py:244-251 with
    freeze_vision_tower False): 12 ViT blocks on 8 chunks of (32,256,256) -> SPP -> the 4-layer rma + DiffTS + DMTP tokenizer
    at E = 4096 -> embedding splice, HIP forward + backward, the gradient of EVERY parameter against torch.autograd over the
    oracle in fp32.  The oracle side (3 minutes of host time and 45 GB) was run in the build container and is committed as
    tests/golden/config4_grads.npz (tests/golden/make_config4_grads.py: per parameter the gradient norm, a strided sample of
    1024 entries, and the distance of the reference's own bf16 run -- the yardstick of check_grads)."""
    import sys
    from pathlib import Path
    from types import SimpleNamespace as NS
    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_config4_grads as M
    from cases import CONFIG4_CASE as c
    from helpers import load_golden
    from test_gpu_configs import PathHolder, mm_config, record
    from u2tokenizer_amd.arch import u2MetaForCausalLM
    g = load_golden("config4_grads")
    names = [str(n) for n in g["names"]]
    mc = mm_config(c["E"], c["image_size"])

    class PathOnly(u2MetaForCausalLM):
        def __init__(self, holder):
            self.holder, self.config = holder, NS(**mc)

        def get_model(self):
            return self.holder

    with torch.device("meta"):
        holder = PathHolder(mc, c["vocab"])
    sd = M.path_state_dict(c, bf)
    sd[M.SPP_BIAS] = g["spp_bias"].to(bf)
    holder = holder.to(bf).to_empty(device=D)
    holder.load_state_dict({k[len("model."):]: v for k, v in sd.items()}, strict=True)
    del sd
    for p in holder.parameters():
        p.requires_grad_(True)
    holder.train()
    vol, ids, qids, G = M.config4_inputs(c)
    emb = PathOnly(holder).prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4]
    tok = slice(1, 1 + c["Q"])
    ref_tok = g["out_tokens_s16"].double()
    e_out = ((emb[0, tok, ::16].double().cpu() - ref_tok).norm() / ref_tok.norm()).item()
    (emb.float() * G.to(D)).sum().backward()
    got = {"model." + k: p.grad for k, p in holder.named_parameters() if p.grad is not None}
    assert set(got) == set(names), set(got) ^ set(names)
    floor = float(g["floor"])
    rep, bad, dot, na, nb = {}, {}, 0.0, 0.0, 0.0
    for i, k in enumerate(names):
        gk = got[k].detach().float().flatten()
        assert torch.isfinite(gk).all(), k
        idx = M.sample_index(k, gk.numel()).to(D)
        s_hip = gk[idx].double().cpu()
        s_ref = g["samples"][i, : idx.numel()].double()
        rms_ref = float(g["norms"][i]) / float(g["numels"][i]) ** 0.5
        e = ((s_hip - s_ref).pow(2).mean().sqrt() / (rms_ref + floor)).item()
        e_norm = abs(gk.double().norm().item() - float(g["norms"][i])) / (float(g["norms"][i]) + floor * float(g["numels"][i]) ** 0.5)
        bar = 2e-2 + 1.5 * float(g["rel16"][i])
        rep[k] = {"sample_rel": e, "norm_rel": e_norm, "bar": bar, "share_of_top": rms_ref / (floor / 2e-3)}
        if e > bar + 0.03 or e_norm > bar + 0.03:   # (+ 3 %: a 1024-entry sample estimates the tensor's error to a few per cent)
            bad[k] = rep[k]
        dot, na, nb = dot + (s_hip @ s_ref).item(), na + (s_hip @ s_hip).item(), nb + (s_ref @ s_ref).item()
    cos = dot / (na * nb) ** 0.5
    record("config4_full_depth_gradients", {"out_tokens_rel_vs_fp32": e_out, "out_rel_bf16_oracle": float(g["out_rel16"]),
                                            "cosine_hip_vs_fp32_on_samples": cos, "cosine_bf16_oracle_vs_fp32": float(g["cos16"]),
                                            "parameters": len(names), "outside_bar": bad,
                                            "worst": dict(sorted(rep.items(), key=lambda kv: kv[1]["sample_rel"] - kv[1]["bar"])[-8:])})
    assert e_out <= 1.5 * float(g["out_rel16"]) + 1e-3, (e_out, float(g["out_rel16"]))
    assert not bad, dict(list(bad.items())[:8])
    assert cos >= float(g["cos16"]) - 0.02, (cos, float(g["cos16"]))


def test_backward_runs_on_the_context_of_its_forward():
    """Function.backward runs on the autograd engine's thread, whose thread-local context stack is empty: every Function
    remembers the context its forward ran on and re-enters it.  Visible through the profiling records: the two
    backward GEMMs (dX, dW) of a linear layer used inside `with ops.Context()` must land on THAT context."""
    import ctypes as C
    from u2tokenizer_amd import _lib, autograd as AG, ops

    def gemm_launches(ctx):
        h = _lib.load_library()
        prev = h.u2tok_ctx_get_current()
        h.u2tok_ctx_set_current(ctx.handle)
        try:
            ms, fl, cnt = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_int64 * 6)()
            _lib.check(h.u2tok_profile_collect(ms, fl, cnt, 6), "u2tok_profile_collect")
        finally:
            h.u2tok_ctx_set_current(prev)
        return cnt[0]

    x, w = leaf(rnd(64, 256, seed=1), D), leaf(rnd(128, 256, seed=2), D)
    c = ops.Context()
    try:
        c.set_option("profile", 1)
        with c:
            y = AG.linear(x, w)
        assert gemm_launches(c) == 1
        y.float().sum().backward()              # outside the with-block: the engine thread has no bound context of its own
        torch.cuda.synchronize()
        assert gemm_launches(c) == 2
        assert x.grad is not None and w.grad is not None
    finally:
        c.close()


def test_dpo_duplicate_image_batch_at_benchmark_size():
    """The same at BASELINE configs[2] size (E = 4096, one 256^3 volume duplicated as the DPO trainer does): the de-duplicated
    run equals the plain one -- embeddings bit for bit on the duplicated rows, every gradient within the bf16 noise of summing
    two halves instead of doubling one -- at about half the time (only the toy size was covered)."""
    import sys
    import time
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    torch.set_grad_enabled(True)
    E, vocab = 4096, 32768
    path, _ = bench.build_path(E, vocab, D)
    for p in path.holder.parameters():
        p.requires_grad_(True)
    g = torch.Generator(device=D).manual_seed(5)
    vol = torch.rand((1, 8, 32, 256, 256), device=D, generator=g).half()
    ids = torch.randint(1, vocab, (2, 1024), device=D, generator=g)
    qids = torch.zeros((1, 1024), dtype=torch.int64, device=D)
    qids[:, :40] = torch.randint(1, vocab, (1, 40), device=D, generator=g)
    vol2, qids2 = torch.cat([vol, vol]), torch.cat([qids, qids])
    w = torch.randn(2, 1024, E, device=D, generator=g)
    res, ms = {}, {}
    for flag in (True, False, True, False):
        path.config.u2_dedup_duplicate_images = flag
        for p in path.holder.parameters():
            p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        emb = path.prepare_inputs_for_multimodal(ids, None, None, None, None, vol2, qids2)[4]
        (emb.float() * w).sum().backward()
        torch.cuda.synchronize()
        ms[flag] = min(ms.get(flag, float("inf")), (time.perf_counter() - t0) * 1e3)   # best of the two passes (a host stall is not the subject)
        res[flag] = (emb.detach(), {k: p.grad.float() for k, p in path.holder.named_parameters() if p.grad is not None})
    path.config.u2_dedup_duplicate_images = True

    def drel(a, b, floor=0.0):  # on the device: 2.1 B gradient values
        return ((a - b).double().pow(2).mean().sqrt() / (b.double().pow(2).mean().sqrt() + floor)).item()

    assert torch.equal(res[True][0][0, 1:257], res[True][0][1, 1:257])   # the repeated result: the 256 spliced visual tokens
    e_emb = drel(res[True][0].float(), res[False][0].float())    # (B = 1 and B = 2 launches slice K differently: not bit-equal)
    assert e_emb < 1e-2, e_emb
    assert len(res[True][1]) == len(res[False][1]) > 300
    # the yardstick of check_grads: distances against a floor of 2e-3 x the largest per-parameter RMS gradient (parameters
    # whose exact gradient is ~0 in this regime -- key biases, the first layers' bias tables and query weights -- are rounding
    # noise in any bf16 run, and the two runs round differently: B = 1 and B = 2 launches cut K differently), and the
    # direction of the whole gradient vector
    top = max(gr.double().pow(2).mean().sqrt().item() for gr in res[False][1].values())
    worst, dot, na, nb = 0.0, 0.0, 0.0, 0.0
    for k, gr in res[False][1].items():
        e = drel(res[True][1][k], gr, 2e-3 * top)
        worst = max(worst, e)
        assert e < 5e-2, (k, e)
        x, y = res[True][1][k].double().flatten(), gr.double().flatten()
        dot, na, nb = dot + (x @ y).item(), na + (x @ x).item(), nb + (y @ y).item()
    cosine = dot / (na * nb) ** 0.5
    assert cosine > 0.9995, cosine
    assert ms[True] < 0.75 * ms[False], ms
    print(f"DPO de-dup at E=4096: {ms[True]:.1f} ms against {ms[False]:.1f} ms; embeddings {e_emb:.2e}, worst gradient {worst:.2e}, cosine {cosine:.6f}")


def test_zero1_collective_path_on_rccl_world_of_one():
    """The ZeRO-1 exchange exactly as the 8-GPU run issues it -- bf16 reduce_scatter_tensor on the communication stream, the scalar
    all-reduce of the clipping norm, all_gather_into_tensor + copy-out under the next bucket's AdamW, events both ways -- on the
    "nccl" backend (= RCCL) with a process group of ONE rank (`force_collectives`): the only way to execute that code path on a
    single-GPU box (VERDICT r2 missing 7).  Result: bit-identical to the collective-free single-rank path."""
    import socket
    import torch.distributed as dist
    from u2tokenizer_amd import dp
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.set_grad_enabled(True)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    try:
        def model(seed):
            g = torch.Generator().manual_seed(seed)
            ps = [torch.nn.Parameter((torch.randn(n, generator=g) * 0.1).to(bf).to(dev)) for n in (70_000, 8, 33_001, 4096, 50_000)]
            return ps
        outs = {}
        for forced in (False, True):
            ps = model(7)
            opt = dp.Zero1AdamW([{"params": ps[:2], "weight_decay": 0.1}, {"params": ps[2:], "weight_decay": 0.0}], lr=1e-2,
                                reduce_bucket_size=60_000, max_grad_norm=0.5, gradient_accumulation_steps=2,
                                force_collectives=forced)
            assert opt._multi == forced and len(opt.buckets) >= 3
            x = torch.linspace(-1, 1, 70_000, device=dev)
            for step in range(3):
                for micro in range(2):
                    loss = (ps[0].float() * x).pow(2).sum() * (1 + micro) + sum((p.float() - 0.05 * (step + 1)).pow(2).sum() for p in ps[1:])
                    loss.backward()
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            outs[forced] = ([p.detach().clone() for p in ps], float(opt.last_grad_norm))
        for a, b in zip(outs[False][0], outs[True][0]):
            assert torch.equal(a, b)
        assert outs[False][1] == outs[True][1]
    finally:
        dist.destroy_process_group()

