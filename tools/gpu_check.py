"""Verbose GPU diagnostics (run on the MI355X box through gpurun; prints, never asserts).

    python tools/gpu_check.py <section> [...]      sections: gemm ops attn modules perf

Each section compares the HIP kernels with fp32 CPU computations of the same bf16 inputs (or with the oracle /
golden vectors for the module-level sections) and prints error statistics; `perf` times the big shapes.
"""
import math
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from u2tokenizer_amd import ops, synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
bf = torch.bfloat16


def stats(name, got, ref, extra=""):
    got, ref = got.detach().double().cpu().flatten(), ref.detach().double().cpu().flatten()
    d = (got - ref).abs()
    bad = (~torch.isfinite(got)).sum().item()
    scale = ref.abs().max().item()
    print(f"  {name:58s} max_abs={d.max().item():.3e} mean_abs={d.mean().item():.3e} ref_max={scale:.3e} "
          f"rel_max={d.max().item() / max(scale, 1e-30):.3e} nonfinite={bad} {extra}", flush=True)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(bf)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


# ------------------------------------------------------------------------------------------- gemm
def sec_gemm():
    for glds in (1,):
        for tile in (64, 128):
            ops.set_option("gemm_tile", tile)
            print(f"[gemm] tile={tile}", flush=True)
            for (M, N, K) in [(128, 128, 64), (256, 256, 256), (300, 200, 136), (77, 520, 72), (1000, 768, 1024),
                              (2049, 2304, 768)]:
                a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
                ref = a.float() @ b.float().t()
                got = ops.gemm(a.to(dev), b.to(dev))
                stats(f"plain {M}x{N}x{K} bf16", got, ref)
            M, N, K = 300, 264, 200
            a, b, bias, res = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
            ref = F.gelu(a.float() @ b.float().t() + bias.float()) + res.float()
            got = ops.gemm(a.to(dev), b.to(dev), bias=bias.to(dev), residual=res.to(dev), gelu=True)
            stats("bias+gelu+residual 300x264x200", got, ref)
            got = ops.gemm(a.to(dev), b.to(dev), bias=bias.to(dev), out_f32=True, alpha=0.5)
            stats("bias, fp32 out, alpha .5", got, 0.5 * (a.float() @ b.float().t()) + bias.float())
            bm = rnd(M, seed=7)
            got = ops.gemm(a.to(dev), b.to(dev), bias=bm.to(dev), bias_m=True, out_f32=True)
            stats("bias along M, fp32 out", got, a.float() @ b.float().t() + bm.float()[:, None])
            # N not a multiple of 4 -> scalar epilogue
            b2, bias2 = rnd(203, K, seed=8), rnd(203, seed=9)
            got = ops.gemm(a.to(dev), b2.to(dev), bias=bias2.to(dev))
            stats("N=203 (scalar epilogue)", got, a.float() @ b2.float().t() + bias2.float())
            # batched, per-batch B
            a3, b3 = rnd(6, 100, 64, seed=10), rnd(6, 90, 64, seed=11)
            got = ops.gemm(a3.to(dev), b3.to(dev), out_f32=True)
            stats("batched 6x(100x90x64)", got, torch.einsum("zmk,znk->zmn", a3.float(), b3.float()))
            # strided head-batched QK^T: q,k (nb, S, H*d) -> (nb*H, S, S)
            nb, S, H, d = 2, 50, 4, 64
            q, k = rnd(nb, S, H * d, seed=12), rnd(nb, S, H * d, seed=13)
            qd, kd = q.to(dev), k.to(dev)
            out = torch.empty((nb * H, S, 56), dtype=torch.float32, device=dev)
            from u2tokenizer_amd import _lib
            h = _lib.load_library()
            st = h.u2tok_gemm_bf16(qd.data_ptr(), kd.data_ptr(), out.data_ptr(), None, None, S, S, d, H * d, H * d, 56,
                                   0, nb * H, H, S * H * d, d, S * H * d, d, H * S * 56, S * 56, 0, 0, 1.0, 16,
                                   torch.cuda.current_stream().cuda_stream)
            ref = torch.einsum("bshd,bthd->bhst", q.float().view(nb, S, H, d), k.float().view(nb, S, H, d))
            stats(f"head-strided QK^T status={st}", out[:, :, :S].reshape(nb, H, S, S), ref)
    ops.set_option("gemm_tile", 0)


# ------------------------------------------------------------------------------------------- ops
def sec_ops():
    print("[ops]", flush=True)
    for C_ in (768, 512, 2048, 4096):
        x, r, w, b = rnd(37, C_, seed=1), rnd(37, C_, seed=2), rnd(C_, seed=3), rnd(C_, seed=4)
        got = ops.layernorm(x.to(dev), w.to(dev), b.to(dev))
        stats(f"layernorm C={C_}", got, F.layer_norm(x.float(), (C_,), w.float(), b.float()))
        got = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), residual=r.to(dev))
        stats(f"layernorm+res C={C_}", got, F.layer_norm(x.float() + r.float(), (C_,), w.float(), b.float()))
    for (Z, R, n) in [(8, 40, 40), (3, 17, 1792), (2, 256, 256), (4, 9, 13)]:
        s = torch.randn(Z, R, n, generator=torch.Generator().manual_seed(n)) * 3
        got = ops.softmax_rows(s.to(dev), scale=0.7)
        ref = F.softmax(s * 0.7, dim=-1)
        stats(f"softmax {Z}x{R}x{n}", got[:, :, :n], ref, extra=f"pad_abs_max={got[:, :, n:].abs().max().item() if got.shape[2] > n else 0}")
    H, L = 4, 512
    tbl = rnd(2 * L - 1, H, scale=0.5, seed=5)
    s = torch.randn(2 * H, 40, 40, generator=torch.Generator().manual_seed(7))
    got = ops.softmax_rows(s.to(dev), scale=0.5, rel_bias=tbl.to(dev), heads=H, max_len=L)
    pos = torch.arange(40)
    bias = tbl.float()[pos[None, :] - pos[:, None] + L - 1].permute(2, 0, 1)  # (H, i, j)
    ref = F.softmax(s.view(2, H, 40, 40) * 0.5 + bias[None], dim=-1).view(2 * H, 40, 40)
    stats("softmax + toeplitz bias", got[:, :, :40], ref)
    x = rnd(3, 70, 130, seed=8)
    got = ops.transpose(x.to(dev), ld_out=72)
    stats("transpose 3x70x130 (ld 72)", got[:, :, :70], x.float().transpose(1, 2), extra=f"pad={got[:, :, 70:].abs().max().item()}")
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        vol = torch.rand(2, 1, 8, 32, 32, generator=torch.Generator().manual_seed(3)).to(dt)
        got = ops.im2col(vol.to(dev), (4, 16, 16))
        v = vol.to(bf).float()
        ref = v.reshape(2, 1, 2, 4, 2, 16, 2, 16).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(2, 8, 1024)
        stats(f"im2col {dt}", got, ref)
    x = rnd(2, 8 * 4 * 4, 768, seed=9)
    got = ops.avgpool3d_tokens(x.to(dev), (8, 4, 4), (2, 2, 2))
    ref = F.avg_pool3d(x.float().view(2, 8, 4, 4, 768).permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1).reshape(2, -1, 768)
    stats("avgpool3d 2x2x2", got, ref)
    got = ops.avgpool3d_tokens(x.to(dev), (1, 1, 128), (1, 1, 8))
    stats("avgpool sequence(8)", got, F.avg_pool1d(x.float().permute(0, 2, 1), 8, 8).permute(0, 2, 1))
    table = rnd(100, 64, seed=10)
    ids = torch.randint(0, 100, (2, 12), generator=torch.Generator().manual_seed(1))
    feats = rnd(2, 5, 64, seed=11)
    got = ops.embed_splice(table.to(dev), ids.to(dev), feats.to(dev))
    emb = table.float()[ids]
    ref = torch.cat((emb[:, :1], feats.float(), emb[:, 6:]), 1)
    stats("embed+splice", got, ref)
    stats("embed only", ops.embed_splice(table.to(dev), ids.to(dev)), emb)
    x, w, b = rnd(3, 200, 512, seed=12), rnd(1, 512, seed=13), rnd(1, seed=14)
    sc = ops.score_gemv(x.to(dev), w.to(dev), b.to(dev))
    ref = (x.double() @ w.double().t()).squeeze(-1) + b.double()
    print(f"  score_gemv bit-exact vs fp64->fp32: {torch.equal(sc.cpu(), ref.float())} "
          f"max_abs={(sc.cpu().double() - ref).abs().max().item():.3e}", flush=True)
    for (B, n, k) in [(3, 200, 50), (2, 2048, 1024), (1, 32, 16), (2, 33, 33)]:
        s = torch.randn(B, n, generator=torch.Generator().manual_seed(n))
        s = (s * 4).round() / 4  # many exact ties
        s[0, 1] = -0.0
        s[0, 2] = 0.0
        idx = ops.topk_sorted(s.to(dev), k).cpu()
        ref = torch.sort(s + 0.0, dim=1, descending=True, stable=True).indices[:, :k]
        print(f"  topk B={B} n={n} k={k} exact={torch.equal(idx, ref)}", flush=True)
    x = rnd(2, 50, 64, seed=15)
    idx = torch.randint(0, 50, (2, 20), generator=torch.Generator().manual_seed(2))
    stats("gather_rows", ops.gather_rows(x.to(dev), idx.to(dev)), x.float()[torch.arange(2)[:, None], idx])
    from oracle import u2_oracle as O
    for k in (64, 30, 3, 1):
        x = rnd(2, k, 512, seed=16)
        got = ops.multiscale_pool(x.to(dev))
        stats(f"multiscale fixed k={k}", got, O.multi_scale_pool({}, None, x.float()))
        gw, gb = rnd(1, 512, scale=0.3, seed=17), rnd(1, seed=18)
        got = ops.multiscale_pool(x.to(dev), gw.to(dev), gb.to(dev))
        sd = {"p.gate_fc.weight": gw.float(), "p.gate_fc.bias": gb.float()}
        stats(f"multiscale dmtp k={k}", got, O.multi_scale_pool(sd, "p", x.float()))
    x = rnd(2 * 3 * 5, 3 * 4 * 64, seed=19)  # rows (b t n), (q|k|v) x heads x d ... use H=4,d=64 on first 256 cols
    xd = x.to(dev).clone()
    ops.rope_apply(xd[:, :256], 2, 3, 5, 4, 64)
    xx = x.float()[:, :256].view(2, 3, 5, 4, 64).permute(0, 2, 3, 1, 4)  # (b, n, h, t, d)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    fr = torch.einsum("i,j->ij", torch.arange(512, dtype=torch.float32), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos()[:3].to(bf).float(), emb.sin()[:3].to(bf).float()
    ref = xx * cos + O._rotate_half(xx) * sin
    stats("rope (temporal layout)", xd[:, :256].float().cpu().view(2, 3, 5, 4, 64).permute(0, 2, 3, 1, 4), ref)
    stats("rope leaves other cols", xd[:, 256:], x.float()[:, 256:])


# ------------------------------------------------------------------------------------------- attention
def sec_attn():
    print("[attn]", flush=True)
    for (B, T, N, H, d) in [(1, 8, 6, 8, 512), (2, 4, 5, 8, 256), (1, 2, 16, 8, 64), (1, 3, 7, 4, 128), (1, 16, 3, 2, 64)]:
        E = H * d
        qkv = rnd(B * T * N, 3 * E, seed=d)
        tbl = rnd(1023, H, scale=0.5, seed=3)
        qd = qkv.to(dev)
        got = ops.temporal_attention(qd[:, :E], qd[:, E:2 * E], qd[:, 2 * E:], B, T, N, H, 1 / math.sqrt(d),
                                     tbl.to(dev), 512)
        x = qkv.float().view(B, T, N, 3, H, d).permute(3, 0, 2, 4, 1, 5)  # (3, b, n, h, t, d)
        pos = torch.arange(T)
        bias = tbl.float()[pos[None, :] - pos[:, None] + 511].permute(2, 0, 1)
        p = F.softmax(x[0] @ x[1].transpose(-1, -2) / math.sqrt(d) + bias[None, None], dim=-1)
        ref = (p @ x[2]).permute(0, 3, 1, 2, 4).reshape(B * T * N, E)  # (b, t, n, h, d)
        stats(f"temporal B{B} T{T} N{N} H{H} d{d}", got, ref)
    for (nb, S, H) in [(1, 64, 1), (2, 129, 3), (1, 513, 12), (1, 2049, 2), (3, 100, 12)]:
        Hd = H * 64
        qkv = rnd(nb, S, 3 * Hd, seed=S)
        got = ops.flash_attention_d64(qkv.to(dev), H, 0.125)
        x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
        p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
        ref = (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, Hd)
        stats(f"flash nb{nb} S{S} H{H}", got, ref)
    # spiky logits exercise the online-softmax rescale
    nb, S, H = 1, 300, 2
    qkv = rnd(nb, S, 3 * H * 64, scale=3.0, seed=5)
    got = ops.flash_attention_d64(qkv.to(dev), H, 0.125)
    x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    stats("flash spiky (scale 3)", got, (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, H * 64))


# ------------------------------------------------------------------------------------------- modules
def sec_modules():
    from cases import FULL_CASES, SPP_CASES, TOKENIZER_CASES, VIT_CASES, spp_inputs, tokenizer_inputs
    from helpers import load_golden, module_sd, tok_cfg
    from oracle import u2_oracle as O
    from u2tokenizer_amd.projector import SpatialPoolingProjector
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    from u2tokenizer_amd.vit import ViT3DTower
    print("[modules]", flush=True)
    for name, c in TOKENIZER_CASES.items():
        m = u2Tokenizer(embed_size=c["E"], num_heads=c["heads"], num_layers=c["layers"], top_k=c["top_k"],
                        use_multi_scale=c["use_multi_scale"], num_3d_query_token=c["Q"], hidden_size=c["E"],
                        attn_type=c["attn_type"], enable_diffts=c["enable_diffts"], enable_dmtp=c["enable_dmtp"])
        synth.fill_module_(m, seed=c["seed"], prefix="u2tokenizer.")
        m = m.to(bf).to(dev)
        v, t = tokenizer_inputs(c)
        got = m(v_token=v.to(bf).to(dev), t_token=t.to(bf).to(dev))
        g = load_golden(f"tokenizer_{name}")
        stats(f"tokenizer {name} vs reference fp32", got, g["out"])
        sd16 = module_sd(m, "u2tokenizer.", c["seed"], bf)
        o16, idx16 = O.tokenizer_forward(sd16, "u2tokenizer", v.to(bf), t.to(bf), tok_cfg(c))
        stats(f"tokenizer {name} vs oracle bf16", got, o16)
        stats(f"   (oracle bf16 vs reference fp32)", o16, g["out"])
        if not c["enable_diffts"]:
            print(f"   topk idx == oracle(bf16 inputs): {torch.equal(m.last_topk_indices.cpu(), idx16)}; "
                  f"== reference fp32: {torch.equal(m.last_topk_indices.cpu(), g['ref_topk_idx'])}", flush=True)
    for name, c in SPP_CASES.items():
        m = SpatialPoolingProjector(c["image_size"], c["patch_size"], c["in_dim"], c["E"], c["layer_type"],
                                    c["layer_num"], c["pooling_type"], c["pooling_size"])
        synth.fill_module_(m, seed=c["seed"], prefix="mm_projector.")
        got = m.to(bf).to(dev)(spp_inputs(c).to(bf).to(dev))
        stats(f"spp {name} vs reference fp32", got, load_golden(f"spp_{name}")["out"])
    for flash in (1, 0):
        ops.set_option("vit_flash", flash)
        for name, c in VIT_CASES.items():
            m = ViT3DTower(NS(vision_select_layer=-1, vision_select_feature=c["select_feature"], image_channel=1,
                              image_size=c["image_size"], patch_size=c["patch_size"]))
            synth.fill_module_(m, seed=c["seed"], prefix="vision_tower.")
            vol = synth.synth_volume(1, c["nchunk"], c["image_size"], seed=c["seed"], dtype=torch.float16)
            got = m.to(bf).to(dev)(vol.view(c["nchunk"], 1, *c["image_size"]).to(dev))
            stats(f"vit {name} flash={flash} vs reference fp32", got, load_golden(f"vit_{name}")["out"])
    ops.set_option("vit_flash", 1)
    from test_oracle_golden import _full_model, full_path_cfg
    for name, c in FULL_CASES.items():
        m, cfg = _full_model(c)
        g = load_golden(f"full_{name}")
        m = m.to(bf).to(dev)
        vol = synth.synth_volume(c["B"], c["C"], c["mm"]["image_size"], seed=c["seed"], dtype=torch.float16)
        ids = synth.synth_ids(c["B"], c["S"], c["n_real"], cfg.vocab_size, seed=c["seed"], name="input_ids")
        qids = synth.synth_ids(c["B"], c["Lt"], c["n_q"], cfg.vocab_size, seed=c["seed"], name="question_ids")
        r = m.prepare_inputs_for_multimodal(ids.to(dev), None, None, None, None, vol.to(dev), qids.to(dev))
        stats(f"full {name} inputs_embeds vs reference fp32", r[4], g["inputs_embeds"])
        out = m(images=vol.to(dev), input_ids=ids.to(dev), question_ids=qids.to(dev))
        stats(f"full {name} last logits vs reference fp32", out.logits[:, -1], g["logits_last"])
        gen = m.generate(vol.to(dev), ids.to(dev), question_ids=qids.to(dev), max_new_tokens=c["new_tokens"],
                         do_sample=False)
        print(f"   greedy ids {gen.cpu().tolist()} reference {g['greedy_ids'].tolist()} "
              f"equal={torch.equal(gen.cpu(), g['greedy_ids'])}", flush=True)
        print(f"   topk idx {m.model.u2tokenizer.last_topk_indices.cpu().tolist()}", flush=True)



# ------------------------------------------------------------------------------------------- ping-pong GEMM
def _pp_cases():
    return [  # (M, N, K, kwargs)
        (256, 256, 64, {}), (256, 192, 128, {}), (512, 384, 256, {}), (300, 200, 136, {}), (77, 520, 72, {}),
        (1000, 768, 1024, {}), (2049, 2304, 768, {}), (515, 264, 200, dict(bias=True, gelu=True)),
        (515, 264, 200, dict(bias=True, residual=True)), (515, 264, 200, dict(residual=True)),
        (515, 264, 200, dict(bias=True, out_f32=True, alpha=0.5)), (515, 264, 200, dict(out_f32=True)),
        (8192, 3072, 256, dict(bias=True)), (4096, 1536, 768, dict(bias=True, residual=True)),
        (16392, 768, 768, dict(bias=True, residual=True)),
    ]


def sec_ppc(v=None):
    """correctness + repeatability of one forced big-tile variant (python tools/gpu_check.py ppc:<20|21>)"""
    vs = [int(x) for x in str(v).split(",")] if v else [20, 21]
    for v in vs:
        ops.set_option("gemm_big", v)
        print(f"[pp correctness] variant {v}", flush=True)
        for (M, N, K, kw) in _pp_cases():
            a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
            bias = rnd(N, seed=3) if kw.get("bias") else None
            res = rnd(M, N, seed=4) if kw.get("residual") else None
            ref = kw.get("alpha", 1.0) * (a.float() @ b.float().t())
            if bias is not None:
                ref = ref + bias.float()
            if kw.get("gelu"):
                ref = F.gelu(ref)
            if res is not None:
                ref = ref + res.float()
            args = dict(bias=None if bias is None else bias.to(dev), residual=None if res is None else res.to(dev),
                        gelu=bool(kw.get("gelu")), out_f32=bool(kw.get("out_f32")), alpha=kw.get("alpha", 1.0))
            ad, bd = a.to(dev), b.to(dev)
            outs = [ops.gemm(ad, bd, **args).clone() for _ in range(4)]
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            stats(f"v{v} {M}x{N}x{K} {sorted(kw)}", outs[0], ref, extra=f"repeatable={same}")
        # batched (z-major tiles)
        a3, b3 = rnd(6, 300, 64, seed=10), rnd(6, 96, 64, seed=11)
        got = ops.gemm(a3.to(dev), b3.to(dev), out_f32=True)
        stats(f"v{v} batched 6x(300x96x64)", got, torch.einsum("zmk,znk->zmn", a3.float(), b3.float()))
    ops.set_option("gemm_big", 0)


PP_PERF_VARIANTS = (-1, 20, 21)


def sec_coldperf():
    """GEMMs with COLD weights (16 different weight matrices in rotation: 0.5-1.6 GB, beyond the 256 MB Infinity Cache),
    the way the pipeline sees them at batch 1: us per call, by kernel choice"""
    scratch = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    ops.set_gemm_scratch(scratch)
    for (M, N, K) in [(256, 4096, 4096), (256, 12288, 4096), (2048, 4096, 4096), (2048, 12288, 4096), (1792, 8192, 4096)]:
        nw = 16
        a, bias = rnd(M, K, seed=1).to(dev), rnd(N, seed=3).to(dev)
        ws = [rnd(N, K, seed=10 + i).to(dev) for i in range(nw)]
        out = torch.empty((1, M, N), dtype=bf, device=dev)
        line = f"  {M:5d}x{N:5d}x{K:4d} bias, cold weights"
        wt = [ops.pack_ktile_major(w) for w in ws]
        cfgs = [("classic", -1, -1, 0), ("classic split 4", -1, 4, 0), ("heuristic", 0, 0, 0), ("K-tile-major classic", -1, -1, 1),
                ("K-tile-major split 4", -1, 4, 1), ("K-tile-major split 8", -1, 8, 1)]
        for name, pp, sk, kt in cfgs:
            ops.set_option("gemm_big", pp)
            ops.set_option("gemm_splitk", sk)
            it = [0]

            def f():
                if kt:
                    ops.gemm(a, wt[it[0] % nw], bias=bias, out=out[0], b_ktile=True)
                else:
                    ops.gemm(a, ws[it[0] % nw], bias=bias, out=out)
                it[0] += 1
            ms = timeit(f, iters=32, warm=4)
            line += f" | {name}: {ms * 1e3:6.1f}"
        print(line, flush=True)
        del ws, wt
    ops.set_option("gemm_big", 0)
    ops.set_option("gemm_splitk", 0)
    ops.set_gemm_scratch(None)


def sec_skperf():
    """split-K on the skinny linear layers of the tokenizer (M = 256 queries)"""
    scratch = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    ops.set_gemm_scratch(scratch)
    for (M, N, K) in [(256, 4096, 4096), (256, 12288, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (256, 2048, 4096)]:
        a, b, bias = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2).to(dev), rnd(N, seed=3).to(dev)
        out = torch.empty((1, M, N), dtype=bf, device=dev)
        line = f"  {M:5d}x{N:5d}x{K:4d} bias"
        for sk in (-1, 0, 2, 4, 8):
            ops.set_option("gemm_splitk", sk)
            ms = timeit(lambda: ops.gemm(a, b, bias=bias, out=out), iters=10, warm=2)
            line += f" | split {sk:2d}: {ms * 1e3:6.1f} us {2 * M * N * K / ms / 1e9:5.0f} TF"
        print(line, flush=True)
    ops.set_option("gemm_splitk", 0)
    ops.set_gemm_scratch(None)


def sec_ppperf(shapes_sel=None):
    print("[pp perf] (random normal operands; us and TF/s; c = classic 128^2/64^2 kernel, a = heuristic)", flush=True)
    shapes = [(16384, 2304, 768, {}), (16392, 2304, 768, {}), (16384, 768, 768, dict(bias=True, residual=True)),
              (16384, 3072, 768, dict(bias=True, gelu=True)), (16384, 3072, 768, dict(bias=True)),
              (16384, 768, 3072, dict(bias=True, residual=True)), (16384, 768, 1024, dict(bias=True, residual=True)),
              (2048, 4096, 4096, dict(bias=True)), (2048, 12288, 4096, dict(bias=True)), (1792, 8192, 4096, dict(bias=True)),
              (1024, 8192, 4096, dict(bias=True)), (1024, 4096, 4096, dict(bias=True)), (256, 4096, 4096, dict(bias=True)),
              (256, 12288, 4096, dict(bias=True)), (2048, 1024, 4096, {}), (4096, 4096, 4096, {}), (8192, 8192, 8192, {})]
    if shapes_sel == "few":
        shapes = [shapes[0], shapes[3], shapes[5], shapes[7], shapes[8], shapes[15], shapes[16]]
    for (M, N, K, kw) in shapes:
        a, b = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2).to(dev)
        bias = rnd(N, seed=3).to(dev) if kw.get("bias") else None
        res = rnd(M, N, seed=4).to(dev) if kw.get("residual") else None
        out = torch.empty((1, M, N), dtype=bf, device=dev)
        line = f"  {M:5d}x{N:5d}x{K:4d} {'+'.join(sorted(kw)) or '-':18s}"
        for v in PP_PERF_VARIANTS:
            ops.set_option("gemm_big", v)
            try:
                ms = timeit(lambda: ops.gemm(a, b, bias=bias, residual=res, gelu=bool(kw.get("gelu")), out=out), iters=8, warm=2)
                line += f" | {'c' if v < 0 else ('a' if v == 0 else v)} {ms * 1e3:6.1f} {2 * M * N * K / ms / 1e9:5.0f}"
            except Exception as e:  # noqa: BLE001
                line += f" | {v} ERR"
        print(line, flush=True)
    ops.set_option("gemm_big", 0)


def sec_geluperf():
    """fc1 of the ViT (bias + GELU): small-tile kernel vs the big-tile kernel with the GELU epilogue"""
    M, N, K = 16384, 3072, 768
    a, b, bias = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2).to(dev), rnd(N, seed=3).to(dev)
    out = torch.empty((1, M, N), dtype=bf, device=dev)
    for name, big, bg in (("128^2 kernel", -1, 0), ("heuristic", 0, 0), ("big tile + GELU (heuristic pick)", 0, 1),
                          ("big tile 256x256 + GELU", 20, 1), ("big tile 256x192 + GELU", 21, 1)):
        ops.set_option("gemm_big", big)
        ops.set_option("gemm_big_gelu", bg)
        ms = timeit(lambda: ops.gemm(a, b, bias=bias, gelu=True, out=out), iters=10, warm=3)
        ms2 = timeit(lambda: ops.gemm(a, b, bias=bias, out=out), iters=10, warm=3)
        print(f"  fc1 {M}x{N}x{K} {name:34s}: gelu {ms * 1e3:7.1f} us ({2 * M * N * K / ms / 1e9:5.0f} TF/s)   bias only "
              f"{ms2 * 1e3:7.1f} us", flush=True)
    ops.set_option("gemm_big", 0)
    ops.set_option("gemm_big_gelu", 1)


def sec_flashperf():
    print("[flash perf] nb=8 H=12 (the ViT-B block at 256^3): us per launch incl. the V transpose, TF/s, MFMA util of 2.5 PF", flush=True)
    for (nb, S, H, extra) in [(8, 2049, 12, True), (8, 2049, 12, False), (8, 2048, 12, False), (16, 513, 12, True)]:
        qkv = rnd(nb, S, 3 * H * 64, seed=3).to(dev)
        fl = 4 * nb * H * S * S * 64
        for mode, split in ((7, 0), (7, 1)):   # (mode, flash_q_prescaled: the loop without its per-score multiply)
            ops.set_option("flash_mode", mode)
            ops.set_option("flash_q_prescaled", split)
            ms = timeit(lambda: ops.flash_attention_d64(qkv, H, 0.125, extra_last=extra), iters=10)
            # the transpose alone
            ops.set_option("flash_q_prescaled", 0)
            ref = _flash_ref(qkv[:1, :, :], H)
            got = ops.flash_attention_d64(qkv[:1].contiguous(), H, 0.125, extra_last=extra)
            err = (got.float() - ref).abs().max().item()
            print(f"  nb={nb} S={S} extra={int(extra)} mode={mode} q_prescaled={split}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s  "
                  f"util={fl / ms / 1e9 / 2500:.3f}  max_err={err:.2e}", flush=True)
    ops.set_option("flash_mode", 0)
    ops.set_option("flash_q_prescaled", 0)
    qkv = rnd(8, 2049, 3 * 768, seed=3).to(dev)
    vt = torch.empty((8, 768, 2048), dtype=bf, device=dev)
    ms = timeit(lambda: ops.transpose(qkv[:, :2048, 1536:].contiguous(), ld_out=2048, perm16=True), iters=10)
    print(f"  (V slice copy + transpose alone: {ms * 1e3:8.1f} us)", flush=True)


def _flash_ref(qkv, H):
    nb, S, _ = qkv.shape
    x = qkv.float().view(nb, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = F.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    return (p @ x[2]).permute(0, 2, 1, 3).reshape(nb, S, H * 64)


def flash_timeline(buf, split):
    """modes 7 / 8, TIMED build: wall-clock (100 MHz) stamps per wave: [0] entry, [1] before the KV loop, [2] after it,
    [3] exit; the first region of the buffer has the number of key tiles the wave walked in [7]"""
    n = 2048 * 4
    t = buf[65536:65536 + n * 8].view(-1, 8).cpu().double()
    tiles = buf[:n * 8].view(-1, 8).cpu().double()[:, 7]
    ok = t[:, 0] > 0
    t, tiles = t[ok], tiles[ok]
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    ent, pre, post, ex = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    main = t[:, 1] > 0
    print(f"  timeline (us from the first workgroup's entry; {int(main.sum())} main waves, {int((~main).sum())} extra-row waves)")
    tmax = tiles[main].max()
    groups = [("whole units, entry < 5 us", main & (tiles == tmax) & (ent < 5.0)), ("whole units, later", main & (tiles == tmax) & (ent >= 5.0)),
              ("half units", main & (tiles < tmax))]
    for name, sel in groups:
        if sel.sum() == 0:
            continue
        e_, p_, q_, x_ = ent[sel], pre[sel], post[sel], ex[sel]
        print(f"    {name}: {int(sel.sum())} waves | entry {e_.min():6.1f}..{e_.max():6.1f} | prologue {(p_ - e_).mean():5.2f} (max {(p_ - e_).max():5.2f}) | "
              f"KV loop {(q_ - p_).mean():6.2f} (min {(q_ - p_).min():6.2f} max {(q_ - p_).max():6.2f}) | epilogue {(x_ - q_).mean():5.2f} (max {(x_ - q_).max():5.2f}) | "
              f"exit {x_.min():6.1f}..{x_.max():6.1f}")
    # first-round whole units by XCD (workgroup w runs on XCD w % 8): is the spread of the loop times systematic?
    widx = torch.arange(n)[ok] // 4
    sel = groups[0][1]
    if sel.sum():
        loop = (post - pre)
        per_xcd = [loop[sel & ((widx & 7) == x)].mean().item() for x in range(8)]
        print("    first round, KV loop by XCD: " + "  ".join(f"{v:5.1f}" for v in per_xcd) +
              f" | spread inside a workgroup (max - min of its 4 waves), mean: "
              f"{(loop[sel].view(-1, 4).max(1).values - loop[sel].view(-1, 4).min(1).values).mean().item():4.2f}")
    if (~main).sum():
        e_, x_ = ent[~main], ex[~main]
        print(f"    extra-row workgroups: entry {e_.min():6.1f}..{e_.max():6.1f}, duration {(x_ - e_).mean():5.2f} (max {(x_ - e_).max():5.2f}), last exit {x_.max():6.1f}")
        tx = t[~main]
        if (tx[:, 4] > 0).all():
            sc, sm, pv = (tx[:, 4] - tx[:, 0]) / 100, (tx[:, 5] - tx[:, 4]) / 100, (tx[:, 3] - tx[:, 5]) / 100
            print(f"      of which scores {sc.mean():5.2f}  softmax {sm.mean():5.2f}  P V + store {pv.mean():5.2f} us")
    print(f"    kernel span by these stamps: {ex.max():6.1f} us")


def sec_flashtime():
    """s_memtime phase breakdown of the flash attention kernel (cycles per KV tile per wave)"""
    from u2tokenizer_amd import _lib
    h = _lib.load_library()
    nb, S, H = 8, 2049, 12
    qkv = rnd(nb, S, 3 * H * 64, seed=3).to(dev)
    names = ["gload", "QK^T", "softmax", "PV", "wait+lstore", "-", "barrier"]
    for mode, split in ((7, 0), (7, 1)):
        ops.set_option("flash_mode", mode)
        ops.set_option("flash_q_prescaled", split)
        ms0 = timeit(lambda: ops.flash_attention_d64(qkv, H, 0.125, extra_last=True), iters=5)
        buf = torch.zeros(65536 + 2048 * 4 * 8, dtype=torch.int64, device=dev)
        _lib.check(h.u2tok_flash_debug_buffer(buf.data_ptr()), "flash_debug_buffer")
        ops.flash_attention_d64(qkv, H, 0.125, extra_last=True)
        torch.cuda.synchronize()
        h.u2tok_flash_debug_buffer(None)
        r = buf.view(-1, 8).double()
        r = r[r[:, 7] > 0]
        per = r[:, :7].sum(0) / r[:, 7].sum()
        if mode == 7:
            flash_timeline(buf, split)
        if mode % 10 in (5, 7):
            print(f"  mode {mode} q_prescaled {split}: {ms0 * 1e3:7.1f} us untimed | per 64-key tile per wave (2 blocks): phases u=2t {per[0]:6.0f}  dma wait {per[1]:6.0f}  "
                  f"barrier {per[2]:6.0f}  dma issue {per[3]:6.0f}  phases u=2t+1 {per[4]:6.0f}  total {per[:5].sum():6.0f}", flush=True)
            continue
        if mode % 10 == 4:
            print(f"  mode {mode} q_prescaled {split}: {ms0 * 1e3:7.1f} us untimed | per KV tile per wave: V {per[0]:6.0f}  wait {per[1]:6.0f}  M {per[2]:6.0f}  "
                  f"wait {per[3]:6.0f}  total {per[:4].sum():6.0f}", flush=True)
            continue
        print(f"  mode {mode} q_prescaled {split}: {ms0 * 1e3:7.1f} us untimed | per KV tile per wave: " +
              "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, per.tolist()) if n != "-") + f"  total {per.sum():6.0f}", flush=True)
    ops.set_option("flash_mode", 0)
    ops.set_option("flash_q_prescaled", 0)


def sec_preperf():
    """u2tok_preprocess_volume (u2Transform.adaptive_resize on the GPU) vs the CPU oracle, 512 x 512 x 200 CT-like volume"""
    import numpy as np
    from oracle import u2_preprocess_oracle as P
    from u2tokenizer_amd.preprocess import u2Transform
    rng = np.random.default_rng(0)
    H, W, Dz = 512, 512, 200
    vol = np.full((H, W, Dz), -1024.0)
    vol[40:470, 60:450, 10:190] = rng.normal(40, 250, size=(430, 390, 180)).round()
    tr = u2Transform(device=dev, out_dtype=torch.float16)
    v = torch.as_tensor(vol).permute(2, 0, 1).to(device=dev, dtype=torch.float32).contiguous()
    ms = timeit(lambda: tr.from_dhw(v), iters=10, warm=2)
    nbytes = v.numel() * 4
    print(f"[preprocess] GPU {H}x{W}x{Dz} -> (8,32,256,256) fp16: {ms:7.3f} ms  ({nbytes / 1e6:.0f} MB in; "
          f"{9 * nbytes / ms / 1e9:.2f} TB/s over ~9 passes)", flush=True)
    t0 = time.time()
    ref, _ = P.adaptive_resize(vol)
    t1 = time.time() - t0
    got = tr.from_dhw(v).float().cpu()
    print(f"[preprocess] CPU oracle (numpy/torch, {torch.get_num_threads()} threads): {t1:6.2f} s;  max |gpu - oracle| = "
          f"{(got - ref).abs().max().item():.2e} (fp16 output)", flush=True)

# ------------------------------------------------------------------------------------------- perf
def sec_perf():
    print("[perf]", flush=True)
    shapes = [(16392, 2304, 768), (16392, 768, 768), (16392, 3072, 768), (16392, 768, 3072), (16384, 768, 1024),
              (2048, 4096, 4096), (2048, 2048, 2048), (256, 4096, 4096), (1792, 4096, 4096), (1024, 4096, 4096),
              (4096, 4096, 4096), (8192, 8192, 8192)]
    for bk in (64,):
        for (M, N, K) in shapes:
            a, b = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2).to(dev)
            out = torch.empty((1, M, N), dtype=bf, device=dev)
            for tile in ((64, 128) if M * N < 4096 * 4096 * 2 else (128,)):
                ops.set_option("gemm_tile", tile)
                ms = timeit(lambda: ops.gemm(a, b, out=out), iters=10)
                print(f"  gemm bk={bk} tile={tile:3d} {M}x{N}x{K}: {ms * 1e3:9.1f} us  {2 * M * N * K / ms / 1e9:8.1f} TF/s",
                      flush=True)
    ops.set_option("gemm_tile", 0)
    for (nb, S, H) in [(8, 2049, 12), (16, 513, 12)]:
        qkv = rnd(nb, S, 3 * H * 64, seed=3).to(dev)
        ms = timeit(lambda: ops.flash_attention_d64(qkv, H, 0.125), iters=10)
        print(f"  flash(+V transpose) nb={nb} S={S} H={H}: {ms * 1e3:9.1f} us  {4 * nb * H * S * S * 64 / ms / 1e9:8.1f} TF/s",
              flush=True)
    from u2tokenizer_amd.vit import ViT3DTower
    from u2tokenizer_amd.projector import SpatialPoolingProjector
    from u2tokenizer_amd.tokenizer import u2Tokenizer
    vit = ViT3DTower(NS(vision_select_layer=-1, vision_select_feature="patch", image_channel=1,
                        image_size=[32, 256, 256], patch_size=[4, 16, 16]))
    synth.fill_module_(vit, seed=0, prefix="vision_tower.")
    vit = vit.to(bf).to(dev)
    vol = synth.synth_volume(1, 8, [32, 256, 256], dtype=torch.float16).view(8, 1, 32, 256, 256).to(dev)
    for bk in (64,):
        ms = timeit(lambda: vit(vol), iters=5, warm=2)
        print(f"  ViT tower 256^3 (8 chunks) bk={bk}: {ms:8.3f} ms  {4.048e12 / ms / 1e9:8.1f} TF/s", flush=True)
    feats = vit(vol)
    for E in (2048, 4096):
        spp = SpatialPoolingProjector([32, 256, 256], [4, 16, 16], 768, E, "mlp", 2, "spatial", 2)
        synth.fill_module_(spp, seed=0, prefix="mm_projector.")
        spp = spp.to(bf).to(dev)
        ms = timeit(lambda: spp(feats), iters=5, warm=2)
        print(f"  SPP E={E}: {ms:8.3f} ms", flush=True)
        tok = u2Tokenizer(E, 8, 4, 1024, True, 256, E, "rma", True, True)
        for p in tok.parameters():
            p.data = p.data.to(bf)
        tok = tok.to(dev)
        for k_, p in tok.named_parameters():
            if "relative_bias" in k_:
                p.data.normal_(0, 0.02)
        v = spp(feats).view(1, 8, 256, E)
        t = (torch.randn(1, 1024, E, device=dev) * 0.05).to(bf)
        for bk in (64,):
            ms = timeit(lambda: tok(v_token=v, t_token=t), iters=5, warm=2)
            fl = {2048: 0.888e12, 4096: 3.43e12}[E]
            print(f"  u2Tokenizer E={E} bk={bk}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s  "
                  f"finite={torch.isfinite(tok(v_token=v, t_token=t).float()).all().item()}", flush=True)


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0), flush=True)
    ops.device_check()
    t0 = time.time()
    for s in sys.argv[1:]:
        name, _, arg = s.partition(":")
        globals()["sec_" + name](*([arg] if arg else []))
    print(f"done in {time.time() - t0:.1f}s", flush=True)
