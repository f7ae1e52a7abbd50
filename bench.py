#!/usr/bin/env python
"""bench.py -- 256^3 fp16 CT volumes/s through the u2Tokenizer forward path on N MI355X (BASELINE.json metric).

One step = one pass of the hot path (reference u2_arch.py:96-117) over one batch of synthetic input already resident
in HBM: fp16 volume (B,8,32,256,256) -> im2col/patch-embed -> ViT-B 3D x12 -> SPP -> u2Tokenizer (SVR x4, DiffTS,
DMTP multi-scale, TTA x4, linear aggregation) -> 256 tokens spliced into the (B,1024,E) prompt embeddings.
Workload = BASELINE.json configs[2] (u2Qwen3-8B shape: E=4096, full multi-scale tokenizer, batch 1), random-init
weights of that architecture, synthetic data.  N > 1: independent replicas, one process per GPU (weak scaling, no
data-path collective); launched by torch.distributed.run, timed with barrier + synchronize, MAX over ranks.

Steps are issued round-robin on --streams HIP streams (default 2: two batch-1 volumes in flight per GPU; every step is
still one complete pass over one volume, and `value_one_stream` reports the same K steps on a single stream).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel class (bf16 MFMA GEMM): algorithmic FLOPs of all its launches in one step /
                  their summed HIP-event durations (instrumented pass after the timed region), vs 2.5 PFLOP/s.
  cpu_baseline -- the CPU oracle (oracle/u2_oracle.py, fp32, all host cores) timed on a bounded sample of the
                  same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
BF = torch.bfloat16


def flops_per_volume(E, Lt=1024, C=8, n=2048, Hd=768, mlp=3072, depth=12, L=4, Q=256, k=1024, N=256):
    """2*M*N*K of every GEMM and both attention products."""
    S = n + 1
    vit = 2 * C * n * 1024 * Hd + depth * (2 * C * S * (3 * Hd * Hd + Hd * Hd + 2 * Hd * mlp) + 4 * C * S * S * Hd)
    spp = 2 * C * N * (Hd * E + E * E)

    def self_(bt, s):
        return 2 * bt * s * E * E * 4 + 4 * bt * s * s * E

    def cross(sq, skv, pv, do):
        return 2 * E * E * (sq + skv * (1 + pv) + do * sq) + 4 * sq * skv * E

    V = k + k // 2 + k // 4
    svr = L * (self_(C, N) + self_(N, C))
    diffts = 2 * C * N * E * k + 2 * k * C * N * E
    tta = L * (self_(1, Q) + cross(Q, V, 1, 1) + cross(Q, Lt, 1, 1)) + cross(Q, V, 0, 0)
    return dict(vit=vit, spp=spp, svr=svr, select=diffts, tta=tta, total=vit + spp + svr + diffts + tta)


def build_path(E, vocab, device):
    """ViT3DTower + SPP + u2Tokenizer + embedding table, random-init on the GPU (no decoder: outside the path)."""
    from types import SimpleNamespace as NS
    from u2tokenizer_amd.arch import u2MetaForCausalLM
    from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower

    cfg = NS(vision_tower="vit3d", image_channel=1, image_size=[32, 256, 256], patch_size=[4, 16, 16],
             vision_select_layer=-1, vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp",
             proj_layer_num=2, proj_pooling_type="spatial", proj_pooling_size=2, mm_hidden_size=768, hidden_size=E,
             enable_u2tokenizer=True, u2t_num_heads=8, u2t_num_layers=4, u2t_top_k=1024, use_multi_scale=True,
             num_3d_query_token=256, attn_type="rma", enable_diffts=True, enable_dmtp=True)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vision_tower = build_vision_tower(cfg)
            self.mm_projector = build_mm_projector(cfg)
            self.u2tokenizer = build_u2tokenizer_tower(cfg)
            self.embed_tokens = torch.nn.Embedding(vocab, E)

        def get_vision_tower(self):
            return self.vision_tower

        def get_u2tokenizer(self):
            return self.u2tokenizer

    class PathOnly(u2MetaForCausalLM):
        def __init__(self, holder):
            self.holder, self.config = holder, cfg

        def get_model(self):
            return self.holder

    with torch.device("meta"):
        holder = Holder()
    holder = holder.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(0)
    for name, p in holder.named_parameters():
        t = torch.empty(p.shape, dtype=BF, device=device)
        if p.dim() == 2 and "relative_bias" not in name and "embed_tokens" not in name:
            t.normal_(0, 1.0 / p.shape[1] ** 0.5, generator=g)
        elif "norm" in name and name.endswith("weight"):
            t.fill_(1.0)
        elif "query_tokens" in name:
            t.normal_(0, 0.5, generator=g)
        else:
            t.normal_(0, 0.02, generator=g)
        p.data = t
        p.requires_grad_(False)
    return PathOnly(holder), cfg


def cpu_baseline(E, Lt, sample_chunks=1):
    """Oracle (fp32 CPU restatement of the reference) on a bounded sample of the workload: `sample_chunks` of the 8
    chunks through ViT+SPP (extrapolated x8/sample_chunks) + the full tokenizer + splice.  Weight VALUES do not
    affect CPU time, so same-shape tensors share storage (keeps host RAM/initialisation bounded)."""
    from oracle import u2_oracle as O
    from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower
    from types import SimpleNamespace as NS
    cfgm = NS(vision_tower="vit3d", image_channel=1, image_size=[32, 256, 256], patch_size=[4, 16, 16],
              vision_select_layer=-1, vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp",
              proj_layer_num=2, proj_pooling_type="spatial", proj_pooling_size=2, mm_hidden_size=768, hidden_size=E,
              u2t_num_heads=8, u2t_num_layers=4, u2t_top_k=1024, use_multi_scale=True, num_3d_query_token=256,
              attn_type="rma", enable_diffts=True, enable_dmtp=True)
    with torch.device("meta"):
        mods = {"model.vision_tower.": build_vision_tower(cfgm), "model.mm_projector.": build_mm_projector(cfgm),
                "model.u2tokenizer.": build_u2tokenizer_tower(cfgm)}
    pool, sd = {}, {}
    gen = torch.Generator().manual_seed(0)
    for prefix, m in mods.items():
        for k, v in m.state_dict().items():
            shp = tuple(v.shape)
            if shp not in pool:
                std = 1.0 / shp[1] ** 0.5 if len(shp) == 2 and shp[0] > 8 else 0.02
                pool[shp] = torch.randn(shp, generator=gen) * std
                if "norm" in k and k.endswith("weight"):
                    pool[shp] = torch.ones(shp)
            sd[prefix + k] = pool[shp]
    cfg = O.PathConfig(hidden_size=E)
    threads = torch.get_num_threads()
    with torch.no_grad():
        vol = torch.rand(sample_chunks, 1, 32, 256, 256, generator=gen)
        t0 = time.perf_counter()
        feats = O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol, cfg)
        feats = O.spp_forward(sd, "model.mm_projector", feats, cfg)
        t_vis = (time.perf_counter() - t0) * (8.0 / sample_chunks)
        v = torch.randn(1, 8, 256, E, generator=gen)
        t = torch.randn(1, Lt, E, generator=gen) * 0.05
        t0 = time.perf_counter()
        out, _ = O.tokenizer_forward(sd, "model.u2tokenizer", v, t, cfg)
        t_tok = time.perf_counter() - t0
    return dict(value=1.0 / (t_vis + t_tok), unit="volumes/s", cores=threads, kind="port",
                sample=f"oracle fp32: {sample_chunks}/8 chunks through ViT+SPP ({t_vis:.1f} s extrapolated to 8) + "
                       f"full tokenizer E={E} Lt={Lt} ({t_tok:.1f} s); one run, no warm-up",
                seconds_per_volume=t_vis + t_tok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=4096, help="LLM hidden size E (4096 = Qwen3-8B, 2048 = Qwen3-1.7B)")
    ap.add_argument("--batch", type=int, default=1, help="volumes per step per GPU")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (2 = two volumes in flight per GPU: the small "
                         "launches of one volume's tokenizer fill the machine under the other volume's large GEMMs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from u2tokenizer_amd import replicas
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # RCCL ("nccl" backend on ROCm) is used only for the timing barrier and the MAX over ranks: replicas share nothing
    dist, rank, world = replicas.init_from_env("nccl", device)

    from u2tokenizer_amd import _lib, ops
    ops.device_check()  # fails loudly off gfx950 / without the HIP library
    torch.set_grad_enabled(False)

    E, B, S, Lt, Q = args.hidden, args.batch, 1024, 1024, 256
    vocab = 151936 if E == 4096 else 151936  # Qwen3 vocabulary
    path, cfg = build_path(E, vocab, device)
    g = torch.Generator(device=device).manual_seed(1 + rank)
    nvol = 4  # rotate volumes so no step re-reads its input from the 256 MiB Infinity Cache
    vols = [torch.rand((B, 8, 32, 256, 256), device=device, generator=g).half() for _ in range(nvol)]
    for v in vols:
        v.view(B, 256, 256, 256)[:, 205:] = 0  # trailing depth padding (u2Transform.py:93-94)
    ids = torch.randint(1, vocab, (B, S), device=device, generator=g)
    qids = torch.zeros((B, Lt), dtype=torch.int64, device=device)
    qids[:, :40] = torch.randint(1, vocab, (B, 40), device=device, generator=g)

    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else None

    def step(i, multi=True):
        if streams is None or not multi:
            return path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % nvol], qids)[4]
        with torch.cuda.stream(streams[i % len(streams)]):
            return path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % nvol], qids)[4]

    def sync():
        replicas.barrier(dist, device)

    for i in range(args.warmup):
        out = step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    sync()
    elapsed = time.perf_counter() - t0
    assert out.shape == (B, S, E) and bool(torch.isfinite(out.float()).all())
    elapsed = replicas.max_over_ranks(dist, elapsed, device)

    # the same K steps issued on ONE stream (one volume in flight), for reference next to the headline
    single = None
    if streams is not None:
        for i in range(args.warmup):
            step(i, multi=False)
        sync()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i, multi=False)
        sync()
        single = replicas.max_over_ranks(dist, time.perf_counter() - t1, device)

    fl = flops_per_volume(E, Lt)
    value = world * B * args.steps / elapsed
    line = {
        "metric": "CT volumes/sec (256^3 fp16) through u2Tokenizer fwd", "value": round(value, 3),
        "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "tokens_per_s": round(value * Q, 1),
        "value_one_stream": round(world * B * args.steps / single, 3) if single else None,
        "path_tflops": round(value * fl["total"] / 1e12, 1),
        "path_frac_of_bf16_mfma_peak": round(value * fl["total"] / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "config": {"workload": "BASELINE configs[2]: u2Qwen3-8B-shaped path, 256^3 volume = 8x(32,256,256) fp16, "
                               "ViT-B 3D x12, SPP, 4-layer rma+diffts+dmtp tokenizer (8 heads, top_k 1024, scales "
                               "{1,2,4}, 256 queries), text 1024, prompt 1024",
                   "hidden_size": E, "batch_per_gpu": B, "streams_per_gpu": args.streams, "flop_per_volume": fl["total"],
                   "parallelism": f"replicas x{world}"},
    }

    if rank == 0 and not args.no_roofline:
        h = _lib.load_library()
        import ctypes as C
        ops.set_option("profile", 1)
        nprof = 3
        for i in range(nprof):
            step(i, multi=False)  # one stream: a kernel's own duration, not stretched by a co-running volume
        torch.cuda.synchronize()
        ms, flops, byts, cnt = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_double * 5)(), (C.c_int64 * 5)()
        _lib.check(h.u2tok_profile_collect2(ms, flops, byts, cnt, 5), "u2tok_profile_collect2")
        ops.set_option("profile", 0)
        names = ["gemm_bf16 (gemm_bt_kernel + gemm_bf16_nt_kernel + gemm_splitk_reduce_kernel)", "flash_d64 (flash_dp_kernel)", "temporal_attention_kernel", "row_ops",
                 "data_movement"]
        # per class: time, launches, algorithmic TFLOP/s and algorithmic GB/s (operands + results once) of its launches
        classes = {n: {"ms_per_step": round(ms[i] / nprof, 4), "launches_per_step": cnt[i] // nprof,
                       "tflops": round(flops[i] / ms[i] / 1e9, 1) if ms[i] > 0 and flops[i] > 0 else None,
                       "algorithmic_gbytes_per_s": round(byts[i] / ms[i] / 1e6, 1) if ms[i] > 0 and byts[i] > 0 else None}
                   for i, n in enumerate(names)}
        achieved = flops[0] / ms[0] / 1e9
        # HBM-side bytes of the same kernel class come from rocprofv3 PMC passes of THIS command (a process cannot
        # read its own counters): tools/gpu_round.sh -> tools/pmc_traffic.py -> profiles/r01_traffic.json
        traffic, traffic_src = None, None
        tfile = ROOT / "profiles" / "r01_traffic.json"
        if tfile.exists() and E == 4096 and B == 1:
            tj = json.loads(tfile.read_text())["kernels"].get("gemm_bf16")
            if tj:
                # per GEMM call as counted here (a call = its main kernel + the 128^2 launch of its row tail / the
                # split-K reduce where used: 198 dispatches for 137 calls), so that it compares with the algorithmic bytes
                traffic = round(tj["hbm_bytes_per_step"] / max(cnt[0] // nprof, 1))
                traffic_src = ("profiles/r01_traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per volume / GEMM calls "
                               "per volume, rocprofv3 --pmc, separate passes; fabric-side requests (Infinity Cache hits "
                               "included)")
        line["roofline"] = {"bound": "mfma", "kernel": "bf16 MFMA GEMM, all launches of one step (gemm_bt_kernel 256x256/256x192 tiles, gemm_bf16_nt_kernel 128^2/64^2 tiles, gemm_splitk_reduce_kernel)",
                            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                            "traffic_unit": "bytes per launch (average)", "traffic_source": traffic_src,
                            "algorithmic_bytes_per_launch": round(byts[0] / max(cnt[0], 1)),
                            "avg_launch_us": round(1e3 * ms[0] / cnt[0], 2),
                            "flop_per_step": flops[0] / nprof, "classes": classes}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(E, Lt)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
