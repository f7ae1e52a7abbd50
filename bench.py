#!/usr/bin/env python
"""bench.py -- 256^3 fp16 CT volumes/s through the u2Tokenizer forward path on N MI355X (BASELINE.json metric).

One step = one pass of the hot path (reference u2_arch.py:96-117) over one batch of synthetic input already resident
in HBM: fp16 volume (B,8,32,256,256) -> im2col/patch-embed -> ViT-B 3D x12 -> SPP -> u2Tokenizer (SVR x4, DiffTS,
DMTP multi-scale, TTA x4, linear aggregation) -> 256 tokens spliced into the (B,1024,E) prompt embeddings.
Workload = BASELINE.json configs[2] (u2Qwen3-8B shape: E=4096, full multi-scale tokenizer, batch 1), random-init
weights of that architecture, synthetic data.

N > 1: independent replicas, one process per GPU (weak scaling, no data-path collective: SURVEY.md 8e), timed with
barrier + synchronize on both sides, MAX over ranks.  `python bench.py --gpus N` launches its N ranks itself
(re-executes under torch.distributed.run on 127.0.0.1); started under torch.distributed.run it uses the ranks it is
given.

The timed region (EXACTLY --steps steps between barrier + synchronize) is repeated --repeats times; `value` is the
MEDIAN repeat, all repeats are listed under "repeats".  Steps are issued round-robin on --streams HIP streams (default 2:
two batch-1 volumes in flight per GPU; every step is still one complete pass over one volume; `ms_per_step` is
elapsed / steps, i.e. the reciprocal of the throughput, and `ms_per_step_one_stream` the latency of one volume alone).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline            -- the dominant kernel class (bf16 MFMA GEMM): algorithmic FLOPs of its launches in one step /
                         their summed HIP-event durations on the launch stream (instrumented one-stream pass after the
                         timed region; the timed region itself carries no events), vs 2.5 PFLOP/s.
  roofline_attention  -- the same for the ViT flash-attention kernel (north_star's ">= 40 % MFMA" target).
  cpu_baseline        -- the CPU oracle (oracle/u2_oracle.py) on the host cores: fp32 and bf16, 1 warm-up + 3 timed
                         iterations per stage on a bounded sample (1 of 8 chunks through ViT + SPP, 1 of 4 layers of
                         SVR and TTA, selection / pooling / aggregation in full), rank 0, N = 1 only.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
BF = torch.bfloat16
WORKLOAD = ("BASELINE configs[2]: u2Qwen3-8B-shaped path, 256^3 volume = 8x(32,256,256) fp16, ViT-B 3D x12, SPP, "
            "4-layer rma+diffts+dmtp tokenizer (8 heads, top_k 1024, scales {1,2,4}, 256 queries), text 1024, prompt 1024")


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


def path_config(E):
    from types import SimpleNamespace as NS
    return NS(vision_tower="vit3d", image_channel=1, image_size=[32, 256, 256], patch_size=[4, 16, 16],
              vision_select_layer=-1, vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp",
              proj_layer_num=2, proj_pooling_type="spatial", proj_pooling_size=2, mm_hidden_size=768, hidden_size=E,
              enable_u2tokenizer=True, u2t_num_heads=8, u2t_num_layers=4, u2t_top_k=1024, use_multi_scale=True,
              num_3d_query_token=256, attn_type="rma", enable_diffts=True, enable_dmtp=True)


def build_path(E, vocab, device):
    """ViT3DTower + SPP + u2Tokenizer + embedding table, random-init on the GPU (no decoder: outside the path)."""
    from u2tokenizer_amd.arch import u2MetaForCausalLM
    from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower

    cfg = path_config(E)

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
    holder.u2tokenizer.pack_weights()  # q|k|v packing happens here, not inside the first timed / warm-up step
    return PathOnly(holder), cfg


# ---------------------------------------------------------------------------------------------------- CPU baseline
def _median_time(fn, iters):
    fn()  # warm-up
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline(E, Lt, iters=3, sample_chunks=1):
    """Oracle (CPU restatement of the reference, oracle/u2_oracle.py) per BASELINE.md section 3: torch.no_grad(), all
    host threads, fp32 AND bf16, 1 warm-up + `iters` timed iterations per stage (median), per-stage split.  Bounded by
    SAMPLING the workload, not by dropping repeats: `sample_chunks` of the 8 chunks go through ViT + SPP (chunks are
    independent: x 8 / sample_chunks), ONE of the 4 identical SVR layers and ONE of the 4 identical TTA layers are
    timed (x 4); selection, pooling and the final aggregation run in full.  Weight VALUES do not affect CPU time, so
    same-shape tensors share storage (keeps host RAM / initialisation bounded)."""
    from oracle import u2_oracle as O
    from u2tokenizer_amd.builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower
    cfgm = path_config(E)
    with torch.device("meta"):
        mods = {"model.vision_tower.": build_vision_tower(cfgm), "model.mm_projector.": build_mm_projector(cfgm),
                "model.u2tokenizer.": build_u2tokenizer_tower(cfgm)}
    pool, sd32 = {}, {}
    gen = torch.Generator().manual_seed(0)
    for prefix, m in mods.items():
        for k, v in m.state_dict().items():
            shp = tuple(v.shape)
            if shp not in pool:
                std = 1.0 / shp[1] ** 0.5 if len(shp) == 2 and shp[0] > 8 else 0.02
                pool[shp] = torch.randn(shp, generator=gen) * std
                if "norm" in k and k.endswith("weight"):
                    pool[shp] = torch.ones(shp)
            sd32[prefix + k] = pool[shp]
    pool16 = {id(v): v.to(BF) for v in pool.values()}
    sd16 = {k: pool16[id(v)] for k, v in sd32.items()}
    cfg = O.PathConfig(hidden_size=E)
    cfg1 = O.PathConfig(hidden_size=E, u2t_num_layers=1)
    threads = torch.get_num_threads()
    tp = "model.u2tokenizer"
    res = {}
    with torch.no_grad():
        vol = torch.rand(sample_chunks, 1, 32, 256, 256, generator=gen)
        v = torch.randn(1, 8, 256, E, generator=gen)
        t = torch.randn(1, Lt, E, generator=gen) * 0.05
        for label, sd, dt in (("fp32", sd32, torch.float32), ("bf16", sd16, BF)):
            vol_d, v_d, t_d = vol.to(dt), v.to(dt), t.to(dt)
            st = {}
            feats = O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol_d, cfg)
            st["vit"], _ = _median_time(lambda: O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol_d, cfg), iters)
            st["vit"] *= 8.0 / sample_chunks
            st["spp"], _ = _median_time(lambda: O.spp_forward(sd, "model.mm_projector", feats, cfg), iters)
            st["spp"] *= 8.0 / sample_chunks
            lp = f"{tp}.svt_module.attention_network.layers.0"
            x1 = O.st_attention_layer(sd, lp, v_d, cfg)
            t_layer, _ = _median_time(lambda: O.st_attention_layer(sd, lp, v_d, cfg), iters)
            st["svr"] = 4.0 * t_layer
            sel = O.diff_token_selection(sd, f"{tp}.svt_module.token_selection", x1)
            pooled = O.multi_scale_pool(sd, f"{tp}.svt_module.dynamic_pool", sel)
            st["select_pool"], _ = _median_time(
                lambda: O.multi_scale_pool(sd, f"{tp}.svt_module.dynamic_pool",
                                           O.diff_token_selection(sd, f"{tp}.svt_module.token_selection", x1)), iters)
            q = sd[f"{tp}.query_tokens"]
            t_agg, _ = _median_time(lambda: O.cross_attention(sd, f"{tp}.tta_module.layer_linagg.linear_aggregator", q,
                                                               pooled, 8, is_compress=True), iters)
            t_l1, _ = _median_time(lambda: O.tta_forward(sd, f"{tp}.tta_module", q, pooled, t_d, cfg1), iters)
            st["tta"] = 4.0 * max(t_l1 - t_agg, 0.0) + t_agg
            st["total"] = sum(st.values())
            res[label] = {k: round(x, 3) for k, x in st.items()}
    return dict(value=round(1.0 / res["fp32"]["total"], 5), unit="volumes/s", cores=threads, kind="port",
                value_bf16=round(1.0 / res["bf16"]["total"], 5),
                seconds_per_volume_fp32=res["fp32"], seconds_per_volume_bf16=res["bf16"],
                sample=f"oracle/u2_oracle.py on {threads} host threads, torch.no_grad, 1 warm-up + {iters} timed iterations "
                       f"(median) per stage; {sample_chunks}/8 chunks through ViT + SPP (x{8 // sample_chunks}), 1/4 SVR layers "
                       f"and 1/4 TTA layers (x4), DiffTS + DMTP pooling + linear aggregation in full; E={E}, text {Lt}")


# ---------------------------------------------------------------------------------------------------- HBM traffic (PMC)
PMC_CLASSES = [("gemm_bf16_nt_kernel", "gemm_bf16"), ("gemm_bt_kernel", "gemm_bf16"), ("gemm_bt_drain_kernel", "gemm_bf16"),
               ("gemm_skinny", "gemm_bf16"), ("gemm_splitk", "gemm_bf16"), ("gemm_rows16", "gemm_bf16"), ("flash_", "flash_d64"), ("tok_attn", "tok_attention")]
# kernel name fragment -> class of the per-kernel table (order matters: first match)
KERNEL_CLASSES = [("gemm_bf16_nt_kernel", "gemm_bf16"), ("gemm_bt_kernel", "gemm_bf16"), ("gemm_bt_drain_kernel", "gemm_bf16"), ("gemm_skinny", "gemm_bf16"),
                  ("gemm_splitk", "gemm_bf16"), ("gemm_rows16", "gemm_bf16"),
                  ("flash_", "flash_d64"), ("tok_attn", "tok_attention"), ("temporal_attention", "temporal_attention"),
                  ("layernorm", "row_ops"), ("softmax", "row_ops"), ("rope", "row_ops"), ("score_gemv", "row_ops"),
                  ("topk", "row_ops"), ("multiscale_pool", "row_ops"), ("dmtp_gate", "row_ops"), ("avgpool3d", "row_ops"),
                  ("im2col", "data_movement"), ("transpose", "data_movement"), ("gather_rows", "data_movement"),
                  ("embed_splice", "data_movement"), ("fill_rows", "data_movement"), ("copyBuffer", "data_movement")]


def _child_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    return env


def measure_kernels(E, options, timeout=300):
    """Per-kernel durations of the path from the profiler's own clock: ONE `rocprofv3 --kernel-trace --stats` child of THIS
    command (6 timed + 1 warm-up volume on one stream, no counters, no HIP events in the stream), parsed into
    {kernel name: (launches per volume, average microseconds)}.  Per-launch HIP events (the instrumented pass below) carry the
    launch gaps of ~250 short launches and overstate kernel time by >= 10 %; this is what `rocprofv3 --stats` of the same
    command reports, so the committed profiles/rNN_bench_kernel_stats.csv must agree with it.  None if rocprofv3 cannot run."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    nvol = 7
    tmp = tempfile.mkdtemp(prefix="u2tok_kt_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "k", "--",
               sys.executable, str(Path(__file__).resolve()), "--steps", "6", "--warmup", "1", "--repeats", "1",
               "--streams", "1", "--hidden", str(E), "--no-cpu-baseline", "--no-roofline", "--no-train-step"]
        for o in options:
            cmd += ["--option", o]
        r = subprocess.run(cmd, cwd="/tmp", env=_child_env(), capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            return None
        # the child's OWN wall time per volume for the very volumes the trace covers (its bench line: 6 timed steps, one stream)
        child_wall_ms = None
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                try:
                    child_wall_ms = float(json.loads(ln)["ms_per_step"])
                except (ValueError, KeyError):
                    pass
                break
        out = {}
        # the TIMED volumes only: the child's wall clock covers its 6 timed steps, so the warm-up volume's dispatches (the first
        # 1/7 of the trace: cold caches, first touch of the workspaces) must not be in the kernel sum it is compared with
        rows = []
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name", "")
                    if "u2::" in name:   # (blit kernels of the child's set-up copies are not the path's)
                        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name))
        rows.sort()
        # the split into volumes is only valid if every volume issued the SAME kernel sequence (`per = len // nvol` assumed it)
        per = len(rows) // nvol if rows and len(rows) % nvol == 0 else 0
        seqs_equal = bool(per) and all([n for _, _, n in rows[v * per:(v + 1) * per]] == [n for _, _, n in rows[:per]] for v in range(1, nvol))
        if seqs_equal:
            agg = {}
            for st, en, name in rows[per:]:
                a = agg.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += (en - st) / 1e3
            out = {name: (c / (nvol - 1), us / c) for name, (c, us) in agg.items()}
            # the span of the same dispatches IN THE PROFILER'S OWN CLOCK (first start to last end of the timed volumes): kernels issued
            # one behind the other on one stream cannot add up to more than that -- one process, ONE clock.  (Against the host's
            # wall clock of the same steps the GPU timestamps run ~1 % fast on this pool: the ratio is reported, not allowed for.)
            out["__span_ms__"] = (max(en for _, en, _ in rows[per:]) - rows[per][0]) / 1e6 / (nvol - 1)
            out["__volumes_same_sequence__"] = True
        else:  # (trace missing or ragged: the profiler's own summary over all 7 volumes)
            for f in glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row["Name"]
                        if "u2::" not in name and "copyBuffer" not in name:
                            continue  # torch's own kernels: input / weight initialisation of the child, not the path
                        out[name] = (int(row["Calls"]) / nvol, float(row["AverageNs"]) / 1e3)
        if out:
            out["__child_wall_ms__"] = child_wall_ms
        return out or None
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(E, timeout=240):
    """HBM-side bytes per volume of the GEMM class and of the flash kernel, LIVE: a process cannot read its own
    counters, so two short rocprofv3 passes of THIS command (one counter each: FETCH_SIZE, WRITE_SIZE -- combined passes
    abort rocprofv3 on this pool; counters + kernel trace only) run as subprocesses, 3 volumes on one stream each.
    Corrections per MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE reports half of the bytes of wide
    coalesced reads on gfx950 -> the read side is doubled.  Returns {class: bytes per volume} or None if rocprofv3 is
    unavailable / fails (bench.py then cites the newest profiles/rNN_traffic.json instead)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    nvol = 3
    agg = {}
    tmp = tempfile.mkdtemp(prefix="u2tok_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "p", "--",
                   sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--repeats", "1",
                   "--streams", "1", "--hidden", str(E), "--no-cpu-baseline", "--no-roofline", "--no-train-step"]
            r = subprocess.run(cmd, cwd="/tmp", env=_child_env(), capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(tmp, ctr, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = next((lab for key, lab in PMC_CLASSES if key in row.get("Kernel_Name", "")), None)
                        if name and row["Counter_Name"] == ctr:
                            agg.setdefault(name, {}).setdefault(ctr, 0.0)
                            agg[name][ctr] += float(row["Counter_Value"])
        if not agg:
            return None
        return {k: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 / nvol for k, v in agg.items()}
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------- training step
def train_step_full(timeout=240):
    """BASELINE configs[3] on this GPU: one whole stage-1 step (path + 36-layer Qwen3-8B-shaped decoder with gradient
    checkpointing + Zero1AdamW.step() over 9.7 B parameters; tools/train_step_full.py) in a CHILD process -- it needs 163 GiB of
    HBM of its own and must not be able to cost the inference line anything.  Not part of `value`."""
    tool = Path(__file__).resolve().parent / "tools" / "train_step_full.py"
    try:
        torch.cuda.empty_cache()
        r = subprocess.run([sys.executable, str(tool), "3"], env=_child_env(), capture_output=True, text=True, timeout=timeout)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": (r.stderr or "no output")[-300:]}
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def decoder_after_path(timeout=240):
    """The consumer of the path (SURVEY 8f rank 3 and the decode steps behind it): tools/prefill_probe.py in a CHILD process --
    Qwen3-8B-shaped decoder, prefill of the 1024 spliced embeddings and 64 greedy decode steps, stock HF layers against the
    HIP layers of u2tokenizer_amd.prefill.  Not part of `value`."""
    tool = Path(__file__).resolve().parent / "tools" / "prefill_probe.py"
    try:
        torch.cuda.empty_cache()
        r = subprocess.run([sys.executable, str(tool)], env=_child_env(), capture_output=True, text=True, timeout=timeout)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": (r.stderr or "no output")[-300:]}
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def batched_throughput(E, options, batch=4, timeout=240):
    """Context, not `value`: the same path with `batch` volumes per call (two calls in flight) -- what a serving layer that
    coalesces queued requests would see.  BASELINE configs[2] is batch 1, so the headline stays two concurrent batch-1 calls; with
    more rows per launch the ViT attention fills whole rounds of its 512 workgroup slots (768 units per volume = 1.5 rounds),
    the M = 256 products of the TTA chain become M = 256 x batch, and the tokenizer's attention needs no key splits.  A CHILD of
    this command (`--batch N --streams 2`, no profiling legs)."""
    try:
        torch.cuda.empty_cache()
        cmd = [sys.executable, str(Path(__file__).resolve()), "--batch", str(batch), "--streams", "2", "--steps", "10", "--warmup", "2",
               "--repeats", "3", "--hidden", str(E), "--no-cpu-baseline", "--no-roofline", "--no-train-step"]
        for o in options:
            cmd += ["--option", o]
        r = subprocess.run(cmd, env=_child_env(), capture_output=True, text=True, timeout=timeout)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                d = json.loads(ln)
                return {"batch": batch, "calls_in_flight": 2, "volumes_per_s": d["value"], "ms_per_call": d["ms_per_step"],
                        "what": "same path, `batch` volumes per call; not the BASELINE configuration (batch 1), not part of `value`"}
        return {"error": (r.stderr or "no output")[-300:]}
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def train_step(path, ids, qids, vol, E, iters=3):
    """Forward under autograd + backward of the path (ViT, projector, tokenizer, embedding table) on the benchmark
    configuration with a dummy loss on the spliced embeddings (the decoder and the optimiser are not part
    of it): milliseconds per phase (best of `iters`) and peak HBM.  Leaves the parameters as it found them."""
    import torch
    params = list(path.holder.parameters())
    was = [p.requires_grad for p in params]
    for p in params:
        p.requires_grad_(True)
    g = torch.Generator(device=vol.device).manual_seed(7)
    w = torch.randn(ids.shape[0], ids.shape[1], E, device=vol.device, generator=g)

    def one():
        t0 = time.perf_counter()
        emb = path.prepare_inputs_for_multimodal(ids, None, None, None, None, vol, qids)[4]
        loss = (emb.float() * w).sum()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for p in params:
            p.grad = None
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3

    try:
        with torch.enable_grad():
            one()
            torch.cuda.reset_peak_memory_stats()
            ts = [one() for _ in range(iters)]
        return {"ms_forward": round(min(t[0] for t in ts), 2), "ms_backward": round(min(t[1] for t in ts), 2),
                "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                "parameters": sum(p.numel() for p in params),
                "what": "autograd forward + backward of prepare_inputs_for_multimodal (ViT, SPP, u2Tokenizer, embedding "
                        "table) at the benchmark configuration, batch 1, dummy loss on inputs_embeds; HIP kernels both ways "
                        "(u2tokenizer_amd/autograd.py); decoder and optimiser not included; not part of `value`"}
    finally:
        for p, r in zip(params, was):
            p.requires_grad_(r)
            p.grad = None
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a torch.distributed.run parent: start the N ranks ourselves."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def timed_repeats(step, steps, warmup, repeats, sync, reduce_max):
    """`repeats` x {barrier + synchronize, EXACTLY `steps` steps, barrier + synchronize}; elapsed = MAX over ranks."""
    out = None
    for i in range(warmup):
        out = step(i)
    times = []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(i)
        sync()
        times.append(reduce_max(time.perf_counter() - t0))
    return times, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = median")
    ap.add_argument("--hidden", type=int, default=4096, help="LLM hidden size E (4096 = Qwen3-8B, 2048 = Qwen3-1.7B)")
    ap.add_argument("--batch", type=int, default=1, help="volumes per step per GPU")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (2 = two volumes in flight per GPU: the small "
                         "launches of one volume's tokenizer fill the machine under the other volume's large GEMMs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the forward + backward timing of the training path")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two live rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--cpu-baseline-iters", type=int, default=3)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning switch for A/B measurements (u2tok_set_option), e.g. --option flash_mode=7")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="TEST PLUMBING ONLY (tests/test_bench_launcher.py): replace the step by a host no-op and use the "
                         "gloo backend, so that the launcher / barrier / MAX-over-ranks / JSON path can run on a box "
                         "without GPUs.  The line it prints is marked invalid.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    from u2tokenizer_amd import replicas
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    E, B, S, Lt, Q = args.hidden, args.batch, 1024, 1024, 256
    fl = flops_per_volume(E, Lt)

    if args.stub_cpu:
        device = torch.device("cpu")
        dist, rank, world = replicas.init_from_env("gloo", device)
        x = torch.zeros(8)

        def step(i, multi=True):
            time.sleep(0.002)
            return x

        streams = None
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        # RCCL ("nccl" backend on ROCm) is used only for the timing barrier and the MAX over ranks: replicas share nothing
        dist, rank, world = replicas.init_from_env("nccl", device)
        from u2tokenizer_amd import _lib, ops
        ops.device_check()  # fails loudly off gfx950 / without the HIP library
        for kv in args.option:
            k, _, v = kv.partition("=")
            ops.set_option(k, int(v))
        torch.set_grad_enabled(False)
        vocab = 151936  # Qwen3 vocabulary
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
        torch.cuda.synchronize(device)  # weights / inputs were produced on the default stream: the side streams are non-blocking

        def step(i, multi=True):
            if streams is None or not multi:
                return path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % nvol], qids)[4]
            with torch.cuda.stream(streams[i % len(streams)]):
                return path.prepare_inputs_for_multimodal(ids, None, None, None, None, vols[i % nvol], qids)[4]

    def sync():
        replicas.barrier(dist, device)

    def rmax(x):
        return replicas.max_over_ranks(dist, x, device)

    # Two volumes in flight already overlap one volume's small launches with the other's large GEMMs; the tokenizer's own
    # side stream (TTA k|v projections, there for the latency of a volume running alone) then only adds event traffic:
    # +2.2 % with it off (tools/ab_bench.py, interleaved on one box, profiles/r02_ab_options.log).  The one-stream run
    # below keeps it on.  An explicit --option tta_overlap=... wins.
    auto_side = streams is not None and not args.stub_cpu and not any(o.startswith("tta_overlap=") for o in args.option)
    user_tta_overlap = next((int(o.split("=")[1]) for o in args.option if o.startswith("tta_overlap=")), 1)
    if auto_side:
        ops.set_option("tta_overlap", 0)
    times, out = timed_repeats(step, args.steps, args.warmup, args.repeats, sync, rmax)
    if auto_side:
        ops.set_option("tta_overlap", 1)
    if not args.stub_cpu:
        assert out.shape == (B, S, E) and bool(torch.isfinite(out.float()).all())
    elapsed = statistics.median(times)

    # the same K steps issued on ONE stream (one volume in flight): the latency of a volume
    single = None
    if streams is not None:
        t1, _ = timed_repeats(lambda i: step(i, multi=False), args.steps, args.warmup, max(1, args.repeats // 2), sync, rmax)
        single = statistics.median(t1)

    value = world * B * args.steps / elapsed
    per_step = [round(1e3 * t / args.steps, 4) for t in times]
    line = {
        "metric": "CT volumes/sec (256^3 fp16) through u2Tokenizer fwd", "value": round(value, 3),
        "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "repeats": {"n": args.repeats, "ms_per_step_each": per_step, "ms_per_step_min": min(per_step),
                    "ms_per_step_max": max(per_step), "value_is": "median repeat"},
        "tokens_per_s": round(value * Q, 1),
        "in_flight": (f"{args.streams} volumes per GPU on {args.streams} HIP streams: ms_per_step = elapsed / steps is the "
                      "reciprocal of the throughput, NOT the latency of a volume (that is ms_per_step_one_stream)"
                      if streams is not None else "1 volume per GPU"),
        "value_one_stream": round(world * B * args.steps / single, 3) if single else None,
        "ms_per_step_one_stream": round(1e3 * single / args.steps, 4) if single else None,
        "path_tflops": round(value * fl["total"] / 1e12, 1),
        "path_frac_of_bf16_mfma_peak": round(value * fl["total"] / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "config": {"workload": WORKLOAD, "hidden_size": E, "batch_per_gpu": B, "streams_per_gpu": args.streams,
                   "flop_per_volume": fl["total"], "parallelism": f"replicas x{world}",
                   "tta_side_stream": "off in the multi-stream run, on in the one-stream run" if streams is not None else "on",
                   **({"options": args.option} if args.option else {})},
    }
    if args.stub_cpu:
        line.update({"data": "stub (launcher self-test, no GPU work)", "valid": False})

    if rank == 0 and not args.no_roofline and not args.stub_cpu:
        import ctypes as C
        ops.set_option("profile", 1)
        nprof = 3
        for i in range(nprof):
            step(i, multi=False)  # one stream: a kernel's own duration, not stretched by a co-running volume
        torch.cuda.synchronize()
        h = _lib.load_library()
        h.u2tok_ctx_set_current(ops.active_context(device).handle)
        ms, flops, byts, cnt = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_double * 6)(), (C.c_int64 * 6)()
        _lib.check(h.u2tok_profile_collect2(ms, flops, byts, cnt, 6), "u2tok_profile_collect2")
        ops.set_option("profile", 0)
        names = ["gemm_bf16 (gemm_bt_kernel + gemm_bf16_nt_kernel + gemm_rows16_kernel + gemm_splitk_reduce_kernel)",
                 "flash_d64 (flash_dp2_kernel)", "temporal_attention_kernel", "row_ops", "data_movement",
                 "tok_attention (tok_attn_kernel + tok_attn_combine_kernel)"]
        # per class: time, launches, algorithmic TFLOP/s and algorithmic GB/s (operands + results once) of its launches
        classes = {n: {"ms_per_step": round(ms[i] / nprof, 4), "launches_per_step": cnt[i] // nprof,
                       "tflops": round(flops[i] / ms[i] / 1e9, 1) if ms[i] > 0 and flops[i] > 0 else None,
                       "algorithmic_gbytes_per_s": round(byts[i] / ms[i] / 1e6, 1) if ms[i] > 0 and byts[i] > 0 else None}
                   for i, n in enumerate(names)}
        # HBM-side bytes of the kernel classes: measured live by two rocprofv3 PMC passes of this command (measure_traffic);
        # if rocprofv3 cannot run here, the newest profiles/rNN_traffic.json (tools/gpu_round.sh -> tools/pmc_traffic.py)
        traffic = {}
        live = measure_traffic(E) if (B == 1 and world == 1 and not args.no_traffic) else None
        if live:
            for key, idx in (("gemm_bf16", 0), ("flash_d64", 1), ("tok_attention", 5)):
                if key in live and cnt[idx]:
                    traffic[key] = (round(live[key] / max(cnt[idx] // nprof, 1)),
                                    "live: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per volume / calls per volume, two rocprofv3 "
                                    "--pmc passes of this command started by bench.py itself (3 volumes, one stream); "
                                    "fabric-side requests (Infinity Cache hits included)")
        tfiles = sorted((ROOT / "profiles").glob("r*_traffic.json"))
        if not traffic and tfiles and E == 4096 and B == 1:
            tj = json.loads(tfiles[-1].read_text())["kernels"]
            for key, idx in (("gemm_bf16", 0), ("flash_d64", 1)):
                if key in tj and cnt[idx]:
                    # per call as counted here (a GEMM call = its main kernel + row-tail / split-K reduce launches)
                    traffic[key] = (round(tj[key]["hbm_bytes_per_step"] / max(cnt[idx] // nprof, 1)),
                                    f"{tfiles[-1].relative_to(ROOT)}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per volume / "
                                    "calls per volume; rocprofv3 --pmc, separate passes of this command; fabric-side "
                                    "requests (Infinity Cache hits included)")

        # per-kernel table from the profiler's clock (one rocprofv3 --kernel-trace --stats child of this command), and the wall
        # time it has to fit into.  Both with the tokenizer's side stream OFF: one volume, one stream, every kernel behind
        # the previous one -- the sum of kernel times then cannot exceed the wall, with no allowance for overlap.
        serial_opts = [o for o in args.option if not o.startswith("tta_overlap=")] + ["tta_overlap=0"]
        ktab = measure_kernels(E, serial_opts) if (B == 1 and world == 1 and not args.no_traffic) else None
        child_wall_ms = ktab.pop("__child_wall_ms__", None) if ktab else None
        span_ms = ktab.pop("__span_ms__", None) if ktab else None
        same_seq = ktab.pop("__volumes_same_sequence__", False) if ktab else False
        wall_serial = None
        if ktab:
            ops.set_option("profile", 0)
            ops.set_option("tta_overlap", 0)
            t2, _ = timed_repeats(lambda i: step(i, multi=False), args.steps, args.warmup, 2, sync, rmax)
            ops.set_option("tta_overlap", user_tta_overlap)   # what was in effect before (an --option tta_overlap=0 stays)
            wall_serial = statistics.median(t2)
        cls_key = ["gemm_bf16", "flash_d64", "temporal_attention", "row_ops", "data_movement", "tok_attention"]
        kt_ms = {}
        if ktab:
            table = []
            for name, (calls, us) in sorted(ktab.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                cl = next((lab for frag, lab in KERNEL_CLASSES if frag in name), "other")
                kt_ms[cl] = kt_ms.get(cl, 0.0) + calls * us / 1e3
                table.append({"kernel": name.replace("void u2::", "").replace("u2::", "")[:96], "class": cl,
                              "launches_per_volume": round(calls, 2), "avg_us": round(us, 2),
                              "ms_per_volume": round(calls * us / 1e3, 4)})
            total_ms = sum(kt_ms.values())
            wall_ms = 1e3 * wall_serial / args.steps
            line["kernel_table"] = {
                "source": "rocprofv3 --kernel-trace --stats child of this command: the 6 timed volumes of its trace (the warm-up volume "
                          "dropped), one stream, tokenizer side stream off (tta_overlap=0), no counters",
                "kernels": table[:40],
                "launches_per_volume": round(sum(k["launches_per_volume"] for k in table), 1),
                "classes": {k: {"ms_per_volume": round(v, 4),
                                "tflops": (round(flops[cls_key.index(k)] / nprof / v / 1e9, 1)
                                           if k in cls_key and flops[cls_key.index(k)] > 0 and v > 0 else None),
                                "frac_of_bf16_mfma_peak": (round(flops[cls_key.index(k)] / nprof / v / 1e9 / PEAK_BF16_TFLOPS, 4)
                                                           if k in cls_key and flops[cls_key.index(k)] > 0 and v > 0 else None)}
                            for k, v in sorted(kt_ms.items(), key=lambda kv: -kv[1])},
                "sum_ms_per_volume": round(total_ms, 4), "one_stream_serial_wall_ms_per_volume": round(wall_ms, 4),
                # kernels of one volume issued one behind the other on one stream cannot add up to more than its wall time.
                # Both sides from ONE process: the profiled child times its own 6 volumes (its ms_per_step) while the profiler
                # records the kernels of those very volumes -- no allowance.  (The un-profiled wall of this process is beside it.)
                "one_stream_serial_wall_ms_per_volume_profiled_child": (round(child_wall_ms, 4) if child_wall_ms else None),
                # one process, one clock: the summed durations of the timed volumes' kernels against the span of those very dispatches
                # in the profiler's timestamps (no allowance); the host-side wall of the same steps beside it
                "profiler_span_ms_per_volume": (round(span_ms, 4) if span_ms else None),
                "profiler_clock_over_host_clock": (round(span_ms / child_wall_ms, 4) if span_ms and child_wall_ms else None),
                "sum_le_wall": (bool(total_ms <= span_ms) if span_ms else (bool(total_ms <= child_wall_ms) if child_wall_ms else None)),
                # ... which on one serialized stream holds by construction once the bookkeeping is right; what can still catch a wrong
                # volume split or a mis-attributed dispatch: every traced volume issued the same kernel-name sequence, and
                # the kernel sum also fits the child's own HOST wall of those steps within the stated clock tolerance (the profiler's
                # timestamps run ~1 % fast against the host clock on this pool: 2 %)
                "volumes_same_kernel_sequence": bool(same_seq),
                "sum_le_child_host_wall_x1.02": (bool(total_ms <= 1.02 * child_wall_ms) if child_wall_ms else None)}

        def roof(idx, key, kernel):
            ach_ev = flops[idx] / ms[idx] / 1e9
            per_launch = max(cnt[idx], 1)
            if key in kt_ms and kt_ms[key] > 0:  # profiler clock (what rocprofv3 --stats of this command reports)
                ach = flops[idx] / nprof / kt_ms[key] / 1e9
                us = 1e3 * kt_ms[key] / max(cnt[idx] // nprof, 1)
                how = ("algorithmic FLOPs of the class's launches in one step (counted by the launchers) / their summed "
                       "kernel durations per volume from a rocprofv3 --kernel-trace --stats child of this command (one "
                       "stream, the 6 timed volumes); `achieved_hip_events` = the same FLOPs / HIP-event durations around every launch "
                       "(instrumented one-stream pass after the timed region; carries the launch gaps)")
            else:
                ach, us = ach_ev, 1e3 * ms[idx] / per_launch
                how = ("HIP events on the launch stream around every launch of the class, one-stream instrumented pass of "
                       "3 steps after the timed region")
            tr = traffic.get(key, (None, None))
            return {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": tr[0],
                    "traffic_unit": "bytes per launch (average)", "traffic_source": tr[1],
                    "algorithmic_bytes_per_launch": round(byts[idx] / per_launch),
                    "avg_launch_us": round(us, 2), "flop_per_step": flops[idx] / nprof,
                    "achieved_hip_events": round(ach_ev, 1), "avg_launch_us_hip_events": round(1e3 * ms[idx] / per_launch, 2),
                    "measured": how}

        line["roofline"] = roof(0, "gemm_bf16", "bf16 MFMA GEMM, all launches of one step (gemm_bt_kernel 256x256 / 256x192 / "
                                                "256x128 tiles in two-stage, deep and ring forms, gemm_bt_drain_kernel = the deep 256x192 form with "
                                                "tile i's epilogue under tile i+1's K loop (q|k|v, fc1 + GELU), gemm_bf16_nt_kernel 128^2 / 64^2 "
                                                "tiles, few-rows kernel, split-K reduce)")
        line["roofline"]["classes"] = classes
        if ms[1] > 0:
            line["roofline_attention"] = roof(1, "flash_d64", "flash_dp2_kernel: ViT attention, 8 chunks x 12 heads x 2049 "
                                                               "tokens x head dim 64 (MONAI SABlock, vit.py:100-105)")
        if ms[5] > 0:
            ta = roof(5, "tok_attention", "tok_attn2_kernel (8 waves: a wave pair per 16-query block) + tok_attn_combine_kernel: the "
                                          "tokenizer's own attention cores, head dim E/8 = 512 (rma.py:60-75, tta.py:55-61)")
            # arithmetic intensity of these cores: 4 Sq Skv d flop over 2 d (2 Sq + 2 Skv) bytes = 128 flop/B at 256 x 256 -- below the
            # chip's balance (2.5 PF / 8 TB/s = 310): q, k, v, o alone set an HBM-side floor, and a CU has to ingest every K / V tile
            # of its (batch, head) itself (no multicast): the second view prices the launches against the 8 TB/s roof
            us = ta["avg_launch_us"]
            ta["hbm_view"] = {"bound": "hbm", "achieved": round(ta["algorithmic_bytes_per_launch"] / us / 1e3, 1), "peak": 8000.0,
                              "unit": "GB/s", "frac": round(ta["algorithmic_bytes_per_launch"] / us / 1e3 / 8000.0, 4),
                              "note": "algorithmic bytes (q, k, v read once, o written once) / average launch duration; the key-split "
                                      "launches add fp32 partial sums on top (`traffic`)"}
            line["roofline_tokenizer_attention"] = ta
    if rank == 0 and world == 1 and not args.no_train_step and not args.stub_cpu and B == 1:
        # SURVEY 8f rank 1 (built in round 2): a measured line for the training form of the path, after the timed region
        try:
            line["train_step"] = train_step(path, ids, qids, vols[0], E)
        except Exception as e:  # never lose the inference line to the extra
            line["train_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not args.no_train_step and not args.stub_cpu and B == 1 and E == 4096:
        line["batched_calls"] = batched_throughput(E, args.option)
        line["train_step_full"] = train_step_full()
        line["decoder_after_path"] = decoder_after_path()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub_cpu:
        line["cpu_baseline"] = cpu_baseline(E, Lt, iters=args.cpu_baseline_iters)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
