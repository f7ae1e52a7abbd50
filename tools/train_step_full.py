#!/usr/bin/env python
"""BASELINE configs[3] on ONE GPU: a whole stage-1 training step (src/train/train_stage1.py:244-251 with the reference's
DeepSpeed settings, config/ds_config.json:27-41) of the u2Qwen3-8B-shaped model -- HIP path forward + backward (ViT-B 3D,
projector, 4-layer tokenizer at E = 4096, embedding splice), the stock HF Qwen3-8B-shaped decoder (36 layers, random init,
bf16, gradient checkpointing as the reference's scripts enable it) forward + backward on the 1024 spliced embeddings, and one
Zero1AdamW step (bucketed flat gradients, fused HIP AdamW kernel; one rank: no collective) with global gradient clipping.

    python tools/train_step_full.py [steps]        prints one JSON line {"ms_step": ..., "ms_fwd_bwd": ..., "ms_optimizer": ...}
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def build_model(device, layers=36):
    from u2tokenizer_amd.language_model import u2Qwen3Config, u2Qwen3ForCausalLM
    import bench
    cfg = u2Qwen3Config(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=layers,
                        num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096,
                        tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    for k, v in vars(bench.path_config(4096)).items():
        if k != "hidden_size":
            setattr(cfg, k, v)
    with torch.device("meta"):
        m = u2Qwen3ForCausalLM(cfg)
    m = m.to(torch.bfloat16).to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 2 and "relative_bias" not in name and "embed_tokens" not in name and "lm_head" not in name:
                p.normal_(0, 1.0 / p.shape[1] ** 0.5, generator=g)
            elif "norm" in name and name.endswith("weight"):
                p.fill_(1.0)
            elif "query_tokens" in name:
                p.normal_(0, 0.5, generator=g)
            else:
                p.normal_(0, 0.02, generator=g)
    m.model.rotary_emb.__init__(config=cfg, device=device)  # buffers of a meta-built module are uninitialised
    m.get_model().u2tokenizer.pack_weights()
    return m, cfg


def run(steps=3, layers=36, device=None):
    from u2tokenizer_amd import dp, ops
    device = device or torch.device("cuda", 0)
    ops.device_check()
    m, cfg = build_model(device, layers)
    m.train()
    m.gradient_checkpointing_enable()
    m.config.use_cache = False
    for p in m.parameters():
        p.requires_grad_(True)
    m.get_model().u2tokenizer.offload_dead_parameters()   # never receive gradients (tta.py:47-48,62-65)
    opt = dp.Zero1AdamW(dp.hf_param_groups(m, 0.0), lr=4e-6, max_grad_norm=1.0)
    g = torch.Generator(device=device).manual_seed(1)
    B, S, Lt = 1, 1024, 1024
    vol = torch.rand((B, 8, 32, 256, 256), device=device, generator=g).half()
    ids = torch.randint(1, cfg.vocab_size, (B, S), device=device, generator=g)
    labels = ids.clone()
    labels[:, :300] = -100
    qids = torch.zeros((B, Lt), dtype=torch.int64, device=device)
    qids[:, :40] = torch.randint(1, cfg.vocab_size, (B, 40), device=device, generator=g)
    times, losses = [], []
    torch.cuda.reset_peak_memory_stats()
    with torch.enable_grad():
        for i in range(steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = m(images=vol, input_ids=ids, labels=labels, question_ids=qids)
            out.loss.backward()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            losses.append(float(out.loss))
            if i:  # the first step warms up allocator / autotuned library paths
                times.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    nparam = sum(p.numel() for p in m.parameters())
    fb, op = min(t[0] for t in times), min(t[1] for t in times)
    return {"ms_step": round(fb + op, 1), "ms_forward_backward": round(fb, 1), "ms_optimizer": round(op, 1),
            "parameters": nparam, "decoder_layers": layers, "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "grad_norm": opt.last_grad_norm, "losses": [round(x, 4) for x in losses],
            "what": "one stage-1 step on ONE GPU at BASELINE configs[3] size: u2Qwen3-8B-shaped model (HIP path + stock HF "
                    "36-layer decoder with gradient checkpointing), batch 1, 1024 spliced embeddings, loss.backward(), "
                    "Zero1AdamW.step() (flat bf16 gradient buckets, fused AdamW kernel, clipping at 1.0; one rank = no "
                    "collective); best of the timed steps"}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    print(json.dumps(run(n)), flush=True)
