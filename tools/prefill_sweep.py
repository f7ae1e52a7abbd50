#!/usr/bin/env python
"""Fused decoder prefill (Qwen3-8B shape, S = 1024, logits_to_keep = 1) under several library option sets, one model build.

    python tools/prefill_sweep.py "gemm_splitk=-1" "gemm_splitk=3" "gemm_tile=64,gemm_splitk=0" ...
"""
import sys
import time
from pathlib import Path

import torch
from transformers import Qwen3Config, Qwen3ForCausalLM

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402
from u2tokenizer_amd.prefill import enable_fused_prefill  # noqa: E402

cfg = Qwen3Config(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
                  num_key_value_heads=8, head_dim=128, max_position_embeddings=4096, tie_word_embeddings=False)
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
with torch.device("meta"):
    m = Qwen3ForCausalLM(cfg)
m = m.to(torch.bfloat16).to_empty(device=dev)
for p in m.parameters():
    p.normal_(0, 0.02)
m.model.rotary_emb.__init__(config=cfg, device=dev)
enable_fused_prefill(m)
x = (torch.randn(1, 1024, 4096, device=dev) * 0.05).to(torch.bfloat16)


def run(n=6):
    for _ in range(2):
        m(inputs_embeds=x, use_cache=True, logits_to_keep=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m(inputs_embeds=x, use_cache=True, logits_to_keep=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = run()
print(f"{'defaults':40s} {base:7.2f} ms")
for spec in sys.argv[1:]:
    kv = [s.split("=") for s in spec.split(",")]
    for k, v in kv:
        ops.set_option(k, int(v))
    t = run()
    for k, _ in kv:
        ops.set_option(k, 0 if k != "gemm_big_grid" else 256)
    print(f"{spec:40s} {t:7.2f} ms  x{base / t:.3f}")
print(f"{'defaults again':40s} {run():7.2f} ms")
