#!/usr/bin/env python
"""Fused prefill + 33 greedy decode steps (Qwen3-8B shape) for `rocprofv3 --kernel-trace --stats`: per-kernel time of a decode
step (the prefill's kernels are listed too; divide the decode kernels' calls by 32 steps + 1 warm-up generate of 3)."""
import sys
from pathlib import Path

import torch
from transformers import Qwen3Config, Qwen3ForCausalLM

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
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
m.generate(inputs_embeds=x, max_new_tokens=33, min_new_tokens=33, do_sample=False)
torch.cuda.synchronize()
