#!/usr/bin/env python
"""Fused decoder prefill only (Qwen3-8B shape, S = 1024), for `rocprofv3 --kernel-trace --stats`: 1 warm-up + 4 prefills."""
import sys
from pathlib import Path

import torch
from transformers import Qwen3Config, Qwen3ForCausalLM

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd import ops  # noqa: E402
from u2tokenizer_amd.prefill import enable_fused_prefill  # noqa: E402

for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    ops.set_option(k, int(v))
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
for _ in range(5):
    m(inputs_embeds=x, use_cache=True)
torch.cuda.synchronize()
