#!/usr/bin/env python
"""What comes AFTER the path: the decoder prefill over the spliced embeddings.  Times a Qwen3-8B-
shaped stock HF decoder (random init, bf16) on inputs_embeds (1, 1024, 4096) -- the consumer of the 256 aligned tokens --
next to the tokenizer path itself, to size the next bottleneck.  Measurement only; nothing here is on the product path.

    python tools/prefill_probe.py [layers]
"""
import sys
import time

import torch
from transformers import Qwen3Config, Qwen3ForCausalLM

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 36
cfg = Qwen3Config(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=layers,
                  num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096,
                  tie_word_embeddings=False)
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
with torch.device("meta"):
    m = Qwen3ForCausalLM(cfg)
m = m.to(torch.bfloat16).to_empty(device=dev)
for p in m.parameters():
    p.normal_(0, 0.02)
m.model.rotary_emb.__init__(config=cfg, device=dev)  # buffers of a meta-built module are uninitialised
x = (torch.randn(1, 1024, 4096, device=dev) * 0.05).to(torch.bfloat16)
nparam = sum(p.numel() for p in m.parameters())
for _ in range(2):
    m(inputs_embeds=x, use_cache=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = m(inputs_embeds=x, use_cache=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
flop = 2.0 * (nparam - cfg.vocab_size * cfg.hidden_size) * 1024 + 4.0 * layers * 1024 * 1024 * 4096
print(f"Qwen3-8B-shaped prefill, {layers} layers, S=1024, bf16, stock HF on PyTorch-ROCm: {ms:.2f} ms "
      f"({flop / ms / 1e9:.0f} TFLOP/s of {flop / 1e12:.1f} TFLOP); logits {tuple(out.logits.shape)}; {nparam / 1e9:.2f} B params")

# the same prefill through the HIP layers (u2tokenizer_amd/prefill.py): fused q|k|v and gate|up GEMMs, causal grouped-query
# flash attention, residuals in the GEMM epilogues
from pathlib import Path  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from u2tokenizer_amd.prefill import enable_fused_prefill  # noqa: E402
ref = out.logits[:, -1].float()
enable_fused_prefill(m)
for _ in range(2):
    m(inputs_embeds=x, use_cache=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = m(inputs_embeds=x, use_cache=True)
torch.cuda.synchronize()
ms2 = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    m(inputs_embeds=x, use_cache=True, logits_to_keep=1)     # what generate() asks of its prefill: the last position's logits
torch.cuda.synchronize()
ms3 = (time.perf_counter() - t0) / n * 1e3
print(f"HIP layers with logits_to_keep=1 (generate's prefill): {ms3:.2f} ms")
d = (out.logits[:, -1].float() - ref)
print(f"same through u2tokenizer_amd.prefill (HIP layers): {ms2:.2f} ms ({flop / ms2 / 1e9:.0f} TFLOP/s), x{ms / ms2:.2f}; "
      f"last-position logits vs stock: rel rms {(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.3e}")

# the decode steps of generate() (eval/mrg.py asks for up to 768 new tokens): stock layers against the fused decode step
from u2tokenizer_amd.prefill import disable_fused_prefill  # noqa: E402
steps = 64
per_step = {}
for label, decode in (("stock HF layers", False), ("HIP layers (u2tokenizer_amd.prefill._decode_step)", True)):
    disable_fused_prefill(m)
    enable_fused_prefill(m, decode=decode)
    m.generate(inputs_embeds=x, max_new_tokens=4, do_sample=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.generate(inputs_embeds=x, max_new_tokens=steps + 1, min_new_tokens=steps + 1, do_sample=False)
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) * 1e3
    per_step[decode] = (tg - ms3) / steps
    print(f"generate(): prefill (HIP layers) + {steps} greedy decode steps on the {label}: {tg:.0f} ms -> "
          f"{(tg - ms3) / steps:.2f} ms per decode step (weights alone: {2 * nparam / 1e9:.1f} GB per step = "
          f"{2 * nparam / 5e12 * 1e3:.1f} ms at 5 TB/s)")
import json  # noqa: E402
print(json.dumps({"what": "Qwen3-8B-shaped decoder (36 layers, random bf16 weights) after the path: prefill of the 1024 spliced "
                          "embeddings and greedy decode steps of generate(), stock HF layers on PyTorch-ROCm against "
                          "u2tokenizer_amd.prefill (HIP layers); not part of `value`",
                  "prefill_ms_stock": round(ms, 2), "prefill_ms": round(ms2, 2), "prefill_ms_last_logits_only": round(ms3, 2),
                  "decode_ms_per_step_stock": round(per_step[False], 2), "decode_ms_per_step": round(per_step[True], 2),
                  "decode_steps_timed": steps}))

