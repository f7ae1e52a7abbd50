"""HuggingFace surface (drop-in for /root/reference/src/model/language_model/{u2llama,u2qwen3,u2phi3}.py).

`forward(images, input_ids, labels, attention_mask, question_ids, ...)` and
`generate(images, inputs, question_ids=..., **kwargs)` keep the reference's call shapes
(u2llama.py:41-127; eval/mrg.py:74; dpo_u2trainer.py:71-79,267-272).  The decoder itself is stock HF on
PyTorch-ROCm; only `prepare_inputs_for_multimodal` (arch.py) runs the HIP path.
"""
from __future__ import annotations

from typing import Any, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from transformers import (AutoConfig, AutoModelForCausalLM, LlamaConfig, LlamaForCausalLM, LlamaModel, Phi3Config,
                          Phi3ForCausalLM, Phi3Model, Qwen3Config, Qwen3ForCausalLM, Qwen3Model)
from transformers.modeling_outputs import CausalLMOutputWithPast

from .arch import u2MetaForCausalLM, u2MetaModel


class u2Config(LlamaConfig):
    model_type = "u2llama"


class u2Qwen3Config(Qwen3Config):
    model_type = "u2Qwen3"


class u2Phi3Config(Phi3Config):
    model_type = "u2phi3"


class u2LlamaModel(u2MetaModel, LlamaModel):
    config_class = u2Config

    def __init__(self, config: LlamaConfig):
        super(u2LlamaModel, self).__init__(config)


class u2Qwen3Model(u2MetaModel, Qwen3Model):
    config_class = u2Qwen3Config

    def __init__(self, config: Qwen3Config):
        super(u2Qwen3Model, self).__init__(config)


class u2Phi3Model(u2MetaModel, Phi3Model):
    config_class = u2Phi3Config

    def __init__(self, config: Phi3Config):
        super(u2Phi3Model, self).__init__(config)


class _u2CausalLMMixin(u2MetaForCausalLM):
    """forward / generate / prepare_inputs_for_generation shared by the Llama, Qwen3 and Phi3 builds."""

    def get_model(self):
        return self.model

    def _maybe_fuse_prefill(self) -> None:
        """The prefill of the spliced embeddings through the HIP decoder layers (u2tokenizer_amd/prefill.py)
        unless `config.u2_fused_prefill` is False: patched once, when the decoder sits on the GPU in bf16 (or fp16: the f16 build).  Training,
        decode steps, padded batches and CPU runs keep the stock HuggingFace layers."""
        if getattr(self, "_u2_prefill_checked", False) or not getattr(self.config, "u2_fused_prefill", True):
            return
        p = next(self.model.layers[0].parameters(), None) if len(self.model.layers) else None
        if p is not None and p.is_cuda and p.dtype in (torch.bfloat16, torch.float16):   # (either build of the library)
            from .prefill import enable_fused_prefill
            enable_fused_prefill(self, strict=False)   # (layers of another layout -- Phi3 -- stay stock)
            self._u2_prefill_checked = True

    def forward(self, images: Optional[torch.FloatTensor] = None, input_ids: torch.LongTensor = None,
                labels: Optional[torch.LongTensor] = None, attention_mask: Optional[torch.Tensor] = None,
                question_ids: Optional[torch.LongTensor] = None, position_ids: Optional[torch.LongTensor] = None,
                past_key_values: Optional[List[torch.FloatTensor]] = None,
                inputs_embeds: Optional[torch.FloatTensor] = None, use_cache: Optional[bool] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None, **kwargs) -> Union[Tuple, CausalLMOutputWithPast]:
        # aliases used by the in-tree Qwen3 variant (u2qwen3.py:42-47)
        images = kwargs.pop("vision_input", images)
        question_ids = kwargs.pop("raw_question_ids", question_ids)
        if not torch.is_grad_enabled():
            self._maybe_fuse_prefill()
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_for_multimodal(input_ids, position_ids, attention_mask, past_key_values, labels,
                                                   images, question_ids)
        for k, v in (("output_attentions", output_attentions), ("output_hidden_states", output_hidden_states),
                     ("return_dict", return_dict)):
            if v is not None:
                kwargs[k] = v
        return super().forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                               past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                               use_cache=use_cache, **kwargs)

    @torch.no_grad()
    def generate(self, images: Optional[torch.Tensor] = None, inputs: Optional[torch.Tensor] = None,
                 question_ids: Optional[torch.Tensor] = None, **kwargs) -> Any:
        position_ids = kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        if images is not None:
            (inputs, position_ids, attention_mask, _, inputs_embeds, _) = self.prepare_inputs_for_multimodal(
                inputs, position_ids, attention_mask, None, None, images, question_ids)
        else:
            inputs_embeds = self.get_model().embed_tokens(inputs)
        return super().generate(inputs_embeds=inputs_embeds, **kwargs)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        images = kwargs.pop("images", None)
        inputs = super().prepare_inputs_for_generation(input_ids, past_key_values=past_key_values,
                                                       inputs_embeds=inputs_embeds, **kwargs)
        if images is not None:
            inputs["images"] = images
        return inputs


class u2LlamaForCausalLM(_u2CausalLMMixin, LlamaForCausalLM):
    config_class = u2Config

    def __init__(self, config):
        super(LlamaForCausalLM, self).__init__(config)
        self.model = u2LlamaModel(config)
        self.pretraining_tp = getattr(config, "pretraining_tp", 1)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()


class u2Qwen3ForCausalLM(_u2CausalLMMixin, Qwen3ForCausalLM):
    config_class = u2Qwen3Config

    def __init__(self, config):
        super(Qwen3ForCausalLM, self).__init__(config)
        self.model = u2Qwen3Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()


class u2Phi3ForCausalLM(_u2CausalLMMixin, Phi3ForCausalLM):
    """language_model/u2phi3.py:25-140 (train_stage1.py:290-296, model_type "phi3").  The path in front of the decoder is the
    same HIP path; the Phi3 decoder (fused qkv_proj / gate_up_proj modules) stays the stock HuggingFace one: the fused
    prefill of prefill.py knows the Llama / Qwen3 layer layout only and leaves these layers alone."""
    config_class = u2Phi3Config

    def __init__(self, config):
        super(Phi3ForCausalLM, self).__init__(config)
        self.model = u2Phi3Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()


def register_auto_classes() -> None:
    """AutoConfig/AutoModelForCausalLM registration (u2llama.py:141-142, u2qwen3.py:144-145, u2phi3.py:139-140); idempotent."""
    for cfg, cls in ((u2Config, u2LlamaForCausalLM), (u2Qwen3Config, u2Qwen3ForCausalLM), (u2Phi3Config, u2Phi3ForCausalLM)):
        try:
            AutoConfig.register(cfg.model_type, cfg)
            AutoModelForCausalLM.register(cfg, cls)
        except ValueError:
            pass  # already registered (e.g. the reference package was imported first)


register_auto_classes()
