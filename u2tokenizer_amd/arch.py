"""Multimodal glue mixins (drop-in for /root/reference/src/model/u2_arch.py:10-159).

`prepare_inputs_for_multimodal` is the hot path's entry point: chunk view -> ViT3DTower -> projector ->
embedding lookup of the question -> u2Tokenizer -> splice into the prompt embeddings.  All five stages run
HIP kernels from libu2tok_hip.so; the decoder that consumes `inputs_embeds` stays stock HF.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from . import ops
from .builder import build_mm_projector, build_u2tokenizer_tower, build_vision_tower


class u2MetaModel:
    def __init__(self, config):
        super(u2MetaModel, self).__init__(config)
        self.config = config
        if hasattr(config, "vision_tower"):
            self.vision_tower = build_vision_tower(config)
            self.mm_projector = build_mm_projector(config)
            # The in-tree reference leaves the tokenizer to initialize_vision_modules (u2_arch.py:19,60-61) while the
            # shipped remote-code twin builds it here (modeling_u2Llama.py:1728); build it when the config is complete
            # so that from_pretrained() of a shipped checkpoint finds the keys.
            if getattr(config, "enable_u2tokenizer", False) and hasattr(config, "u2t_num_heads"):
                self.u2tokenizer = build_u2tokenizer_tower(config)

    def get_u2tokenizer(self):
        return getattr(self, "u2tokenizer", None)

    def get_vision_tower(self):
        return getattr(self, "vision_tower", None)

    def initialize_vision_modules(self, model_args):
        """u2_arch.py:29-78."""
        cfg = self.config
        cfg.image_channel = model_args.image_channel
        cfg.image_size = model_args.image_size
        cfg.patch_size = model_args.patch_size
        cfg.vision_tower = model_args.vision_tower
        cfg.vision_select_layer = model_args.vision_select_layer
        cfg.vision_select_feature = model_args.vision_select_feature
        cfg.mm_projector_type = model_args.mm_projector_type
        cfg.proj_layer_type = model_args.proj_layer_type
        cfg.proj_layer_num = model_args.proj_layer_num
        cfg.proj_pooling_type = model_args.proj_pooling_type
        cfg.proj_pooling_size = model_args.proj_pooling_size
        cfg.enable_u2tokenizer = model_args.enable_u2tokenizer
        cfg.u2t_num_heads = model_args.u2t_num_heads
        cfg.u2t_num_layers = model_args.u2t_num_layers
        cfg.u2t_top_k = model_args.u2t_top_k
        cfg.use_multi_scale = model_args.use_multi_scale
        cfg.num_3d_query_token = model_args.num_3d_query_token
        cfg.attn_type = getattr(model_args, "attn_type", "rma")
        cfg.enable_diffts = model_args.enable_diffts
        cfg.enable_dmtp = model_args.enable_dmtp

        if self.get_vision_tower() is None:
            self.vision_tower = build_vision_tower(cfg)
            self.vision_tower.requires_grad_(not model_args.freeze_vision_tower)
        if self.get_u2tokenizer() is None and model_args.enable_u2tokenizer:
            self.u2tokenizer = build_u2tokenizer_tower(cfg)
        if getattr(model_args, "pretrain_vision_model", None) is not None:
            weights = torch.load(model_args.pretrain_vision_model, map_location="cpu")
            self.vision_tower.vision_tower.load_state_dict(weights, strict=True)
        cfg.mm_hidden_size = self.vision_tower.hidden_size
        if getattr(self, "mm_projector", None) is None:
            self.mm_projector = build_mm_projector(cfg)
        if getattr(model_args, "pretrain_mm_mlp_adapter", None) is not None:
            weights = torch.load(model_args.pretrain_mm_mlp_adapter, map_location="cpu")

            def get_w(w, keyword):
                return {k.split(keyword + ".")[1]: v for k, v in w.items() if keyword in k}

            self.mm_projector.load_state_dict(get_w(weights, "mm_projector"), strict=True)


class u2MetaForCausalLM(ABC):
    @abstractmethod
    def get_model(self):
        pass

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def get_u2tokenizer(self):
        return self.get_model().get_u2tokenizer()

    def encode_images(self, images):
        image_features = self.get_model().get_vision_tower()(images)
        image_features = self.get_model().mm_projector(image_features)
        return image_features

    def prepare_inputs_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                      question_ids):
        """u2_arch.py:96-117, same arguments and 6-tuple return."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        embed = self.get_model().embed_tokens
        embed_w = embed.weight
        dev = embed_w.device
        # Training with a trainable embedding table (initialize_vision_tokenizer turns it on for the new tokens,
        # u2_arch.py:131-135): lookup and splice go through autograd (nn.Embedding + cat, as in the reference) so the
        # table receives its gradient; otherwise the fused HIP gather / splice kernel.
        track = torch.is_grad_enabled() and embed_w.requires_grad
        # the fused gather / splice kernel takes the table in the path's 16-bit element type (bf16, or fp16 for a model loaded in
        # float16, evalscipt/ourmodel_amos.py:33); any other table (an fp32 master copy) goes through the reference's torch ops
        plain = embed_w.dtype not in ops.ELEM_OF

        def lookup(ids):
            return embed(ids) if (track or plain) else ops.embed_splice(embed_w, ids)

        if self.config.enable_u2tokenizer:
            B, C, D, H, W = images.shape
            images, question_ids = images.to(dev), question_ids.to(dev)
            # The DPO trainer feeds every image twice -- cat([images, images]) for the chosen / rejected completions
            # (dpo_u2trainer.py:160-162) with the same question: the two halves of the vision batch are byte-identical.
            # The path is a deterministic function of (image, question): run one half and repeat the result (autograd
            # sums the two halves' gradients, as it would have).  Costs one comparison of the halves per call.
            # The probe is a device synchronisation (torch.equal), so it only runs where the duplication can occur: under
            # autograd (a training step), or when config.u2_dedup_duplicate_images is set to "always".
            dedup = getattr(self.config, "u2_dedup_duplicate_images", True)
            dup = (bool(dedup) and (dedup == "always" or torch.is_grad_enabled()) and B >= 2 and B % 2 == 0
                   and torch.equal(question_ids[:B // 2], question_ids[B // 2:])
                   and torch.equal(images[:B // 2], images[B // 2:]))
            if dup:
                B, images, question_ids = B // 2, images[:B // 2], question_ids[:B // 2]
            images = images.reshape(B * C, 1, D, H, W)
            image_features = self.encode_images(images)
            v_tokens = image_features.reshape(B, C, image_features.shape[-2], image_features.shape[-1])
            t_tokens = lookup(question_ids)
            image_features = self.get_u2tokenizer()(v_token=v_tokens, t_token=t_tokens)
            if dup:
                image_features = image_features.repeat(2, 1, 1)
        else:
            image_features = self.encode_images(images.to(dev))
        if track or plain or (torch.is_grad_enabled() and image_features.requires_grad):
            emb = embed(input_ids.to(dev))
            inputs_embeds = torch.cat((emb[:, :1, :], image_features.to(emb.dtype),
                                       emb[:, image_features.shape[1] + 1:, :]), dim=1)  # u2_arch.py:113-116
        else:
            inputs_embeds = ops.embed_splice(embed_w, input_ids.to(dev), image_features)
        return None, position_ids, attention_mask, past_key_values, inputs_embeds, labels

    def initialize_vision_tokenizer(self, model_args, tokenizer):
        """u2_arch.py:119-159 (vocabulary growth by num_new_tokens, mean-initialised)."""
        num_new_tokens = model_args.num_new_tokens
        self.resize_token_embeddings(len(tokenizer))
        if num_new_tokens > 0:
            input_embeddings = self.get_input_embeddings().weight.data
            output_embeddings = self.get_output_embeddings().weight.data
            input_embeddings[-num_new_tokens:] = input_embeddings[:-num_new_tokens].mean(dim=0, keepdim=True)
            output_embeddings[-num_new_tokens:] = output_embeddings[:-num_new_tokens].mean(dim=0, keepdim=True)
            for p in self.get_input_embeddings().parameters():
                p.requires_grad = True
            for p in self.get_output_embeddings().parameters():
                p.requires_grad = not model_args.tune_mm_mlp_adapter
        if getattr(model_args, "pretrain_mm_mlp_adapter", None):
            weights = torch.load(model_args.pretrain_mm_mlp_adapter, map_location="cpu")
            embed_tokens_weight = weights["model.embed_tokens.weight"]
            input_embeddings = self.get_input_embeddings().weight.data
            if input_embeddings.shape == embed_tokens_weight.shape:
                # The reference REBINDS its local name here (`input_embeddings = embed_tokens_weight`, u2_arch.py:155), which
                # leaves the model's embedding table untouched: the checkpoint's table is dropped.  A drop-in keeps that
                # behaviour; the evident intent (and the sibling branch below), a copy into the table, is opt-in:
                # config.u2_fix_embed_copy = True.
                if getattr(self.config, "u2_fix_embed_copy", False):
                    input_embeddings.copy_(embed_tokens_weight)
                else:
                    import warnings
                    warnings.warn("initialize_vision_tokenizer: the checkpoint's model.embed_tokens.weight has the shape of the "
                                  "model's table and is NOT loaded (the reference's u2_arch.py:155 rebinds a local name); set "
                                  "config.u2_fix_embed_copy = True to copy it", stacklevel=2)
            elif embed_tokens_weight.shape[0] == num_new_tokens:
                input_embeddings[-num_new_tokens:] = embed_tokens_weight
            else:
                raise ValueError(f"Unexpected embed_tokens_weight shape. Pretrained: {embed_tokens_weight.shape}. "
                                 f"Current: {input_embeddings.shape}. Numer of new tokens: {num_new_tokens}.")
