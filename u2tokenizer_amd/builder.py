"""Factory trio mirroring the reference seam (u2_arch.py:6-8):
multimodal_encoder/builder.py:4-9, multimodal_projector/builder.py:80-100, u2tokenizer/builder.py:3-14."""
from .projector import SpatialPoolingProjector
from .tokenizer import u2Tokenizer
from .vit import ViT3DTower


def build_vision_tower(config, **kwargs):
    vision_tower = getattr(config, "vision_tower", None)
    if vision_tower is not None and "vit3d" in vision_tower.lower():
        return ViT3DTower(config, **kwargs)
    raise ValueError(f"Unknown vision tower: {vision_tower}")


def build_mm_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, "mm_projector_type")
    if projector_type == "spp":
        return SpatialPoolingProjector(image_size=config.image_size, patch_size=config.patch_size,
                                       in_dim=config.mm_hidden_size, out_dim=config.hidden_size,
                                       layer_type=config.proj_layer_type, layer_num=config.proj_layer_num,
                                       pooling_type=config.proj_pooling_type, pooling_size=config.proj_pooling_size)
    if projector_type in ("linear", "identity"):
        raise NotImplementedError(f"mm_projector_type={projector_type!r} is outside the hot path scope "
                                  "(shipped config uses 'spp': config.json:9)")
    raise ValueError(f"Unknown projector type: {projector_type}")


def _attn_type(config):
    """u2tokenizer/builder.py:12 reads `attn_type` (default "rma").  Checkpoints written by the shipped code generation
    (base_model_tokenizers/Llama-3.2-1B-Instruct/config.json:21, u2Tokenizer.py:86-93,422) carry the boolean `enable_rpe`
    instead: true = RelativeMultiheadAttention ("rma"), false = nn.MultiheadAttention ("linvt").  `attn_type` wins when
    both are present."""
    attn_type = getattr(config, "attn_type", None)
    if attn_type is not None:
        return attn_type
    enable_rpe = getattr(config, "enable_rpe", None)
    if enable_rpe is not None:
        return "rma" if enable_rpe else "linvt"
    return "rma"


def build_u2tokenizer_tower(config, **kwargs):
    return u2Tokenizer(
        embed_size=config.hidden_size,
        num_heads=config.u2t_num_heads,
        num_layers=config.u2t_num_layers,
        top_k=config.u2t_top_k,
        use_multi_scale=config.use_multi_scale,
        num_3d_query_token=config.num_3d_query_token,
        hidden_size=config.hidden_size,
        attn_type=_attn_type(config),
        enable_diffts=config.enable_diffts,
        enable_dmtp=config.enable_dmtp,
    )
