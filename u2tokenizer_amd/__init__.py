"""u2tokenizer_amd -- MI355X (gfx950) implementation of the u2Tokenizer forward path.

Drop-in `nn.Module`s with the reference's constructor arguments, state-dict keys and forward
signatures (reference: /root/reference/src/model/{u2_arch.py, multimodal_encoder/vit.py,
multimodal_projector/spatial_pooling_projector.py, u2tokenizer/*.py, language_model/u2llama.py});
their forwards run hand-written HIP kernels through the C ABI in include/u2tok.h
(libu2tok_hip.so, bound with ctypes).  There is no CPU / eager fallback: if the library is missing,
or the device is not gfx950, the forwards raise.
"""
from ._lib import lib_path, load_library, LibraryNotBuilt  # noqa: F401
from .builder import build_vision_tower, build_mm_projector, build_u2tokenizer_tower  # noqa: F401
from .vit import ViT3DTower, share_frozen_vision_tower  # noqa: F401
from .projector import SpatialPoolingProjector  # noqa: F401
from .tokenizer import u2Tokenizer  # noqa: F401
from .arch import u2MetaModel, u2MetaForCausalLM  # noqa: F401

__version__ = "0.1.0"
