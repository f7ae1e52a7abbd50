"""ctypes binding of libu2tok_hip.so / libu2tok_hip_f16.so (C ABI: include/u2tok.h).

Two builds of the same sources export the same symbols: the element type of activations and parameters is bfloat16 in
libu2tok_hip.so and IEEE half in libu2tok_hip_f16.so (csrc/common.h, U2_ELEM_F16; u2tok_elem() names it).  load_library(elem)
picks one; ops.py chooses by the dtype of the tensors it is handed."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_LIB = _PKG / "lib" / "libu2tok_hip.so"
_LIBS = {"bf16": _LIB, "f16": _PKG / "lib" / "libu2tok_hip_f16.so"}
_lock = threading.Lock()
_handles = {}


class LibraryNotBuilt(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB


def build(verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_PKG / "csrc"), "-j", str(os.cpu_count() or 4)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libu2tok_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return _LIB


class VitConfig(C.Structure):
    _fields_ = [("nchunk", C.c_int32), ("img", C.c_int32 * 3), ("patch", C.c_int32 * 3), ("hidden", C.c_int32),
                ("mlp_dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32), ("vol_dtype", C.c_int32),
                ("keep_cls", C.c_int32), ("ln_eps", C.c_float)]


class SppConfig(C.Structure):
    _fields_ = [("nchunk", C.c_int32), ("grid", C.c_int32 * 3), ("pooling_size", C.c_int32),
                ("pooling_type", C.c_int32), ("in_dim", C.c_int32), ("out_dim", C.c_int32),
                ("layer_type", C.c_int32), ("layer_num", C.c_int32)]


class TokConfig(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("E", C.c_int32), ("Lt", C.c_int32),
                ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("top_k", C.c_int32), ("num_query", C.c_int32),
                ("use_multi_scale", C.c_int32), ("attn_type", C.c_int32), ("enable_diffts", C.c_int32),
                ("enable_dmtp", C.c_int32), ("max_seq_len", C.c_int32), ("diffts_tau", C.c_float),
                ("ln_eps", C.c_float)]


class DecodeConfig(C.Structure):   # u2tok_decode_config
    _fields_ = [("B", C.c_int32), ("E", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32), ("I", C.c_int32),
                ("eps", C.c_float), ("qk_eps", C.c_float), ("scale", C.c_float)]


class TokTaps(C.Structure):
    _fields_ = [("svr_in", C.POINTER(C.c_void_p)), ("svr_out", C.POINTER(C.c_void_p)), ("visual_in", C.c_void_p),
                ("visual_out", C.c_void_p), ("tta_in", C.POINTER(C.c_void_p)), ("tta_out", C.POINTER(C.c_void_p))]


class Augment(C.Structure):
    _fields_ = [("rot90_k", C.c_int32), ("flip", C.c_int32 * 3), ("scale_factor", C.c_float), ("shift_offset", C.c_float)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/u2tok.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "u2tok_version": (_i32, []),
    "u2tok_arch": (C.c_char_p, []),
    "u2tok_elem": (C.c_char_p, []),
    "u2tok_device_check": (_i32, []),
    "u2tok_ctx_create": (_i32, [C.POINTER(_vp)]),
    "u2tok_ctx_destroy": (_i32, [_vp]),
    "u2tok_ctx_set_current": (_i32, [_vp]),
    "u2tok_ctx_get_current": (_vp, []),
    "u2tok_set_option": (_i32, [C.c_char_p, _i32]),
    "u2tok_profile_collect": (_i32, [_vp, _vp, _vp, _i32]),
    "u2tok_profile_collect2": (_i32, [_vp, _vp, _vp, _vp, _i32]),
    "u2tok_set_gemm_scratch": (_i32, [_vp, _sz, _vp]),
    "u2tok_flash_debug_buffer": (_i32, [_vp]),
    "u2tok_tok_attention_debug_buffer": (_i32, [_vp]),
    "u2tok_vit_workspace_bytes": (_sz, [C.POINTER(VitConfig)]),
    "u2tok_vit_forward": (_i32, [C.POINTER(VitConfig), C.POINTER(_vp), _vp, _vp, _vp, _sz, _vp]),
    "u2tok_spp_workspace_bytes": (_sz, [C.POINTER(SppConfig)]),
    "u2tok_spp_forward": (_i32, [C.POINTER(SppConfig), C.POINTER(_vp), _vp, _vp, _vp, _sz, _vp]),
    "u2tok_tokenizer_workspace_bytes": (_sz, [C.POINTER(TokConfig)]),
    "u2tok_tokenizer_forward": (_i32, [C.POINTER(TokConfig), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "u2tok_tokenizer_forward_taps": (_i32, [C.POINTER(TokConfig), C.POINTER(_vp), _vp, _vp, _vp, _vp, C.POINTER(TokTaps),
                                            _vp, _sz, _vp]),
    "u2tok_tok_attention_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "u2tok_tok_attention": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                   _i64, _f32, _vp, _i32, _i32, _vp, _sz, _vp]),
    "u2tok_preprocess_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "u2tok_preprocess_volume": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _sz, _vp]),
    "u2tok_preprocess_volume_aug": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32,
                                           C.POINTER(Augment), _vp, _sz, _vp]),
    "u2tok_embed_splice": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp]),
    "u2tok_gemm_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32,
                               _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _i32, _vp]),
    "u2tok_layernorm_bf16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "u2tok_softmax_rows": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _f32, _vp, _i32, _i32, _vp]),
    "u2tok_transpose_bf16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _vp]),
    "u2tok_im2col_patches": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "u2tok_avgpool3d_tokens": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "u2tok_score_gemv": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "u2tok_topk_sorted": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "u2tok_gather_rows": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "u2tok_multiscale_pool": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "u2tok_temporal_attention": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _f32, _vp,
                                        _i32, _vp]),
    "u2tok_flash_attention_d64": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _f32,
                                         _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "u2tok_flash_attention_d64_lse": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _f32,
                                             _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "u2tok_attention_gqa": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                   _i64, _f32, _i32, _vp]),
    "u2tok_attention_gqa_split": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64,
                                         _i64, _i64, _f32, _vp, _sz, _vp]),
    "u2tok_rmsnorm_bf16": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _vp]),
    "u2tok_qk_norm_rope": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _f32, _vp]),
    "u2tok_qk_norm_rope_kv": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _f32, _vp, _vp, _i32,
                                     _i64, _i32, _vp]),
    "u2tok_swiglu_bf16": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _vp]),
    "u2tok_decoder_decode_workspace_bytes": (_sz, [_vp, _i32]),
    "u2tok_decoder_decode_pre": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _i32, _vp,
                                        _sz, _vp]),
    "u2tok_decoder_decode_post": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                         _vp]),
    "u2tok_rope_apply": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp]),
    "u2tok_gelu_fwd": (_i32, [_vp, _vp, _i64, _vp]),
    "u2tok_gelu_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "u2tok_colsum_workspace_bytes": (_sz, [_i32, _i32]),
    "u2tok_colsum_bf16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp, _i32, _vp]),
    "u2tok_layernorm_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "u2tok_layernorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp]),
    "u2tok_softmax_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp]),
    "u2tok_relbias_grad": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "u2tok_rowdot_bf16": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp]),
    "u2tok_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_f32), C.POINTER(_f32), _i32, _f32, _f32, _f32,
                                _i32, _f32, _vp, _vp]),
    "u2tok_flash_attention_d64_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "u2tok_flash_attention_d64_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64,
                                             _i32, _i32, _i32, _f32, _vp, _i64, _vp, _sz, _vp]),
}

ERRORS = {-1: "U2TOK_ERR_ARG (bad dimension / null pointer / unsupported combination)",
          -2: "U2TOK_ERR_LAUNCH (kernel launch failed)",
          -3: "U2TOK_ERR_WORKSPACE (workspace too small)",
          -4: "U2TOK_ERR_DEVICE (current device is not gfx950)"}


def load_library(elem: str = None) -> C.CDLL:
    """Load the library of element type `elem` ("bf16" / "f16"; None = what the calling thread's current op runs in, bf16
    outside one); raises LibraryNotBuilt when it is absent (never falls back to anything)."""
    if elem is None:
        elem = getattr(_tls, "elem", "bf16")
    with _lock:
        h = _handles.get(elem)
        if h is not None:
            return h
        path = _LIBS[elem]
        if not path.exists():
            raise LibraryNotBuilt(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {_PKG / 'csrc'}`. There is no CPU fallback for the u2Tokenizer HIP path.")
        h = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError here == header/library drift
            fn.restype, fn.argtypes = res, args
        assert h.u2tok_elem() == elem.encode(), (path, h.u2tok_elem())
        _handles[elem] = h
        return h


_tls = threading.local()   # .elem: element type of the op the calling thread is inside (set by ops.on_device)


def thread_elem() -> str:
    return getattr(_tls, "elem", "bf16")


def set_thread_elem(elem):
    """-> previous value (None = unset)."""
    prev = getattr(_tls, "elem", None)
    if elem is None:
        if hasattr(_tls, "elem"):
            del _tls.elem
    else:
        _tls.elem = elem
    return prev


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(status, status)}")
