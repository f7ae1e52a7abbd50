"""CPU: the C-ABI library -- both builds of it: bf16 elements (libu2tok_hip.so) and IEEE-half elements (libu2tok_hip_f16.so) --
builds, loads and exports exactly what include/u2tok.h declares (no compute calls)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from u2tokenizer_amd import _lib

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "u2tok.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(u2tok_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module", params=["bf16", "f16"])
def lib(request):
    if not all(p.exists() for p in _lib._LIBS.values()):
        _lib.build()
    h = _lib.load_library(request.param)
    assert h.u2tok_elem() == request.param.encode()
    return h


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/u2tok.h but not exported by {_lib.lib_path().name}"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)


def test_identity(lib):
    assert lib.u2tok_version() >= 100
    assert lib.u2tok_arch() == b"gfx950"
    assert lib.u2tok_set_option(b"no_such_option", 1) == -1


def test_config_struct_sizes_match_header():
    # the header uses only int32_t / float fields: packed size must be 4 * nfields
    assert C.sizeof(_lib.VitConfig) == 4 * 14
    assert C.sizeof(_lib.SppConfig) == 4 * 10
    assert C.sizeof(_lib.TokConfig) == 4 * 16


def test_workspace_sizing_runs_without_a_gpu(lib):
    cfg = _lib.VitConfig(nchunk=8, img=(C.c_int32 * 3)(32, 256, 256), patch=(C.c_int32 * 3)(4, 16, 16), hidden=768,
                         mlp_dim=3072, depth=12, heads=12, vol_dtype=0, keep_cls=0, ln_eps=1e-5)
    assert lib.u2tok_vit_workspace_bytes(C.byref(cfg)) > 100 << 20
    cfg.heads = 7  # hidden != heads * 64 -> rejected
    assert lib.u2tok_vit_workspace_bytes(C.byref(cfg)) == 0
    t = _lib.TokConfig(B=1, T=8, N=256, E=4096, Lt=1024, num_heads=8, num_layers=4, top_k=1024, num_query=256,
                       use_multi_scale=1, attn_type=0, enable_diffts=1, enable_dmtp=1, max_seq_len=512,
                       diffts_tau=1.0, ln_eps=1e-5)
    assert lib.u2tok_tokenizer_workspace_bytes(C.byref(t)) > 64 << 20
    t.N = 600  # > max_seq_len: RelativeMultiheadAttention cannot index its bias table (rma.py:64-68)
    assert lib.u2tok_tokenizer_workspace_bytes(C.byref(t)) == 0
    t.N, t.enable_diffts, t.top_k = 256, 0, 4096  # top_k > T*N: torch.topk would raise
    assert lib.u2tok_tokenizer_workspace_bytes(C.byref(t)) == 0


def test_null_arguments_are_rejected_not_dereferenced(lib):
    assert lib.u2tok_gemm_bf16(None, None, None, None, None, 8, 8, 8, 8, 8, 8, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 0,
                               None) == -1
    assert lib.u2tok_topk_sorted(None, None, 1, 8, 4, None) == -1
    assert lib.u2tok_im2col_patches(None, 0, None, 1, 32, 64, 64, 4, 16, 16, None) == -1
