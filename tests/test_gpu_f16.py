"""GPU: the building blocks of tests/test_gpu_ops.py once more on the IEEE-half build of the library (libu2tok_hip_f16.so:
same sources, -DU2_ELEM_F16; evalscipt/ourmodel_amos.py:33,70 loads the model in float16).  The cases are the SAME test
functions -- this module re-collects them with that module's element type switched to float16 (inputs are drawn / rounded as
fp16, results compared against the fp32 computation of those inputs) and its tolerance to TWO fp16 roundings (2^-11 instead
of 2^-8) -- so every kernel that takes elements is exercised in both formats, including the generated asm loops whose f16 text
is derived at build time (tools/asm_elem_f16.py).  Round 5 also re-collects tests/test_gpu_prefill.py: a decoder loaded in float16 runs the
fused prefill and decode steps on the f16 build (prefill.py takes either element type)."""
import pytest
import torch

import test_gpu_ops as T
import test_gpu_prefill as P
from test_gpu_prefill import *  # noqa: F401,F403  (the decoder-side kernels, the fused prefill and decode steps: an fp16 decoder takes them too)
from test_gpu_ops import *  # noqa: F401,F403  (test functions + the `ops` fixture, collected again here)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():   # (overrides the two imported ones: inference mode as tests/test_gpu_prefill.py sets it -- the fused prefill is a no_grad path)
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from u2tokenizer_amd import ops as _ops
    _ops.device_check()
    torch.set_grad_enabled(False)
    return _ops


@pytest.fixture(autouse=True)
def _half_elements(monkeypatch):
    monkeypatch.setattr(T, "bf", torch.float16)
    monkeypatch.setattr(T, "ULP", 2.0 ** -11)
    monkeypatch.setattr(P, "bf", torch.float16)
    monkeypatch.setattr(P, "ULP", 2.0 ** -11)
    monkeypatch.setattr(P, "EPS", 1.5e-4)
    yield


def test_the_half_build_is_what_runs(ops):
    """an fp16 GEMM goes to libu2tok_hip_f16.so (u2tok_elem() == "f16"), a bf16 one to libu2tok_hip.so, in one process"""
    from u2tokenizer_amd import _lib
    a = torch.randn(64, 64, device="cuda")
    o16, obf = ops.gemm(a.half(), a.half()), ops.gemm(a.bfloat16(), a.bfloat16())
    assert o16.dtype == torch.float16 and obf.dtype == torch.bfloat16
    assert _lib.load_library("f16").u2tok_elem() == b"f16" and _lib.load_library("bf16").u2tok_elem() == b"bf16"
    ref = a.half().float() @ a.half().float().t()
    assert (o16.float() - ref).abs().max() <= 2.0 ** -10 * ref.abs().max()
    with pytest.raises(RuntimeError):
        ops.gemm(a.half(), a.bfloat16())       # one element type per call
