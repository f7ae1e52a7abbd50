"""CPU: checkpoint round trips (u2tokenizer_amd/checkpoint.py) -- reference key names in, reference key names out, bit for
bit, through torch.save and safetensors, including a tokenizer whose q | k | v parameters are views of one packed buffer
(the packing itself is exercised on the host here: it is plain tensor plumbing)."""
import json
from pathlib import Path

import torch

import u2tokenizer_amd as U
from u2tokenizer_amd import checkpoint as CK, synth
from test_host_modules import _cfg

GOLDEN = Path(__file__).resolve().parent / "golden"


def _modules():
    c = _cfg()
    m = torch.nn.ModuleDict({"vision_tower": U.build_vision_tower(c), "mm_projector": U.build_mm_projector(c),
                             "u2tokenizer": U.build_u2tokenizer_tower(c)})
    synth.fill_module_(m, seed=3)
    return m


def test_round_trip_through_both_file_formats(tmp_path):
    m = _modules()
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    assert {k: list(v.shape) for k, v in ref.items()} == json.loads((GOLDEN / "state_dict_keys.json").read_text())["mu2_small"]
    m.u2tokenizer._packed_key = None
    m.u2tokenizer.pack_weights()                      # q | k | v now share one buffer per attention module
    rep = CK.packing_report(m)
    assert rep["qkv_packed"] == rep["attention_modules"] - 1 > 0   # (the un-projected aggregator is never packed)
    for safe in (False, True):
        path = CK.save_checkpoint(m, str(tmp_path / f"ck{int(safe)}"), safe_serialization=safe)
        sd = CK.read_checkpoint(path)
        assert sd.keys() == ref.keys()
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k
            assert sd[k].untyped_storage().nbytes() == sd[k].numel() * sd[k].element_size() or safe
        fresh = _modules()
        for p in fresh.parameters():
            p.data.zero_()
        res = CK.load_checkpoint(fresh, str(tmp_path / f"ck{int(safe)}"), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        for k, v in fresh.state_dict().items():
            assert torch.equal(v, ref[k]), k


def test_loading_into_a_packed_module_keeps_it_packed():
    m, src = _modules(), _modules()
    synth.fill_module_(src, seed=9)
    m.u2tokenizer._packed_key = None
    m.u2tokenizer.pack_weights()
    before = CK.packing_report(m)
    CK.load_checkpoint(m, {k: v.clone() for k, v in src.state_dict().items()})
    assert CK.packing_report(m) == before
    for (k, a), b in zip(m.state_dict().items(), src.state_dict().values()):
        assert torch.equal(a, b), k


def test_prefix_selection_like_the_reference_projector_load():
    m = _modules()
    whole = {"model." + k: v.clone() for k, v in m.state_dict().items()}
    proj = U.build_mm_projector(_cfg())
    CK.load_checkpoint(proj, whole, strict=True, prefix="mm_projector")       # u2_arch.py:74-78
    for k, v in proj.state_dict().items():
        assert torch.equal(v, whole["model.mm_projector." + k])
