"""CPU: checkpoint round trips (u2tokenizer_amd/checkpoint.py) -- reference key names in, reference key names out, bit for
bit, through torch.save and safetensors, including a tokenizer whose q | k | v parameters are views of one packed buffer
(the packing itself is exercised on the host here: it is plain tensor plumbing)."""
import json
from pathlib import Path

import pytest
import torch

import u2tokenizer_amd as U
from u2tokenizer_amd import checkpoint as CK, synth
from test_host_modules import _cfg

GOLDEN = Path(__file__).resolve().parent / "golden"


def _modules(c=None):
    c = c or _cfg()
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


def test_dead_aggregator_parameters_can_live_on_the_host_losslessly():
    """offload_dead_parameters on a host module: names / values / state dict unchanged, the four tensors stop requiring grad,
    and the weight table no longer hands them to the library."""
    m = _modules()
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    tok = m.u2tokenizer
    assert tok.offload_dead_parameters() == 0            # nothing on a GPU here
    assert all(not p.requires_grad for p in tok.dead_parameters())
    assert tok._weights()[-5:] == [None] * 5 and tok._weights()[-9] is tok.tta_module.layer_linagg.linear_aggregator.wq.weight
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k


def test_prefix_selection_like_the_reference_projector_load():
    m = _modules()
    whole = {"model." + k: v.clone() for k, v in m.state_dict().items()}
    proj = U.build_mm_projector(_cfg())
    CK.load_checkpoint(proj, whole, strict=True, prefix="mm_projector")       # u2_arch.py:74-78
    for k, v in proj.state_dict().items():
        assert torch.equal(v, whole["model.mm_projector." + k])


@pytest.mark.gpu
def test_reference_checkpoint_into_packed_gpu_model_forward_parity(tmp_path):
    """SURVEY 8f-4 on the device: a reference-keyed `pytorch_model.bin` (what `u2Trainer._save` writes,
    sft_u2Trainer.py:19-22) loaded into modules that already sit PACKED on the GPU (u2_arch.py:64-66,74-78 /
    train_stage1.py:339 do strict loads into built models) -- the packed q | k | v buffers must receive it (the library
    reads those, not the nn.Parameter objects), the forward must equal the oracle run on the file's weights, and saving
    the GPU model again must reproduce the file bit for bit in both formats."""
    import pytest as _pt  # noqa: F401
    from types import SimpleNamespace as NS
    from helpers import err_stats
    from oracle import u2_oracle as O
    assert torch.cuda.is_available()
    bf, D = torch.bfloat16, "cuda"
    c = _cfg(u2t_num_layers=2, u2t_top_k=32)
    src = torch.nn.ModuleDict({"vision_tower": U.build_vision_tower(c), "mm_projector": U.build_mm_projector(c),
                               "u2tokenizer": U.build_u2tokenizer_tower(c)})
    synth.fill_module_(src, seed=11, lively=True, prefix="model.")
    ref = {k: v.to(bf) for k, v in src.state_dict().items()}       # a bf16 checkpoint, as the trainer writes one
    path = tmp_path / "ref_ckpt"
    path.mkdir()
    torch.save(ref, path / CK.WEIGHTS_NAME)

    m = _modules(c).to(bf).to(D)                                   # other weights, packed on the GPU
    before = CK.packing_report(m)
    assert before["qkv_packed"] > 0
    ptr_before = m.u2tokenizer.svt_module.attention_network.layers[0].spatial_attention.wq.weight.data_ptr()
    res = CK.load_checkpoint(m, str(path), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert CK.packing_report(m) == before
    assert m.u2tokenizer.svt_module.attention_network.layers[0].spatial_attention.wq.weight.data_ptr() == ptr_before

    img = c.image_size
    vol = synth.synth_volume(1, 2, img, seed=11, dtype=torch.float16)
    t = (0.25 * synth.synth_tensor("t_token", (1, 24, c.hidden_size), 11)).to(bf)
    with torch.no_grad():
        f = m.mm_projector(m.vision_tower(vol.to(D).view(2, 1, *img)))
        got = m.u2tokenizer(v_token=f.view(1, 2, -1, c.hidden_size), t_token=t.to(D))
    sd32 = {"model." + k: v.float() for k, v in ref.items()}
    sd16 = {"model." + k: v for k, v in ref.items()}
    oc = O.PathConfig(image_size=img, hidden_size=c.hidden_size, u2t_num_layers=2, u2t_top_k=32, num_3d_query_token=16)

    def oracle(sd, dt):
        with torch.no_grad():
            x = O.vit_tower_forward(sd, "model.vision_tower.vision_tower", vol.to(dt).view(2, 1, *img), oc)
            x = O.spp_forward(sd, "model.mm_projector", x, oc)
            return O.tokenizer_forward(sd, "model.u2tokenizer", x.view(1, 2, -1, c.hidden_size), t.to(dt), oc)[0]

    o32, o16 = oracle(sd32, torch.float32), oracle(sd16, bf)
    e_hip, e_orc = err_stats(got.float().cpu(), o32), err_stats(o16.float(), o32)
    assert e_hip["rel_rms"] <= 1.5 * e_orc["rel_rms"] + 1e-3, (e_hip, e_orc)
    for safe in (False, True):
        out = CK.read_checkpoint(CK.save_checkpoint(m, str(tmp_path / f"out{int(safe)}"), safe_serialization=safe))
        assert out.keys() == ref.keys()
        for k in ref:
            assert torch.equal(out[k], ref[k]), k
    # the reference's never-read aggregator projections (tta.py:47-48,62-65) leave HBM: same output bit for bit, same file
    E = c.hidden_size
    freed = m.u2tokenizer.offload_dead_parameters()
    assert freed == (2 * E * E + 2 * E) * 2 and not any(p.is_cuda for p in m.u2tokenizer.dead_parameters())
    with torch.no_grad():
        again = m.u2tokenizer(v_token=f.view(1, 2, -1, E), t_token=t.to(D))
    assert torch.equal(again, got)
    out = CK.read_checkpoint(CK.save_checkpoint(m, str(tmp_path / "out_offloaded"), safe_serialization=True))
    assert out.keys() == ref.keys() and all(torch.equal(out[k], ref[k]) for k in ref)
    m.to(D)                                                     # sticky: moving the model parks them again
    assert not any(p.is_cuda for p in m.u2tokenizer.dead_parameters())
