"""INTEGRATION.md section A, checked mechanically (VERDICT r5 missing #7): the REFERENCE's own `u2_arch.py` is imported
from /root/reference with its three factories (`u2_arch.py:6-8`) swapped for `u2tokenizer_amd`'s, and the reference's own
code -- `u2MetaModel.__init__` (`u2_arch.py:10-19`), `initialize_vision_modules` (`:29-78`) through `u2LlamaForCausalLM`
(`language_model/u2llama.py:19-38`) -- constructs the path.  Asserted: the modules it built are this package's, a state dict taken
from the model the UNSWAPPED reference builds from the same arguments loads with strict=True (key for key, shape for shape), the
`pretrain_vision_model` / `pretrain_mm_mlp_adapter` branches (`u2_arch.py:64-66,74-78`) load strictly, and the two properties the
reference reads off the modules (`vision_tower.hidden_size`, `u2_arch.py:68`; `mm_projector.proj_out_num`, `train_stage1.py:369`)
have the reference's values.

CPU only, and only where the reference tree exists (the build container): MONAI is absent offline, so `vit.py:19-20` imports the
two MONAI blocks from the stub `tests/golden/make_golden.py` installs (parameter names = MONAI's; "parity unpinned" applies to their
arithmetic, not to the seam checked here).  No forward is run (the product has no CPU path)."""
import importlib
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import pytest
import torch

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "src" / "model" / "u2_arch.py").exists(), reason="needs the reference tree (build container)")

MODEL_ARGS = dict(image_channel=1, image_size=[32, 64, 64], patch_size=[4, 16, 16], vision_tower="vit3d", vision_select_layer=-1,
                  vision_select_feature="patch", mm_projector_type="spp", proj_layer_type="mlp", proj_layer_num=2,
                  proj_pooling_type="spatial", proj_pooling_size=2, enable_u2tokenizer=True, u2t_num_heads=4, u2t_num_layers=2,
                  u2t_top_k=16, use_multi_scale=True, num_3d_query_token=16, attn_type="rma", enable_diffts=True, enable_dmtp=True,
                  freeze_vision_tower=False, pretrain_vision_model=None, pretrain_mm_mlp_adapter=None)
LLAMA = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
             num_key_value_heads=2, max_position_embeddings=256, pad_token_id=0, bos_token_id=1, eos_token_id=2)


@pytest.fixture(scope="module")
def ref():
    """The reference's modules, imported once: (u2_arch module, u2llama module, its original factory trio)."""
    grad = torch.is_grad_enabled()
    sys.path.insert(0, str(REF))
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_u2_make_golden", Path(__file__).resolve().parent / "golden" / "make_golden.py")
        mk = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mk)          # (switches autograd off globally at import: restored below)
        mk.install_monai_stub()
        arch = importlib.import_module("src.model.u2_arch")
        llama = importlib.import_module("src.model.language_model.u2llama")
    finally:
        torch.set_grad_enabled(grad)
    orig = (arch.build_vision_tower, arch.build_mm_projector, arch.build_u2tokenizer_tower)
    yield arch, llama, orig
    arch.build_vision_tower, arch.build_mm_projector, arch.build_u2tokenizer_tower = orig
    sys.path.remove(str(REF))


def _construct(llama, args):
    cfg = llama.u2Config(**LLAMA)
    torch.manual_seed(0)
    m = llama.u2LlamaForCausalLM(cfg)
    assert m.get_model().get_vision_tower() is None    # (a bare Llama config: u2_arch.py:16 finds no `vision_tower`)
    m.get_model().initialize_vision_modules(NS(**args))
    return m


def _swap(arch):
    import u2tokenizer_amd as U
    arch.build_vision_tower, arch.build_mm_projector, arch.build_u2tokenizer_tower = (
        U.build_vision_tower, U.build_mm_projector, U.build_u2tokenizer_tower)


def test_reference_arch_constructs_the_path_from_swapped_factories(ref, tmp_path):
    arch, llama, orig = ref
    arch.build_vision_tower, arch.build_mm_projector, arch.build_u2tokenizer_tower = orig
    m_ref = _construct(llama, MODEL_ARGS)                 # the reference as it is
    sd_ref = {k: v.clone() for k, v in m_ref.state_dict().items()}
    _swap(arch)                                           # the two-line change
    m = _construct(llama, MODEL_ARGS)
    mm = m.get_model()
    for name in ("vision_tower", "mm_projector", "u2tokenizer"):
        mod = getattr(mm, name)
        assert type(mod).__module__.startswith("u2tokenizer_amd."), (name, type(mod))
        assert type(mod).__name__ == type(getattr(m_ref.get_model(), name)).__name__
    # the reference's checkpoint, key for key and shape for shape
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd_ref.items()}
    missing, unexpected = m.load_state_dict(sd_ref, strict=True)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd_ref[k]), k
    # what the reference reads off the modules
    assert mm.vision_tower.hidden_size == m_ref.get_model().vision_tower.hidden_size == 768 == mm.config.mm_hidden_size
    assert mm.mm_projector.proj_out_num == m_ref.get_model().mm_projector.proj_out_num == 2 * 2 * 4
    assert m.get_u2tokenizer() is mm.u2tokenizer and m.get_vision_tower() is mm.vision_tower
    # a config that already names a vision tower takes the constructor route (u2_arch.py:16-18) -- what from_pretrained does
    cfg2 = llama.u2Config(**LLAMA)
    for k, v in MODEL_ARGS.items():
        setattr(cfg2, k, v)
    cfg2.mm_hidden_size = 768
    m2 = llama.u2LlamaForCausalLM(cfg2)
    assert type(m2.get_model().vision_tower).__module__.startswith("u2tokenizer_amd.")
    assert type(m2.get_model().mm_projector).__module__.startswith("u2tokenizer_amd.")

    # pretrain_vision_model / pretrain_mm_mlp_adapter (u2_arch.py:64-66,74-78): files written from the REFERENCE's modules
    vit_file, proj_file = tmp_path / "pretrained_ViT.bin", tmp_path / "mm_projector.bin"
    torch.save({k: v + 1.0 for k, v in m_ref.get_model().vision_tower.vision_tower.state_dict().items()}, vit_file)
    torch.save({"model.mm_projector." + k: v + 2.0 for k, v in m_ref.get_model().mm_projector.state_dict().items()}, proj_file)
    m3 = _construct(llama, dict(MODEL_ARGS, pretrain_vision_model=str(vit_file), pretrain_mm_mlp_adapter=str(proj_file)))
    for k, v in m3.get_model().vision_tower.vision_tower.state_dict().items():
        assert torch.equal(v, m_ref.get_model().vision_tower.vision_tower.state_dict()[k] + 1.0), k
    for k, v in m3.get_model().mm_projector.state_dict().items():
        assert torch.equal(v, m_ref.get_model().mm_projector.state_dict()[k] + 2.0), k


def test_reference_arch_error_behaviour_through_swapped_factories(ref):
    """Unknown tower / projector names raise the reference's ValueError (builder.py of both packages)."""
    arch, llama, _ = ref
    _swap(arch)
    with pytest.raises(ValueError, match="Unknown vision tower"):
        _construct(llama, dict(MODEL_ARGS, vision_tower="resnet"))
    with pytest.raises(ValueError, match="Unknown projector type"):
        _construct(llama, dict(MODEL_ARGS, mm_projector_type="qformer"))
