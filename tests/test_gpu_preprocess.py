"""GPU: u2tok_preprocess_volume (the producer of the path's (8, 32, 256, 256) input, u2Transform.adaptive_resize) against
oracle/u2_preprocess_oracle.py.  PARITY UNPINNED: the oracle restates MONAI 1.3.0 (absent here) -- see its header.

Bars: crop box, resized size and both percentiles exact; voxels within 2e-5 absolute in fp32 (the value range is [0, 1];
only the summation order of the 3 x 8 interpolation / filter products differs), within one fp16 / bf16 rounding of the
oracle's fp32 value for half-precision outputs."""
import numpy as np
import pytest
import torch

from oracle import u2_preprocess_oracle as P

pytestmark = pytest.mark.gpu
D = "cuda"


def _ct_like(H, W, Dz, seed, body=None):
    """integer HU-like values with an air border so that CropForeground has something to crop"""
    rng = np.random.default_rng(seed)
    vol = np.full((H, W, Dz), -1024.0)
    h0, h1, w0, w1, d0, d1 = body or (H // 9, H - H // 11, W // 8, W - W // 13, Dz // 10, Dz - Dz // 17)
    vol[h0:h1, w0:w1, d0:d1] = rng.normal(40, 250, size=(h1 - h0, w1 - w0, d1 - d0)).round()
    vol[rng.integers(h0, h1, 50), rng.integers(w0, w1, 50), rng.integers(d0, d1, 50)] = 3000.0  # metal-like outliers
    return vol


def _run(vol, T, pad, dtype):
    from u2tokenizer_amd.preprocess import u2Transform
    tr = u2Transform(device=D, out_dtype=dtype)
    out = tr.from_array(vol, T, pad)
    torch.cuda.synchronize()
    info = tr.last_info.cpu()
    return out.cpu(), info


@pytest.mark.parametrize("shape,T,pad", [((90, 70, 40), 64, 64),      # in-plane downsample (anti-aliased), depth padded
                                         ((40, 52, 30), 64, 64),      # upsample: no anti-aliasing at all
                                         ((70, 90, 100), 64, 64),     # depth > padding_size: depth resized + filtered
                                         ((333, 301, 77), 256, 96),   # odd sizes, 3 chunks
                                         ((512, 512, 130), 256, 256)])  # a real CT geometry
def test_adaptive_resize_matches_oracle_fp32(shape, T, pad):
    vol = _ct_like(*shape, seed=sum(shape))
    ref, ri = P.adaptive_resize(vol, T, pad)
    got, info = _run(vol, T, pad, torch.float32)
    assert info[0].item() == 0
    assert info[1:4].tolist() == ri["lo"] and info[4:7].tolist() == ri["hi"]
    assert info[7:10].tolist() == ri["out_size"]
    pct = info[10:12].view(torch.float32)
    assert pct[0].item() == np.float32(ri["a_min"]) and pct[1].item() == np.float32(ri["a_max"])
    assert got.shape == ref.shape == (pad // 32, 32, T, T)
    err = (got - ref).abs().max().item()
    assert err <= 2e-5, err
    # everything outside the resized block is exact zero padding
    d, h, w = ri["out_size"]
    g = got.reshape(pad, T, T)
    assert (g[d:] == 0).all() and (g[:, h:] == 0).all() and (g[:, :, w:] == 0).all()


@pytest.mark.parametrize("aug", [dict(rot90_k=1, flip=[False, False, False], scale_factor=0.0, shift_offset=0.0),
                                 dict(rot90_k=3, flip=[True, False, True], scale_factor=0.07, shift_offset=-0.04),
                                 dict(rot90_k=2, flip=[False, True, False], scale_factor=-0.1, shift_offset=0.1),
                                 dict(rot90_k=0, flip=[True, True, True], scale_factor=0.03, shift_offset=0.0)])
def test_training_augmentations_match_oracle(aug):
    """data_type="training" (u2Transform.py:32-44) with explicit draws: rotation / flips are pure index maps (they also
    swap the H and W extents the resize geometry is computed from), the intensity jitter is one multiply-add."""
    from u2tokenizer_amd.preprocess import u2Transform
    vol = _ct_like(90, 70, 40, seed=17)
    ref, ri = P.adaptive_resize(vol, 64, 64, aug=aug)
    tr = u2Transform(data_type="training", device=D, out_dtype=torch.float32, seed=0)
    got = tr.from_array(vol, 64, 64, aug=aug).cpu()
    info = tr.last_info.cpu()
    assert info[0].item() == 0 and info[7:10].tolist() == ri["out_size"]
    assert (got - ref).abs().max().item() <= 2e-5
    # and the sampler: parameters inside MONAI's documented ranges, reproducible from the seed
    a, b = u2Transform(data_type="training", device=D, seed=3), u2Transform(data_type="training", device=D, seed=3)
    draws = [a.sample_augmentation() for _ in range(200)]
    assert draws == [b.sample_augmentation() for _ in range(200)]
    assert all(d["rot90_k"] in (0, 1, 2, 3) and abs(d["scale_factor"]) <= 0.1 and abs(d["shift_offset"]) <= 0.1 for d in draws)
    assert 60 < sum(d["rot90_k"] > 0 for d in draws) < 140 and 0 < sum(any(d["flip"]) for d in draws) < 110


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_adaptive_resize_half_outputs(dtype):
    vol = _ct_like(120, 100, 50, seed=5)
    ref, _ = P.adaptive_resize(vol, 128, 64)
    got, info = _run(vol, 128, 64, dtype)
    assert info[0].item() == 0 and got.dtype == dtype
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert ((got.float() - ref).abs() <= ulp * ref.abs() + 1e-6).all()


def test_percentiles_are_exact_order_statistics():
    """non-integer data, interpolation between two neighbours, ranks far from the ends (radix select, 3 passes)"""
    rng = np.random.default_rng(3)
    vol = rng.standard_normal((64, 48, 40)) * 1000 + rng.standard_normal((64, 48, 40))
    vol = vol.astype(np.float32).astype(np.float64)  # the GPU side takes fp32 voxels
    _, info = _run(vol, 64, 64, torch.float32)
    pct = info[10:12].view(torch.float32)
    assert pct[0].item() == np.float32(np.percentile(vol, 0.5)) and pct[1].item() == np.float32(np.percentile(vol, 99.5))


def test_empty_foreground_is_reported():
    vol = np.full((32, 32, 16), 7.0)  # constant: a_max == a_min, x - a_min == 0 everywhere -> nothing > 0
    got, info = _run(vol, 32, 32, torch.float32)
    assert info[0].item() == 1 and (got == 0).all()


def test_output_feeds_the_path():
    """(8, 32, 256, 256) fp16 straight into the ViT tower's first stage (im2col): shapes and dtype line up"""
    from u2tokenizer_amd import ops
    vol = _ct_like(300, 280, 200, seed=9)
    got, info = _run(vol, 256, 256, torch.float16)
    assert got.shape == (8, 32, 256, 256) and info[0].item() == 0
    patches = ops.im2col(got.to(D).view(8, 1, 32, 256, 256), (4, 16, 16))
    assert patches.shape == (8, 2048, 1024) and torch.isfinite(patches.float()).all()
