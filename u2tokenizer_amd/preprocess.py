"""GPU form of the reference's volume preprocessing (drop-in for /root/reference/src/utils/u2Transform.py, validation
transforms): NIfTI array -> percentile intensity scaling -> foreground crop -> anti-aliased in-plane resize to 256 (depth
padded or resized to 256) -> (8, 32, 256, 256), all in libu2tok_hip.so (`u2tok_preprocess_volume`).  data_type="training"
adds the reference's augmentations (u2Transform.py:32-44: RandRotate90 over (H, W), RandFlip x3, RandScaleIntensity,
RandShiftIntensity) between the crop and the resize (`u2tok_preprocess_volume_aug`): the random draws are made here on the
host from the distributions MONAI documents, the kernels are deterministic given the draws."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib, ops

_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


class u2Transform:
    """Same call surface as the reference class for data_type="validation": `t(input_path)` or
    `t.adaptive_resize(input_path, target_image_size=256, padding_size=256)`; plus `from_array` for a volume that is
    already in memory.  Output dtype is selectable (the hot path takes fp16 / bf16 / fp32 voxels)."""

    def __init__(self, mode: str = "bilinear", data_type: str = "validation", device="cuda",
                 out_dtype: torch.dtype = torch.float16, lower: float = 0.5, upper: float = 99.5, seed: Optional[int] = None):
        if mode != "bilinear":
            raise ValueError("the HIP resize implements the reference's mode='bilinear' (trilinear for volumes)")
        if data_type not in ("validation", "training"):
            raise ValueError(f"data_type must be 'validation' or 'training', got {data_type!r}")
        self.training = data_type == "training"
        self.R = np.random.RandomState(seed)
        if out_dtype not in _DT:
            raise ValueError(f"unsupported output dtype {out_dtype}")
        self.device, self.out_dtype, self.lower, self.upper = torch.device(device), out_dtype, lower, upper
        self._ws = ops._Workspace()
        self.last_info: Optional[torch.Tensor] = None  # int32[12]: status, crop box, resized size, percentiles (bits)

    def sample_augmentation(self) -> dict:
        """One draw of the training-time augmentations (u2Transform.py:37-42), in Compose order, from the distributions
        MONAI documents: RandRotate90(prob=0.5, max_k=3), RandFlip(prob=0.1) per spatial axis, RandScaleIntensity(
        factors=0.1, prob=0.5) -> factor ~ U(-0.1, 0.1), RandShiftIntensity(offsets=0.1, prob=0.5) -> offset ~ U(-0.1, 0.1).
        (MONAI seeds each transform of a Compose separately; reproducing its exact stream needs MONAI itself.)"""
        R = self.R
        k = int(R.randint(3) + 1) if R.rand() < 0.5 else 0
        flip = [bool(R.rand() < 0.10) for _ in range(3)]
        factor = float(R.uniform(-0.1, 0.1)) if R.rand() < 0.5 else 0.0
        offset = float(R.uniform(-0.1, 0.1)) if R.rand() < 0.5 else 0.0
        return dict(rot90_k=k, flip=flip, scale_factor=factor, shift_offset=offset)

    def from_array(self, data_hwd, target_image_size: int = 256, padding_size: int = 32 * 8, aug: Optional[dict] = None):
        """data_hwd: array / tensor of shape (H, W, D) as nib.load(path).get_fdata() returns it."""
        t = torch.as_tensor(data_hwd)
        # u2Transform.py:68-69: .transpose(2, 0, 1) then torch.tensor(..., device); the channel axis is implicit here
        vol = t.permute(2, 0, 1).to(device=self.device, dtype=torch.float32).contiguous()
        return self.from_dhw(vol, target_image_size, padding_size, aug)

    def from_dhw(self, vol: torch.Tensor, target_image_size: int = 256, padding_size: int = 32 * 8,
                 aug: Optional[dict] = None) -> torch.Tensor:
        """aug: explicit augmentation parameters (keys of sample_augmentation()); None = draw them when
        data_type == "training", none otherwise."""
        h = _lib.load_library()
        vol = ops._need(vol, torch.float32, "volume").contiguous()
        if vol.dim() == 4 and vol.shape[0] == 1:
            vol = vol[0]
        if vol.dim() != 3:
            raise RuntimeError(f"expected a (D, H, W) volume, got {tuple(vol.shape)}")
        if padding_size % 32:
            raise RuntimeError("padding_size must be a multiple of 32 (the path consumes 32-slice chunks)")
        D, H, W = vol.shape
        with ops.on_device(vol) as (h, stream):
            ws = self._ws.get(h.u2tok_preprocess_workspace_bytes(D, H, W), vol.device)
            out = torch.empty((padding_size // 32, 32, target_image_size, target_image_size), dtype=self.out_dtype,
                              device=vol.device)
            info = torch.empty(12, dtype=torch.int32, device=vol.device)
            if aug is None and self.training:
                aug = self.sample_augmentation()
            if aug is None:
                _lib.check(h.u2tok_preprocess_volume(vol.data_ptr(), out.data_ptr(), info.data_ptr(), D, H, W,
                                                     target_image_size, padding_size, float(self.lower), float(self.upper),
                                                     _DT[self.out_dtype], ws.data_ptr(), ws.numel(), stream),
                           "u2tok_preprocess_volume")
            else:
                import ctypes as C
                a = _lib.Augment(rot90_k=int(aug.get("rot90_k", 0)), flip=(C.c_int32 * 3)(*[int(f) for f in aug.get("flip", (0, 0, 0))]),
                                 scale_factor=float(aug.get("scale_factor", 0.0)), shift_offset=float(aug.get("shift_offset", 0.0)))
                _lib.check(h.u2tok_preprocess_volume_aug(vol.data_ptr(), out.data_ptr(), info.data_ptr(), D, H, W,
                                                         target_image_size, padding_size, float(self.lower),
                                                         float(self.upper), _DT[self.out_dtype], C.byref(a), ws.data_ptr(),
                                                         ws.numel(), stream), "u2tok_preprocess_volume_aug")
        self.last_info, self.last_aug = info, aug
        return out

    def adaptive_resize(self, input_path, target_image_size: int = 256, padding_size: int = 32 * 8) -> torch.Tensor:
        import nibabel as nib  # same loader as the reference (u2Transform.py:68); not needed for from_array
        return self.from_array(nib.load(input_path).get_fdata(), target_image_size, padding_size)

    __call__ = adaptive_resize
