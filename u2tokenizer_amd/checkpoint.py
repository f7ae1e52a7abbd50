"""Checkpoint-level weight handling: the reference's checkpoints in, the path's packed layout on
the GPU, the reference's checkpoints out -- losslessly.

Reference behaviour being matched:
  * `u2Trainer._save` writes `model.state_dict()` with torch.save to `pytorch_model.bin` (src/train/sft_u2Trainer.py:11-31);
    `merge_lora_weights_and_save_hf_model.py:134-153` does torch.save + `save_pretrained` (safetensors);
  * strict loads everywhere: the M3D-CLIP ViT (u2_arch.py:64-66), the projector (u2_arch.py:74-78), whole checkpoints
    (train_stage1.py:339, lu2_model.py:47).
The drop-in modules keep the reference's parameter NAMES and SHAPES, so a reference checkpoint loads with strict=True
as it is.  What differs is storage: on the GPU `u2Tokenizer.pack_weights()` lays wq | wk | wv (and biases) of every
attention module back to back so that the library runs the three projections as one GEMM -- the parameters become views
of one buffer.  torch.save would write that buffer once per view set (fine) but safetensors / `save_pretrained` refuse
tensors that share storage; `state_dict()` of the tokenizer therefore hands out unshared copies of packed parameters
(tokenizer.py: _unshare_packed), and this module adds the file-level helpers.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

WEIGHTS_NAME = "pytorch_model.bin"          # transformers.utils.WEIGHTS_NAME
SAFE_WEIGHTS_NAME = "model.safetensors"     # transformers.utils.SAFE_WEIGHTS_NAME


def export_state_dict(model: torch.nn.Module, cpu: bool = True) -> Dict[str, torch.Tensor]:
    """Reference-keyed state dict with dense, unshared tensors (ready for torch.save or safetensors)."""
    out = {}
    for k, v in model.state_dict().items():
        t = v.detach()
        if cpu:
            t = t.cpu()
        if not t.is_contiguous() or t.untyped_storage().nbytes() != t.numel() * t.element_size():
            t = t.clone(memory_format=torch.contiguous_format)
        out[k] = t
    return out


def save_checkpoint(model: torch.nn.Module, output_dir: str, safe_serialization: bool = False) -> str:
    """What u2Trainer._save writes (sft_u2Trainer.py:19-22): the state dict as `pytorch_model.bin`, or `model.safetensors`."""
    os.makedirs(output_dir, exist_ok=True)
    sd = export_state_dict(model)
    if safe_serialization:
        from safetensors.torch import save_file
        path = os.path.join(output_dir, SAFE_WEIGHTS_NAME)
        save_file(sd, path, metadata={"format": "pt"})
    else:
        path = os.path.join(output_dir, WEIGHTS_NAME)
        torch.save(sd, path)
    return path


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    if os.path.isdir(path):
        for name in (SAFE_WEIGHTS_NAME, WEIGHTS_NAME):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"no {SAFE_WEIGHTS_NAME} / {WEIGHTS_NAME} under {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint(model: torch.nn.Module, path_or_state_dict, strict: bool = True, prefix: Optional[str] = None,
                    offload_dead: bool = False):
    """model.load_state_dict(..., strict) of a reference checkpoint (file, directory or dict).  `prefix` selects a
    sub-dict the way the reference's get_w does for the projector (u2_arch.py:75-77): keys containing `prefix + "."`
    are kept with everything up to and including it removed.  Parameters are copied INTO the existing storages, so a
    tokenizer that is already packed on the GPU stays packed; it is (re)packed otherwise.  offload_dead: park the
    aggregator's never-read wv / dense (tta.py:47-48,62-65) in host memory after loading (u2Tokenizer.offload_dead_parameters)."""
    sd = path_or_state_dict if isinstance(path_or_state_dict, dict) else read_checkpoint(path_or_state_dict)
    if prefix is not None:
        sd = {k.split(prefix + ".")[1]: v for k, v in sd.items() if prefix in k}
    result = model.load_state_dict(sd, strict=strict)
    for m in model.modules():
        if hasattr(m, "pack_weights") and next(m.parameters()).is_cuda:
            m.pack_weights()
            if offload_dead:  # the reference's never-read linear_aggregator.wv / dense leave HBM; saving stays lossless
                m.offload_dead_parameters()
    return result


def packing_report(model: torch.nn.Module) -> Dict[str, int]:
    """How many attention modules have their q | k | v weights adjacent in memory (what pipeline.hip:qkv_packed tests)."""
    n = packed = 0
    for m in model.modules():
        if all(hasattr(m, a) for a in ("wq", "wk", "wv")) and hasattr(m.wq, "weight"):
            n += 1
            w = [m.wq.weight, m.wk.weight, m.wv.weight]
            step = w[0].numel() * w[0].element_size()
            packed += int(all(w[i + 1].data_ptr() == w[i].data_ptr() + step for i in (0, 1)))
    return {"attention_modules": n, "qkv_packed": packed}
