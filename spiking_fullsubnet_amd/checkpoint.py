"""Checkpoint bridge: the reference's on-disk formats -> this package's modules.

The recipes save with ``accelerator.save_state(dir, safe_serialization=False)`` (audiozen/trainer.py:238-242) and restore
with ``accelerator.load_state`` (trainer.py:225): a directory holding ``pytorch_model.bin`` (model 0 -- in the frozen GAN
recipe the generator; ``pytorch_model_1.bin`` is the discriminator) or, with safe serialisation, ``model.safetensors``.
``model_zoo/intel_ndns/spike_fsb/baseline_{s,m}/checkpoints/best`` are such directories.

The two front-ends name the output projection differently (``proj`` in modeling_spiking_fullsubnet.py:82,
``fc_output_layer`` in model_low_freq.py:101) and only the live one has ``pre_layer_norm``; everything else is shared, so
a checkpoint of one can be loaded into a shape-compatible module of the other (``translate=True``).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

_MODEL_FILES = ("pytorch_model.bin", "model.safetensors")


def find_weights_file(path: str) -> str:
    """The generator's weight file for a checkpoint directory (or the file itself)."""
    path = os.path.expanduser(path)
    if os.path.isfile(path):
        return path
    if not os.path.isdir(path):
        raise FileNotFoundError(f"Checkpoint {path} not found.")  # audiozen/trainer.py:222-223
    for name in _MODEL_FILES:
        f = os.path.join(path, name)
        if os.path.isfile(f):
            return f
    raise FileNotFoundError(f"no {' / '.join(_MODEL_FILES)} in {path} (pytorch_model_1.bin is the discriminator, not the model)")


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Tensors by reference key name, on the CPU; a DistributedDataParallel ``module.`` prefix is dropped."""
    f = find_weights_file(path)
    if f.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(f, device="cpu")
    else:
        sd = torch.load(f, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
            sd = sd["state_dict"]
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def translate_keys(sd: Dict[str, torch.Tensor], target_keys) -> Dict[str, torch.Tensor]:
    """Rename the output projection to the target module's convention (``proj`` <-> ``fc_output_layer``)."""
    target_keys = set(target_keys)
    out = {}
    for k, v in sd.items():
        if k not in target_keys:
            for a, b in ((".fc_output_layer.", ".proj."), (".proj.", ".fc_output_layer.")):
                if a in k and k.replace(a, b) in target_keys:
                    k = k.replace(a, b)
                    break
        out[k] = v
    return out


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True, translate: bool = True, prepack: bool = True) -> Tuple[list, list]:
    """Load a reference checkpoint (directory or file) into ``model``; returns ``(missing_keys, unexpected_keys)``.

    With ``prepack`` and the module already on a HIP device, the kernel-side weight images (3-digit int8 planes of
    ``weight_hh`` / spike-input ``weight_ih`` / projection, folded BatchNorm) are built right away instead of at the first
    forward."""
    sd = read_state_dict(path)
    if translate:
        sd = translate_keys(sd, model.state_dict().keys())
    res = model.load_state_dict(sd, strict=strict)
    if prepack and hasattr(model, "engine") and next(model.parameters()).device.type == "cuda":
        model.engine()
    return list(res.missing_keys), list(res.unexpected_keys)
