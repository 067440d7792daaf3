"""Weights interchange with the reference.

The reference saves checkpoints as a pickle of the whole model object (`ckpt["ema"]` / `ckpt["model"]`,
ultralytics/engine/trainer.py:579-618; loaded by ultralytics/nn/tasks.py:2291-2406), which can only be unpickled where
the reference package is importable.  The exchange format here is therefore the plain `state_dict` - the keys and shapes of
this build's modules are the reference's (tests/golden/builder_*.json) - exported once in the reference environment by
`tools/export_reference_state_dict.py`.  `load_weights` accepts such a file (`.pt` holding a state_dict or a dict with one
under "state_dict" / "ema" / "model", `.safetensors`, `.npz`), half-precision tensors (`model.half()` checkpoints) and
fused checkpoints (`model.fuse()`: conv weight + bias, no `bn.*` keys - mapped onto an identity BatchNorm so that the
fold performed when the HIP weights are packed reproduces exactly (W', b'))."""

from __future__ import annotations

from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


def _read(source) -> "OrderedDict[str, torch.Tensor]":
    if isinstance(source, dict):
        obj = source
    else:
        path = Path(source)
        if path.suffix == ".safetensors":
            from safetensors.torch import load_file
            obj = load_file(str(path))
        elif path.suffix == ".npz":
            obj = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(path).items()}
        else:
            try:
                obj = torch.load(str(path), map_location="cpu", weights_only=True)
            except Exception as e:  # a pickled reference model object: needs the reference's classes
                raise L.UpaError(
                    f"{path}: not a plain state_dict ({type(e).__name__}: {e}). Reference checkpoints pickle the whole model; "
                    "export its state_dict once where the reference is importable: "
                    "`python tools/export_reference_state_dict.py best.pt best_state.pt`") from e
    for key in ("state_dict", "ema", "model"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
            break
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise L.UpaError("weights source does not hold a state_dict (name -> tensor)")
    return OrderedDict((k, v) for k, v in obj.items())


def _unfuse(sd, model):
    """Fused checkpoint -> keys of the unfused module tree: conv keeps (W', no bias), BN becomes the identity carrying b'."""
    out = OrderedDict(sd)
    for name, mod in model.named_modules():
        bn = getattr(mod, "bn", None)
        conv = getattr(mod, "conv", None)
        if not (isinstance(bn, nn.BatchNorm2d) and isinstance(conv, nn.Conv2d)):
            continue
        wkey, bkey = f"{name}.conv.weight", f"{name}.conv.bias"
        if wkey in sd and f"{name}.bn.weight" not in sd:
            c = bn.num_features
            b = sd.get(bkey)
            out.pop(bkey, None)
            out[f"{name}.bn.weight"] = torch.ones(c)
            out[f"{name}.bn.bias"] = torch.zeros(c) if b is None else b.float()
            out[f"{name}.bn.running_mean"] = torch.zeros(c)
            out[f"{name}.bn.running_var"] = torch.full((c,), 1.0 - bn.eps)  # gamma / sqrt(var + eps) == 1 exactly
            out[f"{name}.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return out


def load_weights(model: nn.Module, source, strict: bool = False) -> dict:
    """Load a reference state_dict into `model` (same semantics as BaseModel.load, nn/tasks.py:1181-1200: keys are
    intersected by name and shape). Returns {"loaded": n, "total": m, "missing": [...], "unexpected": [...]}."""
    sd = _unfuse(_read(source), model)
    own = model.state_dict()
    ok, unexpected = OrderedDict(), []
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) == tuple(v.shape):
            ok[k] = v.to(own[k].dtype) if v.dtype.is_floating_point else v
        else:
            unexpected.append(k)
    missing = [k for k in own if k not in ok]
    if strict and (missing or unexpected):
        raise L.UpaError(f"load_weights(strict): missing {missing[:5]}..., unexpected {unexpected[:5]}...")
    model.load_state_dict(ok, strict=False)
    for m in model.modules():  # drop packed-weight caches
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()
    return {"loaded": len(ok), "total": len(own), "missing": missing, "unexpected": unexpected}


def save_state_dict(model_or_sd, path) -> None:
    """Write a reference-compatible state_dict (float32 CPU tensors) that the reference can `load_state_dict`."""
    sd = model_or_sd.state_dict() if isinstance(model_or_sd, nn.Module) else model_or_sd
    torch.save(OrderedDict((k, v.detach().float().cpu() if v.dtype.is_floating_point else v.detach().cpu())
                           for k, v in sd.items()), str(path))
