"""ORACLE (test infrastructure, never imported by the product): one YOLOv8 detection training step on CPU.

Restates ultralytics/engine/trainer.py:416-432 (forward, loss.sum() * world_size, backward), :674-682 (optimizer_step:
clip_grad_norm_ 10.0, SGD step, zero_grad, EMA), :891-950 (build_optimizer parameter groups: biases / decayed weights /
norm weights; SGD nesterov) and ultralytics/utils/torch_utils.py:606-650 (ModelEMA) with plain torch autograd.
AMP / GradScaler are off (f32 parity mode); the product's bf16 mode is compared against this within a looser bound.
"""

import math
from copy import deepcopy

import torch
import torch.nn as nn

from .loss import v8_detection_loss

HYP = dict(lr=0.01, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema_decay=0.9999, ema_tau=2000.0)


def param_groups(model):
    """trainer.py:917-926: (decayed weights, norm weights, biases) by parameter name / owning module type."""
    g0, g1, g2 = [], [], []
    norm = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for mname, mod in model.named_modules():
        for pname, p in mod.named_parameters(recurse=False):
            full = f"{mname}.{pname}" if mname else pname
            if "bias" in full:
                g2.append((full, p))
            elif isinstance(mod, norm):
                g1.append((full, p))
            else:
                g0.append((full, p))
    return g0, g1, g2


class TrainState:
    """Momentum buffers + EMA copy (torch.optim.SGD state / ModelEMA)."""

    def __init__(self, model):
        self.momentum = {}
        self.ema = {k: v.detach().clone() for k, v in deepcopy(model).state_dict().items()}
        self.updates = 0


def train_step(model, state, batch, hyp=HYP, world_size=1):
    """Returns (loss_items (3,), grad total norm before clipping). model is updated in place; grads left on .grad."""
    model.train()
    for p in model.parameters():
        p.grad = None
    feats = model(batch["img"])
    loss, items = v8_detection_loss(feats, batch, model.stride, nc=model.model[-1].nc)
    (loss.sum() * world_size).backward()
    params = [p for p in model.parameters() if p.grad is not None]
    total_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=hyp["max_norm"])
    g0, g1, g2 = param_groups(model)
    with torch.no_grad():
        for group, wd in ((g2, 0.0), (g0, hyp["weight_decay"]), (g1, 0.0)):
            for name, p in group:
                if p.grad is None:
                    continue
                g = p.grad
                if wd:
                    g = g.add(p, alpha=wd)
                buf = state.momentum.get(name)
                if buf is None:
                    buf = state.momentum[name] = g.clone()  # torch.optim.SGD: first step copies the gradient
                else:
                    buf.mul_(hyp["momentum"]).add_(g)
                g = g.add(buf, alpha=hyp["momentum"])  # nesterov
                p.add_(g, alpha=-hyp["lr"])
        state.updates += 1
        d = hyp["ema_decay"] * (1 - math.exp(-state.updates / hyp["ema_tau"]))
        msd = model.state_dict()
        for k, v in state.ema.items():
            if v.dtype.is_floating_point:
                v.mul_(d)
                v.add_((1 - d) * msd[k].detach())  # two roundings, as torch_utils.py:645-646
    return items, float(total_norm)
