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


SCHED = dict(nbs=64, warmup_epochs=3.0, warmup_momentum=0.8, warmup_bias_lr=0.1, lrf=0.01, epochs=100)


def schedule(ni, nb, global_batch, hyp=HYP, sched=SCHED, epoch=None):
    """Warm-up and accumulation state of iteration `ni` (trainer.py:337-338, 392-413): returns (accumulate, lr per group in
    the optimizer's order [biases, decayed weights, norm weights], momentum, weight_decay).
    nw = max(round(warmup_epochs * nb), 100); ni <= nw: accumulate = max(1, round(interp(ni, [0, nw], [1, nbs / batch])));
    group 0 (biases) lr falls from warmup_bias_lr to lr0 * lf(epoch), the others rise from 0; momentum rises from
    warmup_momentum; lf = linear 1 -> lrf over the epochs (trainer.py:245); weight_decay * batch * accumulate0 / nbs."""
    import numpy as np
    acc0 = max(round(sched["nbs"] / global_batch), 1)
    wd = hyp["weight_decay"] * global_batch * acc0 / sched["nbs"]
    epoch = ni // nb if epoch is None else epoch
    lf = max(1 - epoch / sched["epochs"], 0) * (1.0 - sched["lrf"]) + sched["lrf"]
    nw = max(round(sched["warmup_epochs"] * nb), 100) if sched["warmup_epochs"] > 0 else -1
    lr = hyp["lr"] * lf
    if ni <= nw:
        xi = [0, nw]
        acc = max(1, int(np.interp(ni, xi, [1, sched["nbs"] / global_batch]).round()))
        lrs = [float(np.interp(ni, xi, [sched["warmup_bias_lr"] if j == 0 else 0.0, lr])) for j in range(3)]
        mom = float(np.interp(ni, xi, [sched["warmup_momentum"], hyp["momentum"]]))
        return acc, lrs, mom, wd
    return acc0, [lr, lr, lr], hyp["momentum"], wd


def train_step(model, state, batch, hyp=HYP, world_size=1, lrs=None, momentum=None, weight_decay=None, optimize=True,
               zero_grad=True):
    """Returns (loss_items (3,), grad total norm before clipping). model is updated in place; grads left on .grad.
    lrs / momentum / weight_decay override the constants of `hyp` (warm-up); optimize=False only accumulates gradients
    (trainer.py:430: the optimizer steps every `accumulate` iterations), zero_grad=False keeps the gradients of the
    previous iteration(s) so that this backward adds to them."""
    model.train()
    if zero_grad:
        for p in model.parameters():
            p.grad = None
    feats = model(batch["img"])
    loss, items = v8_detection_loss(feats, batch, model.stride, nc=model.model[-1].nc)
    (loss.sum() * world_size).backward()
    if not optimize:
        return items, float("nan")
    lrs = [hyp["lr"]] * 3 if lrs is None else lrs
    mom = hyp["momentum"] if momentum is None else momentum
    wd0 = hyp["weight_decay"] if weight_decay is None else weight_decay
    params = [p for p in model.parameters() if p.grad is not None]
    total_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=hyp["max_norm"])
    g0, g1, g2 = param_groups(model)
    with torch.no_grad():
        for (group, wd), lr_g in zip(((g2, 0.0), (g0, wd0), (g1, 0.0)), lrs):
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
                    buf.mul_(mom).add_(g)
                g = g.add(buf, alpha=mom)  # nesterov
                p.add_(g, alpha=-lr_g)
        state.updates += 1
        d = hyp["ema_decay"] * (1 - math.exp(-state.updates / hyp["ema_tau"]))
        msd = model.state_dict()
        for k, v in state.ema.items():
            if v.dtype.is_floating_point:
                v.mul_(d)
                v.add_((1 - d) * msd[k].detach())  # two roundings, as torch_utils.py:645-646
    return items, float(total_norm)
