"""ORACLE (test infrastructure, never shipped in the product path) - rounding-point-exact bf16 emulation of the perf mode.

The HIP bf16 mode (`UPA_BF16`: bf16 storage, f32 accumulate) is NOT "the reference's arithmetic in lower precision everywhere": it
rounds to bf16 at exactly these tensors and nowhere else -

  * the model input (the reference's `im.half()`, engine/predictor.py:151-173);
  * every conv weight after BatchNorm folding (utils/torch_utils.py:236-266, folded in f32, then rounded once); biases stay f32;
  * the output of every `Conv` (conv2d + bias -> SiLU in f32, then ONE rounding) - nn/modules/conv.py:188-197;
  * a Bottleneck with shortcut (nn/modules/block.py:644-668): the second conv's activation is added to the block input IN F32 and
    the SUM is rounded once (the conv epilogue carries the residual) - not rounded before the add;
  * Detect (nn/modules/head.py:94-126, 151-169): the two `Conv`s of a branch round as above; the final `nn.Conv2d` 1x1 multiplies
    bf16 weights and bf16 activations into f32 accumulators and the decode (DFL softmax expectation, dist2bbox, sigmoid) runs on
    those f32 values - the 144 logits are never rounded; the decoded output is f32;
  * MaxPool / Upsample / Concat / chunk move bf16 values unchanged;
  * MHSA inside a BottleneckTransformer (nn/modules/block.py:6020-6092; csrc/attention.hip, the matrix-core kernel): q, k, v = the
    three 1x1 convs (bf16 weights, f32 bias) rounded once each; the scores q^T k stay f32; the keys are walked in blocks of 32 with an
    online softmax whose exponentials exp2((s - running max) log2 e) are rounded to bf16 as the second product's operand while their
    f32 values feed the denominator; the output o / l plus the block input (the shortcut) is rounded once.

`emulate_bf16(model)` returns a copy of an ORACLE model (oracle/tasks.py) that computes exactly that on the CPU: f32 torch ops on
bf16-valued tensors with a rounding where the kernels round.  What is left between it and the HIP bf16 output is the order of f32
additions inside a convolution (and the hardware's v_exp_f32 / v_rcp_f32 in SiLU and the decode, ~1 f32 ulp), i.e. an occasional
one-bf16-ulp difference where a value sits on a rounding boundary - so the HIP bf16 mode can be pinned against this emulation
per element, deterministically, instead of statistically against the f32 output (tests/test_hip_e2e.py).

Follows oracle/modules.py (itself pinned bit for bit against the imported reference); adds roundings only."""

from __future__ import annotations

import copy

import torch
import torch.nn as nn

from . import modules as om


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """f32 -> nearest-even bf16 -> f32 (torch's cast is round-to-nearest-even, as v_cvt_pk_bf16_f32)."""
    return x.to(torch.bfloat16).float()


def bf16_ulp(v: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |v| (8 significand bits): 2^(floor(log2 |v|) - 7); the smallest normal's spacing below 2^-126."""
    a = v.abs().float().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(a)) - 7.0)


def _mhsa_bf16(mh: "om.MHSA", x: torch.Tensor) -> torch.Tensor:
    """MHSA.forward (block.py:6036-6062) with the rounding points of csrc/attention.hip's bf16 matrix-core kernel; returns f32 (the
    caller adds the shortcut and rounds)."""
    b, c, w, h = x.shape
    n, hd, d = w * h, mh.heads, c // mh.heads
    q = bf16_round(mh.query(x)).view(b, hd, d, n)
    k = bf16_round(mh.key(x)).view(b, hd, d, n)
    v = bf16_round(mh.value(x)).view(b, hd, d, n)
    t_all = torch.matmul(q.permute(0, 1, 3, 2), k) * torch.tensor(1.4426950408889634, dtype=torch.float32)  # (b, hd, queries, keys)
    m = torch.full((b, hd, n), float("-inf"))
    l = torch.zeros(b, hd, n)
    o = torch.zeros(b, hd, n, d)
    for kb in range(0, n, 32):
        t = t_all[..., kb:kb + 32]
        mn = torch.maximum(m, t.amax(-1))
        alpha = torch.exp2(m - mn)
        pe = torch.exp2(t - mn[..., None])
        l = l * alpha + pe.sum(-1)
        o = o * alpha[..., None] + torch.matmul(bf16_round(pe), v[..., kb:kb + 32].transpose(-1, -2))
        m = mn
    out = o * (1.0 / l)[..., None]
    return out.permute(0, 1, 3, 2).reshape(b, c, w, h)


def emulate_bf16(model: nn.Module) -> nn.Module:
    m = copy.deepcopy(model).eval()
    # BatchNorm folded in f32 (the oracle's own fold = the product's fold_bn), then every conv weight rounded once
    for mod in m.modules():
        if isinstance(mod, om.Conv) and hasattr(mod, "bn"):
            om.fuse_conv_and_bn(mod.conv, mod.bn)
            del mod.bn
            mod.forward = mod.forward_fuse
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):  # (DFL's arange(16) "conv", block.py:250-253, belongs to the f32 decode: small integers, exact)
            mod.weight.data = bf16_round(mod.weight.data)
    deferred = set()
    for mod in m.modules():
        if isinstance(mod, om.Bottleneck) and mod.add:
            deferred.add(id(mod.cv2))          # rounded after the shortcut add, by the Bottleneck hook below
            mod.register_forward_hook(lambda _m, _i, out: bf16_round(out))
    for mod in m.modules():
        if isinstance(mod, om.BottleneckTransformer):  # x + MHSA(cv1(x)): the kernel adds the shortcut in f32 and rounds the sum
            def bt_forward(x, mod=mod):
                a = _mhsa_bf16(mod.cv2[0], mod.cv1(x))
                return bf16_round(x + a if mod.shortcut else a)
            mod.forward = bt_forward
    for mod in m.modules():
        if isinstance(mod, om.Conv) and id(mod) not in deferred:
            mod.register_forward_hook(lambda _m, _i, out: bf16_round(out))
    m.register_forward_pre_hook(lambda _m, args: (bf16_round(args[0]),) + tuple(args[1:]))
    return m
