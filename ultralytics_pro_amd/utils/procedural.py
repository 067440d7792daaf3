"""Procedural (counter-hash) weights and images.

The reference publishes no checkpoints we can ship and its default random init produces zero detections
(cls bias = log(5/nc/(640/s)^2), ultralytics/nn/modules/head.py:171-178), so every parity / bench run uses
weights and images generated from a counter-based integer hash: value(key, i) depends only on the state_dict
key string and the flat element index, therefore the container that produced the golden fixtures, the GPU box
and every DP rank build bit-identical tensors without shipping data.

Pure numpy + torch (CPU); no reference code involved.
"""

from __future__ import annotations

import math
import zlib

import numpy as np
import torch

_C0 = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_S30, _S27, _S31, _S40 = (np.uint64(v) for v in (30, 27, 31, 40))


def _splitmix64_inplace(z: np.ndarray, tmp: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, in place on a uint64 array (uint64 arithmetic wraps modulo 2**64)."""
    z += _C0
    np.right_shift(z, _S30, out=tmp); z ^= tmp; z *= _C1
    np.right_shift(z, _S27, out=tmp); z ^= tmp; z *= _C2
    np.right_shift(z, _S31, out=tmp); z ^= tmp
    return z


def hash_uniform(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [0, 1): top 24 bits of splitmix64(base(key, seed) + i) / 2**24."""
    stream = np.array([(zlib.crc32(key.encode()) & 0xFFFFFFFF) | ((seed & 0xFFFFFFFF) << 32)], dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _splitmix64_inplace(stream, np.empty_like(stream))[0]
        out = np.empty(n, dtype=np.float32)
        step = 1 << 18  # cache-sized temporaries (rtdetr has 104 M params)
        ramp = np.arange(step, dtype=np.uint64)
        z = np.empty(step, dtype=np.uint64)
        tmp = np.empty(step, dtype=np.uint64)
        for lo in range(0, n, step):
            m = min(step, n - lo)
            np.add(ramp[:m], base + np.uint64(lo), out=z[:m])
            _splitmix64_inplace(z[:m], tmp[:m])
            np.right_shift(z[:m], _S40, out=z[:m])
            out[lo:lo + m] = z[:m]
    out *= np.float32(1.0 / (1 << 24))
    return out


def uniform(key: str, shape, lo: float, hi: float, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(key, n, seed)
    v = (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)
    return torch.from_numpy(v).reshape(tuple(shape))


def synthetic_images(batch: int, ch: int = 3, h: int = 640, w: int = 640, seed: int = 0,
                     first: int = 0) -> torch.Tensor:
    """Structured fp32 NCHW images in [0,1] (the post-`/255` domain of engine/predictor.py:151-173).

    Pure iid noise makes deep random-weight features collapse to constants, so each image is: a blocky background
    (40-px cells, U(0.2,0.8)) + 4..11 constant-colour rectangles ("synthetic boxes") + U(-0.08,0.08) pixel noise,
    clamped.  Only hash draws, float32 add and clip are used (no transcendental functions), so the result is
    bit-identical on every host.  Image i of a batch depends only on (seed, first + i): DP ranks take disjoint
    `first` offsets.
    """
    out = np.empty((batch, ch, h, w), dtype=np.float32)
    cell = 40
    gh, gw = -(-h // cell), -(-w // cell)
    for b in range(batch):
        gi = first + b
        grid = 0.2 + 0.6 * hash_uniform(f"img:cells:{gi}", ch * gh * gw, seed).reshape(ch, gh, gw)
        img = np.repeat(np.repeat(grid, cell, axis=1), cell, axis=2)[:, :h, :w].astype(np.float32).copy()
        pr = hash_uniform(f"img:boxes:{gi}", 16 + 8 * 12, seed)
        nbox = 4 + int(pr[0] * 8)
        for k in range(nbox):
            q = pr[16 + 8 * k: 24 + 8 * k]
            cx, cy = float(q[0]) * w, float(q[1]) * h
            bw, bh = (0.05 + 0.35 * float(q[2])) * w, (0.05 + 0.35 * float(q[3])) * h
            x1, x2 = int(max(0.0, cx - bw / 2)), int(min(float(w), cx + bw / 2))
            y1, y2 = int(max(0.0, cy - bh / 2)), int(min(float(h), cy + bh / 2))
            for c in range(ch):
                img[c, y1:y2, x1:x2] = q[4 + (c % 3)]
        noise = hash_uniform(f"img:noise:{gi}", ch * h * w, seed).reshape(ch, h, w)
        img += np.float32(-0.08) + noise * np.float32(0.16)
        np.clip(img, 0.0, 1.0, out=out[b])
    return torch.from_numpy(out)


# --------------------------------------------------------------------------------------------------------------------
# Weight recipe.  Tuned (oracle, this container) so that a few % of anchors / queries exceed conf=0.25 and the
# DFL distributions are non-degenerate; constants are part of the fixture definition - do not change without
# regenerating tests/golden.
# --------------------------------------------------------------------------------------------------------------------
CONV_GAIN2 = 6.5  # conv weights ~ U(-a, a), a = sqrt(CONV_GAIN2 / fan_in): keeps SiLU activations O(0.3) at any depth
ATTN_GAIN2 = 1.0  # MHSA q/k/v 1x1 convs (block.py:6020-6062): energy = q^T k is NOT scaled by 1/sqrt(d)
RES_GAIN2 = 0.3  # last conv of a residual branch (Bottleneck.cv2 with add=True): damped, else x + f(x) chains explode
CLS_BIAS_SPREAD = 0.5
# Detect-head recipe per model family: (final cls 1x1 weight gain, mean final cls bias, final box 1x1 weight gain).
# The head's input magnitude differs per architecture, so these are tuned (oracle, build container) until ~2 % of
# anchors exceed conf=0.25 with distinct scores and the DFL bins are peaky-but-not-one-hot (logit std 3-4).
HEAD_RECIPE = {
    "default": (5.0, -4.4, 5.0),
    "yolov8n": (5.0, -4.4, 5.0),
    "yolov8s": (1.0, -6.0, 2.0),
    "yolov3-tiny": (0.6, -5.6, 0.7),
    "yolov5-BoT3": (0.8, -5.0, 0.25),
}
RTDETR_SCORE_BIAS = -6.5
RTDETR_SCORE_GAIN = 1.5

# --------------------------------------------------------------------------------------------------------------------
# "smooth" families (`family="smooth:<model>"`, round 3): the same hash draws with a recipe under which bf16 arithmetic
# reproduces the f32 DETECTIONS, not just the head outputs.  What breaks set agreement with the default recipe is not the size
# of the bf16 error (a few 1e-3 in score, < 1 px) but the structure of a random-weight head: neighbouring anchors predict
# large, heavily overlapping boxes with nearly equal scores, so a 1e-3 score change flips which of them survives NMS and the
# surviving box jumps by a stride.  A trained head does not do that (neighbours regress to the same object box); the smooth
# recipe removes the competition instead:
#   * DFL bias falls linearly with the bin index (SMOOTH_DFL_SLOPE per bin), so the expected side distances are 0.3 - 2.5
#     cells and boxes of adjacent anchors overlap with IoU < 0.7: NMS suppresses almost nothing and the detection set is the
#     threshold set;
#   * conv gain a little below the default (perturbations decay instead of growing through the 20+ layers), moderate head
#     gains;
#   * the class bias puts ~1 % of the anchors above conf = 0.25 (fewer threshold-straddling anchors per image);
#   * conv weights are the default draws rounded to bf16 and every BatchNorm has scale exactly 1 (gamma = 1, var = 0.999,
#     eps = 1e-3), so the BN-folded weights ARE bf16 numbers: the bf16 pipeline multiplies exactly the weights the f32
#     reference multiplies (what a checkpoint stored in half precision gives) and the comparison isolates the rounding of
#     the activations - the weight-rounding error has its own per-kernel tests (tests/test_hip_ops.py, bf16_weight_oracle).
# Per model: (conv gain^2, final cls 1x1 weight gain, mean final cls bias, final box 1x1 weight gain).
SMOOTH_DFL_SLOPE = 0.8
# Tuned with tools/experiments/smooth_scan.py (oracle f32 vs a bf16 emulation of it, CPU): the class bias is the 1 % quantile of
# the per-anchor best logit of the two golden images; constants are part of the fixture definition (tests/golden/e2e_*_smooth.npz).
SMOOTH_RECIPE = {
    "default": (6.5, 4.0, -4.0, 1.5),
    "yolov8n": (6.5, 4.0, -4.09, 1.5),
    "yolov8s": (6.5, 4.0, -3.08, 1.5),
    "yolov3-tiny": (5.5, 4.0, -3.16, 1.5),
    "yolov5-BoT3": (6.5, 4.0, -3.75, 1.5),
}


def _fan_in(shape) -> int:
    return int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])


def procedural_tensor(key: str, ref: torch.Tensor, kind: str, seed: int = 0, residual_tail: bool = False,
                      family: str = "default") -> torch.Tensor | None:
    """Procedural replacement for state_dict entry `key` (None = keep the module's own value).

    kind: "norm" (BatchNorm2d / LayerNorm owner), "conv" (Conv2d), "linear" (Linear / MultiheadAttention / other).
    """
    shape = tuple(ref.shape)
    leaf = key.rsplit(".", 1)[-1]
    smooth = family.startswith("smooth:")
    conv_gain2 = CONV_GAIN2
    if smooth:
        conv_gain2, cls_w_gain, cls_bias_shift, box_w_gain = SMOOTH_RECIPE.get(family[7:], SMOOTH_RECIPE["default"])
    else:
        cls_w_gain, cls_bias_shift, box_w_gain = HEAD_RECIPE.get(family, HEAD_RECIPE["default"])
    if not ref.dtype.is_floating_point:  # num_batches_tracked
        return None
    if key.endswith("dfl.conv.weight"):  # DFL expectation weights arange(16): block.py:245-248
        return None
    if "sampling_offsets.bias" in key:  # deterministic grid init: transformer.py:491-502
        return None
    if kind == "norm":
        if smooth and leaf in ("running_var", "weight") and len(shape) == 1:
            # BatchNorm scale gamma / sqrt(var + eps) = 1 / sqrt(0.999 + 1e-3) = 1 exactly: folding leaves the (bf16-exact)
            # conv weights untouched, see SMOOTH_RECIPE
            return torch.full(shape, 0.999 if leaf == "running_var" else 1.0)
        if leaf == "running_var":
            return uniform(key, shape, 0.5, 1.5, seed)
        if leaf == "weight":
            return uniform(key, shape, 0.5, 1.5, seed)
        return uniform(key, shape, -0.3, 0.3, seed)  # bias, running_mean
    # Detect final 1x1 convs (model.N.cv2.l.2 / cv3.l.2): head.py:95-100
    parts = key.split(".")
    det_final = len(parts) >= 5 and parts[-4] in ("cv2", "cv3") and parts[-2] == "2" and parts[-3].isdigit()
    if len(shape) >= 2:  # conv / linear / in_proj weights
        g2 = 3.0
        if kind == "conv" and not det_final:
            g2 = RES_GAIN2 if residual_tail else conv_gain2
            if parts[-2] in ("query", "key", "value"):  # MHSA 1x1 convs: keep the unscaled q^T k energies O(1)
                g2 = ATTN_GAIN2
        a = math.sqrt(g2 / _fan_in(shape))
        gain = 1.0
        if det_final:
            gain = box_w_gain if parts[-4] == "cv2" else cls_w_gain
        if "score_head" in key:
            gain = RTDETR_SCORE_GAIN
        if "sampling_offsets" in key:
            gain = 0.5
        w = uniform(key, shape, -a * gain, a * gain, seed)
        if smooth and kind == "conv":  # weights a bf16 model stores exactly (round-to-nearest-even of the same draws)
            w = w.to(torch.bfloat16).to(torch.float32)
        return w
    if det_final and parts[-4] == "cv3":
        return uniform(key, shape, cls_bias_shift - CLS_BIAS_SPREAD, cls_bias_shift + CLS_BIAS_SPREAD, seed)
    if det_final and parts[-4] == "cv2":
        if smooth:  # 4 sides x reg_max bins: the bias of bin k falls with k (small expected distances, see SMOOTH_RECIPE)
            k = torch.arange(shape[0], dtype=torch.float32) % 16
            return uniform(key, shape, -0.2, 0.2, seed) - SMOOTH_DFL_SLOPE * k
        return uniform(key, shape, 0.5, 1.5, seed)
    if "score_head" in key:
        return uniform(key, shape, RTDETR_SCORE_BIAS - 0.5, RTDETR_SCORE_BIAS + 0.5, seed)
    return uniform(key, shape, -0.1, 0.1, seed)


def model_family(model: torch.nn.Module) -> str:
    """Key into HEAD_RECIPE: the YAML stem the model was built from ('yolov8n', 'yolov5-BoT3', ...)."""
    from pathlib import Path

    y = getattr(model, "yaml", None) or {}
    return Path(str(y.get("yaml_file", "default"))).stem


@torch.no_grad()
def apply_procedural_weights(model: torch.nn.Module, seed: int = 0, family: str | None = None) -> torch.nn.Module:
    """Overwrite every floating-point parameter/buffer of `model` in place with its procedural value.

    Keys are the state_dict names, so an oracle model, the imported reference model and the HIP
    product model built from the same YAML receive identical bits.
    """
    family = family or model_family(model)
    tails = set()  # names of conv modules that end a residual branch
    for mname, m in model.named_modules():
        if getattr(m, "add", False) is True and hasattr(m, "cv2"):  # Bottleneck with shortcut (block.py:644-668)
            tails.add(f"{mname}.cv2.conv")
    for mname, m in model.named_modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.LayerNorm)):
            kind = "norm"
        elif isinstance(m, torch.nn.Conv2d):
            kind = "conv"
        else:
            kind = "linear"
        own = list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False))
        for pname, v in own:
            key = f"{mname}.{pname}" if mname else pname
            t = procedural_tensor(key, v, kind, seed, residual_tail=mname in tails, family=family)
            if t is not None:
                v.copy_(t.to(v.dtype))
    return model


def synthetic_labels(batch: int, seed: int = 0, first: int = 0):
    """Synthetic training labels: per image k ~ U{1..8} boxes, cls ~ U{0..79},
    cx, cy ~ U(0.1, 0.9), w, h ~ U(0.05, 0.4), normalised xywh - the reference's batch dict layout
    {"batch_idx" (n,), "cls" (n,), "bboxes" (n,4)} (utils/loss.py:486). Image `first + i` always gets the same labels,
    so a rank's shard of a global batch is reproducible anywhere."""
    import torch
    bi, cl, bb = [], [], []
    for i in range(batch):
        u = hash_uniform(f"labels:{seed}:{first + i}", 1 + 8 * 5)
        k = 1 + int(u[0] * 8)
        r = u[1:].reshape(8, 5)[:k]
        bi.append(np.full(k, i, dtype=np.float32))
        cl.append(np.floor(r[:, 0] * 80).astype(np.float32))
        box = np.stack([0.1 + 0.8 * r[:, 1], 0.1 + 0.8 * r[:, 2], 0.05 + 0.35 * r[:, 3], 0.05 + 0.35 * r[:, 4]], 1)
        bb.append(box.astype(np.float32))
    return {"batch_idx": torch.from_numpy(np.concatenate(bi)), "cls": torch.from_numpy(np.concatenate(cl)),
            "bboxes": torch.from_numpy(np.concatenate(bb))}
