"""Image-space helpers either side of the hot path, reference signatures (ultralytics/utils/ops.py)."""

from __future__ import annotations

import math

import torch

from .. import _lib as L


def scale_boxes(img1_shape, boxes: torch.Tensor, img0_shape, ratio_pad=None, padding: bool = True, xywh: bool = False):
    """Rescale xyxy boxes (in place, GPU) from the letterboxed `img1_shape` (h, w) to the original `img0_shape` and clip
    (utils/ops.py:102-152 + clip_boxes :154-178).  `boxes` is (..., k>=4) float32 with the box in the first 4 columns."""
    if xywh:
        raise L.UpaError("scale_boxes(xywh=True) is outside the hot-path scope")
    L.require_gpu(boxes, "scale_boxes")
    if boxes.dtype != torch.float32 or (boxes.numel() and not boxes.is_contiguous()):
        raise L.UpaError("scale_boxes expects contiguous float32 rows")
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
        pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    else:
        gain = ratio_pad[0][0]
        pad_x, pad_y = ratio_pad[1]
    rows = boxes.numel() // boxes.shape[-1] if boxes.numel() else 0
    if rows:
        L.check(L.lib().upa_scale_boxes(boxes.data_ptr(), rows, boxes.shape[-1], float(gain), float(pad_x), float(pad_y),
                                        int(bool(padding)), float(img0_shape[1]), float(img0_shape[0]),
                                        L.current_stream(boxes.device)), "scale_boxes")
    return boxes


def make_divisible(x, divisor):
    """Nearest multiple of divisor not below x (utils/ops.py:137-150)."""
    return math.ceil(x / divisor) * divisor
