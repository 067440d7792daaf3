"""`LetterBox` with the reference's constructor and call signature (ultralytics/data/augment.py:1544-1700), executed by
`upa_letterbox_u8`: the frame stays on the GPU as uint8 HWC from the decoder to the stem conv, which reads BGR uint8
directly (BGR->RGB, HWC->CHW, /255 fused: csrc/stem.hip) - no float image, no host resize.

The geometry - scale ratio, `round()`-ed unpadded size, padding split with the reference's `round(d -/+ 0.1)` - is computed
here with the same Python float arithmetic as LetterBox.__call__ (augment.py:1640-1668); the pixels come from the kernel
(cv2.resize INTER_LINEAR fixed-point arithmetic + cv2.copyMakeBorder).  No CPU fallback: a CPU tensor raises.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from ..engine import runtime as R

__all__ = ("LetterBox",)


class LetterBox:
    """Resize image and padding for detection (augment.py:1544).  Labels with `instances` (training-time augmentation)
    are outside the hot path: only the image is transformed; `ratio_pad` bookkeeping is kept."""

    def __init__(self, new_shape=(640, 640), auto: bool = False, scale_fill: bool = False, scaleup: bool = True,
                 center: bool = True, stride: int = 32, padding_value: int = 114, interpolation: int = 1):
        if interpolation != 1:
            raise L.UpaError("HIP LetterBox implements cv2.INTER_LINEAR (= 1), the reference's default")
        self.new_shape = new_shape
        self.auto = auto
        self.scale_fill = scale_fill
        self.scaleup = scaleup
        self.stride = stride
        self.center = center
        self.padding_value = padding_value
        self.interpolation = interpolation

    def geometry(self, shape, new_shape=None):
        """((new_w, new_h), (top, bottom, left, right), (ratio_w, ratio_h)) for a (h, w) frame (augment.py:1640-1668)."""
        new_shape = self.new_shape if new_shape is None else new_shape
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:  # never enlarge a frame, shrink only
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = round(shape[1] * r), round(shape[0] * r)
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:  # pad just up to the next multiple of the stride
            dw, dh = np.mod(dw, self.stride), np.mod(dh, self.stride)
        elif self.scale_fill:  # fill the target exactly: independent x / y ratios, no border
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = round(dh - 0.1) if self.center else 0, round(dh + 0.1)
        left, right = round(dw - 0.1) if self.center else 0, round(dw + 0.1)
        return (int(new_unpad[0]), int(new_unpad[1])), (int(top), int(bottom), int(left), int(right)), ratio

    def __call__(self, labels=None, image=None, slot: int = 0):
        """image: uint8 CUDA tensor (h, w, 3) or a batch (n, h, w, 3) of equally sized frames (rows may be strided, e.g. a crop
        of a larger frame).  Returns the letterboxed uint8 tensor of the same rank, or `labels` updated like the reference.

        Inside a static-buffer context (`runtime.static_buffers`, `compile()`) the result is a pooled buffer keyed by
        (this instance, slot, output shape): two calls with the same slot return the SAME storage.  The reference idiom
        `[letterbox(image=x) for x in ims]` (predictor.py:147) must therefore pass `slot=i` there (or hand over the frames as
        one (n, h, w, 3) batch); outside such a context every call allocates."""
        if labels is None:
            labels = {}
        img = labels.get("img") if image is None else image
        if "instances" in labels:
            raise L.UpaError("LetterBox with `instances` (training augmentation) is outside the hot-path scope")
        if not torch.is_tensor(img):
            raise L.UpaError("HIP LetterBox takes a uint8 CUDA tensor (h, w, 3) / (n, h, w, 3); there is no CPU path")
        L.require_gpu(img, "LetterBox")
        single = img.dim() == 3
        x = img.unsqueeze(0) if single else img
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3 or x.stride(3) != 1 or x.stride(2) != 3:
            raise L.UpaError("LetterBox expects uint8 frames with interleaved channels: (n, h, w, 3), pixel stride 3 bytes")
        n, h0, w0, _ = x.shape
        new_shape = labels.pop("rect_shape", self.new_shape)
        (nw, nh), (top, bottom, left, right), ratio = self.geometry((h0, w0), new_shape)
        H, W = nh + top + bottom, nw + left + right
        out = R.alloc_plain((n, H, W, 3), torch.uint8, x.device, key=(id(self), "letterbox", int(slot), n, H, W))
        L.check(L.lib().upa_letterbox_u8(x.data_ptr(), n, h0, w0, x.stride(0) if n > 1 else h0 * x.stride(1), x.stride(1),
                                         out.data_ptr(), H, W, nh, nw, top, left, int(self.padding_value),
                                         L.current_stream(x.device)), "letterbox_u8")
        res = out[0] if single else out
        if labels.get("ratio_pad"):
            labels["ratio_pad"] = (labels["ratio_pad"], (left, top))  # for evaluation
        if len(labels):
            labels["img"] = res
            labels["resized_shape"] = new_shape if not isinstance(new_shape, int) else (new_shape, new_shape)
            return labels
        return res
