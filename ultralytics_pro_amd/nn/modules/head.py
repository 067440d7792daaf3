"""Detection heads with the reference's operator API (ultralytics/nn/modules/head.py), on HIP kernels."""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R
from .block import DFL
from .conv import Conv, _HipConvMixin, hip_conv2d

__all__ = ("Detect",)


class Detect(nn.Module, _HipConvMixin):
    """YOLO Detect head (head.py:28-191), legacy v3/v5/v8 class branch (:98-100).

    eval forward returns `(y, x)` like the reference: y = (B, 4+nc, A) float32 decoded boxes (xywh, pixels) and class
    probabilities, x = the per-level raw maps (NHWC views, logical (B, 4*reg_max+nc, H, W)).
    Per level the box and class branches write the two channel slices of one (B,H,W,144) buffer (the reference's
    `torch.cat((cv2(x), cv3(x)), 1)`, head.py:122) and one `upa_detect_decode` launch performs
    DFL + dist2bbox + stride scaling + sigmoid (head.py:151-169).
    """

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False
    xyxy = False

    def __init__(self, nc: int = 80, ch: tuple = ()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        if not self.legacy:
            raise L.UpaError("Detect(legacy=False) (DWConv class branch, v11+) is outside the hot-path scope (SURVEY §2)")
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    def _branch(self, seq: nn.Sequential, x: torch.Tensor, out: torch.Tensor) -> None:
        t = seq[1](seq[0](x))
        pk = self._packed(seq[2], None, x.device, x.dtype, False)
        hip_conv2d(t, pk, 1, 0, L.ACT_NONE, out=out)

    def forward(self, x):
        if self.training:
            raise L.UpaError("training-mode Detect is not on the HIP path yet (SURVEY §8f rank 2)")
        x = [R.to_nhwc(t, t.dtype) for t in x]
        nb = 4 * self.reg_max
        raw = []
        for i in range(self.nl):
            n, _, h, w = x[i].shape
            buf = R.alloc_nhwc(n, self.no, h, w, x[i].dtype, x[i].device, key=(id(self), "raw", i))
            self._branch(self.cv2[i], x[i], buf[:, :nb])
            self._branch(self.cv3[i], x[i], buf[:, nb:])
            raw.append(buf)
        y = self._inference(raw)
        return y if self.export else (y, raw)

    def _inference(self, x: list[torch.Tensor]) -> torch.Tensor:
        """Decode boxes and class probabilities of all levels into (B, 4+nc, A) float32 (head.py:151-169)."""
        nb = 4 * self.reg_max
        n = x[0].shape[0]
        a_total = sum(int(t.shape[2]) * int(t.shape[3]) for t in x)
        y = R.alloc_plain((n, 4 + self.nc, a_total), torch.float32, x[0].device, key=(id(self), "y"))
        a0 = 0
        for i, t in enumerate(x):
            vb, vc = R.view_of(t[:, :nb]), R.view_of(t[:, nb:])
            L.check(L.lib().upa_detect_decode(vb.ptr, vb.ld, vc.ptr, vc.ld, vb.n, vb.h, vb.w, self.reg_max, self.nc,
                                              float(self.stride[i]), y.data_ptr(), a_total, a0, vb.dtype,
                                              L.current_stream(t.device)), "detect_decode")
            a0 += vb.h * vb.w
        return y

    def bias_init(self):
        """Initialize Detect() biases, requires stride availability (head.py:171-178)."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)
