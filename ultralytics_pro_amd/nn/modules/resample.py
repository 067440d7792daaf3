"""HIP stand-ins for the `nn.*` rows of the reference YAMLs (resolved by `getattr(torch.nn, name)` in
ultralytics/nn/tasks.py:2836-2842): nn.Upsample(None, 2, 'nearest'), nn.MaxPool2d(k, s, p), nn.ZeroPad2d([0,1,0,1]).
Same constructor signatures; forward runs on libupa_hip.so (no torch kernels)."""

from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R

__all__ = ("Upsample", "MaxPool2d", "ZeroPad2d")

_PAD_ATTR = "_upa_pad_br"


class Upsample(nn.Module):
    """nn.Upsample(size=None, scale_factor=2, mode='nearest') (yolov8.yaml:31,35)."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or float(scale_factor) != 2.0 or mode != "nearest":
            raise L.UpaError("HIP Upsample implements scale_factor=2, mode='nearest' (every reference YAML on the path)")
        self.size, self.scale_factor, self.mode = size, scale_factor, mode

    def forward(self, x, out=None):
        x = R.to_nhwc(x, x.dtype)
        v = R.view_of(x)
        y = out if out is not None else R.alloc_nhwc(v.n, v.c, 2 * v.h, 2 * v.w, x.dtype, x.device, key=(id(self), "y"))
        vy = R.view_of(y)
        L.check(L.lib().upa_upsample2x(v.ptr, v.n, v.h, v.w, v.c, v.ld, vy.ptr, vy.ld, v.dtype,
                                       L.current_stream(x.device)), "upsample2x")
        return y


class ZeroPad2d(nn.Module):
    """nn.ZeroPad2d([0, p, 0, p]) (yolov3-tiny.yaml row 11). Never materialised: the padding is handed to the MaxPool2d
    that follows it (`pad_br` of upa_maxpool2d); any other consumer refuses the tagged tensor."""

    def __init__(self, padding):
        super().__init__()
        p = list(padding) if isinstance(padding, (list, tuple)) else [padding] * 4
        if len(p) != 4 or p[0] != 0 or p[2] != 0 or p[1] != p[3]:
            raise L.UpaError("HIP ZeroPad2d implements right/bottom padding [0, p, 0, p] only")
        self.padding = tuple(p)

    def forward(self, x):
        x = R.to_nhwc(x, x.dtype)
        y = x[:]  # fresh tensor object on the same storage
        setattr(y, _PAD_ATTR, int(self.padding[1]))
        return y


class MaxPool2d(nn.Module):
    """nn.MaxPool2d(kernel_size, stride, padding) with -inf padding semantics (yolov3-tiny.yaml rows 1-12)."""

    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding

    def forward(self, x, out=None):
        pad_br = int(getattr(x, _PAD_ATTR, 0))
        if pad_br:
            x = x[:]
        x = R.to_nhwc(x, x.dtype)
        v = R.view_of(x)
        k, s, p = self.kernel_size, self.stride, self.padding
        oh = (v.h + pad_br + 2 * p - k) // s + 1
        ow = (v.w + pad_br + 2 * p - k) // s + 1
        y = out if out is not None else R.alloc_nhwc(v.n, v.c, oh, ow, x.dtype, x.device, key=(id(self), "y"))
        vy = R.view_of(y)
        L.check(L.lib().upa_maxpool2d(v.ptr, v.n, v.h, v.w, v.c, v.ld, vy.ptr, oh, ow, vy.ld, k, s, p, pad_br, v.dtype,
                                      L.current_stream(x.device)), "maxpool2d")
        return y
