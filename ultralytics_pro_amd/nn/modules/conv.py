"""Convolution modules with the reference's operator API (ultralytics/nn/modules/conv.py), dispatching into HIP.

`Conv(c1, c2, k, s, p, g, d, act)` keeps the reference constructor, attribute names (`conv`, `bn`, `act`) and therefore
state_dict keys (conv.py:147-197).  The nn.Conv2d / nn.BatchNorm2d children are parameter containers only: `forward`
folds BN (utils/torch_utils.py:236-266), packs the weights once per dtype into MFMA fragment order and calls
`upa_conv2d_bias_act` / `upa_conv2d_stem_nchw`.  There is no torch fallback: CPU tensors raise.
"""

from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R

__all__ = ("autopad", "Conv", "Concat", "hip_conv2d", "PackedConv", "version_key", "bump_weights_generation")


def autopad(k, p=None, d=1):
    """Pad to 'same' shape outputs (conv.py:64-70)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d | None):
    """(W', b') of utils/torch_utils.py:236-266 in the same operation order, as float32 CPU tensors."""
    w = conv.weight.detach().float().cpu()
    b = None if conv.bias is None else conv.bias.detach().float().cpu()
    if bn is None:
        return w, (torch.zeros(w.shape[0]) if b is None else b)
    g, beta = bn.weight.detach().float().cpu(), bn.bias.detach().float().cpu()
    mu, var = bn.running_mean.detach().float().cpu(), bn.running_var.detach().float().cpu()
    w_bn = torch.diag(g.div(torch.sqrt(bn.eps + var)))
    wf = torch.mm(w_bn, w.view(w.shape[0], -1)).view(w.shape)
    b_conv = torch.zeros(w.shape[0]) if b is None else b
    bf = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + (beta - g.mul(mu).div(torch.sqrt(var + bn.eps)))
    return wf, bf


class PackedConv:
    """Device-resident packed weights of one conv for one dtype."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor, k: int, device, dtype: torch.dtype, stem: bool):
        self.cout, self.cin, self.k = int(w.shape[0]), int(w.shape[1]), k
        self.bias = b.contiguous().to(device)
        self.stem = stem
        if stem:  # first layer reads NCHW directly; f32 weights repacked [tap][ci][cout padded to 16]
            nbytes = L.lib().upa_stem_packed_weight_bytes(self.cout, self.cin, k)
            host = torch.empty(nbytes // 4, dtype=torch.float32)
            wc = w.contiguous()
            L.check(L.lib().upa_pack_stem_weight(wc.data_ptr(), self.cout, self.cin, k, host.data_ptr()), "pack_stem")
            self.w = host.to(device)
            return
        code = L.dtype_code(dtype)
        nbytes = L.lib().upa_conv_packed_weight_bytes(self.cout, self.cin, k, code)
        host = torch.empty(nbytes, dtype=torch.uint8)
        wc = w.contiguous()
        L.check(L.lib().upa_pack_conv_weight(wc.data_ptr(), self.cout, self.cin, k, code, host.data_ptr()), "pack_conv")
        self.w = host.to(device)
        if R._TRACE_POOL:
            import sys
            for nm, t in (("w", self.w), ("bias", self.bias)):
                print(f"[pool] {t.data_ptr():#x} +{t.numel() * t.element_size():#x} end {t.data_ptr() + t.numel() * t.element_size():#x} "
                      f"packed {nm} cout={self.cout} cin={self.cin} k={k}", file=sys.stderr, flush=True)


class VirtualUpsample:
    """An Upsample(2x nearest) that was NOT launched: `src` is the half-resolution tensor, `channels` the leading channels
    of the concat buffer it would have filled, `materialize()` launches it after all.  The 1x1 conv that consumes the
    Concat reads `src` at (y/2, x/2) instead (`upa_conv1x1_upcat`)."""

    def __init__(self, src: torch.Tensor, channels: int, materialize):
        self.src, self.channels, self._materialize, self.done = src, int(channels), materialize, False

    def materialize(self):
        """Write the upsampled channels after all (once): later consumers then read the buffer like any other."""
        if not self.done:
            self._materialize()
            self.done = True


def hip_conv2d(x: torch.Tensor, pk: PackedConv, stride: int, pad: int, act: int, out: torch.Tensor | None = None,
               residual: torch.Tensor | None = None, out_dtype: torch.dtype | None = None, key=None,
               up: VirtualUpsample | None = None) -> torch.Tensor:
    """y = act(conv(x) + b) (+ residual) on the HIP path. x: NHWC view, or raw NCHW model input when pk.stem."""
    L.require_gpu(x, "conv2d")
    lib = L.lib()
    stream = L.current_stream(x.device)
    if pk.stem:
        if x.dtype == torch.uint8:  # raw (N, H, W, 3) BGR frames: BGR->RGB, HWC->CHW, /255 fused into the stem load
            if not (x.is_contiguous() and x.dim() == 4 and x.shape[-1] == 3):
                raise L.UpaError("uint8 input must be a contiguous (N, H, W, 3) BGR batch")
            n, h, w, cin = x.shape
            odt = out_dtype or torch.float32
            xcode = L.UPA_U8_BGR_HWC
        else:
            if not (x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)):
                raise L.UpaError("first-layer input must be a contiguous NCHW float32/bfloat16 tensor")
            n, cin, h, w = x.shape
            odt = out_dtype or x.dtype
            xcode = L.dtype_code(x.dtype)
        oh, ow = (h + 2 * pad - pk.k) // stride + 1, (w + 2 * pad - pk.k) // stride + 1
        y = out if out is not None else R.alloc_nhwc(n, pk.cout, oh, ow, odt, x.device, key)
        vy = R.view_of(y)
        L.check(lib.upa_conv2d_stem_nchw(x.data_ptr(), xcode, n, cin, h, w, pk.w.data_ptr(),
                                         pk.bias.data_ptr(), vy.ptr, pk.cout, vy.ld, pk.k, stride, pad, act, vy.dtype,
                                         R.opts_ptr(), stream), "conv2d_stem")
        return y
    vx = R.view_of(x)
    oh, ow = (vx.h + 2 * pad - pk.k) // stride + 1, (vx.w + 2 * pad - pk.k) // stride + 1
    y = out if out is not None else R.alloc_nhwc(vx.n, pk.cout, oh, ow, x.dtype, x.device, key)
    vy = R.view_of(y)
    if vy.dtype != vx.dtype or (vy.n, vy.h, vy.w, vy.c) != (vx.n, oh, ow, pk.cout):
        raise L.UpaError(f"conv2d: bad output view {tuple(y.shape)} for input {tuple(x.shape)}")
    rp, rld = None, 0
    if residual is not None:
        vr = R.view_of(residual)
        if (vr.n, vr.h, vr.w, vr.c, vr.dtype) != (vy.n, vy.h, vy.w, vy.c, vy.dtype):
            raise L.UpaError("conv2d: residual shape/dtype mismatch")
        rp, rld = vr.ptr, vr.ld
    if up is not None and not up.done:
        # the leading channels of x were never written: read them from the half-resolution tensor, or write them now
        vu = R.view_of(up.src)
        rc = L.UPA_EUNSUPPORTED
        if pk.k == 1 and stride == 1 and pad == 0 and residual is None and vu.dtype == vx.dtype and vu.c == up.channels \
                and (vu.n, 2 * vu.h, 2 * vu.w) == (vx.n, vx.h, vx.w):
            rc = lib.upa_conv1x1_upcat(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, vu.ptr, vu.c, vu.ld, pk.w.data_ptr(),
                                       pk.bias.data_ptr(), vy.ptr, pk.cout, vy.ld, act, vx.dtype, R.opts_ptr(), stream)
        if rc == 0:
            return y
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "conv1x1_upcat")
        up.materialize()
    L.check(lib.upa_conv2d_bias_act(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, pk.w.data_ptr(), pk.bias.data_ptr(), vy.ptr,
                                    pk.cout, vy.ld, rp, rld, pk.k, stride, pad, act, vx.dtype, R.opts_ptr(), stream), "conv2d")
    return y


_WEIGHTS_GEN = [0]


def bump_weights_generation() -> int:
    """Invalidate every packed-weight cache of the process.  For writers that change parameters through raw device
    pointers, where neither `data_ptr()` nor `_version` moves: the HIP optimizer (`upa_sgd_nesterov_ema` / `upa_ema_update`
    on the trainer's flat buffers) calls this after every step."""
    _WEIGHTS_GEN[0] += 1
    return _WEIGHTS_GEN[0]


def version_key(*tensors):
    """Identity + in-place version of parameter tensors + the process-wide weights generation: changes on
    `load_state_dict` / `copy_` / torch optimizer steps (`_version` bump), on `.to(device)` / `.data = ...` (new storage)
    and on `bump_weights_generation()` (raw-pointer writers: engine/trainer.py), so a cache keyed on it cannot go stale."""
    return (_WEIGHTS_GEN[0],) + tuple(None if t is None else (t.data_ptr(), t._version, t.device.type) for t in tensors)


class _HipConvMixin:
    """Lazily folded / packed weights, cached per (conv, device, dtype) and validated against the parameters' storage and
    in-place version on every use, so `load_state_dict`, `load_weights`, `.to()`, `train()` or an optimizer step can never
    leave a stale packed copy behind (a compiled hipGraph still bakes the packed pointers in: recompile after loading)."""

    def _packed(self, conv: nn.Conv2d, bn, device, dtype, stem: bool, pad_cout: int = 0) -> PackedConv:
        """`pad_cout` > out_channels appends zero filters / zero biases (a Detect class branch whose nc is not a multiple
        of the 16-byte store width writes its padded row; the extra channels are never read)."""
        cache = self.__dict__.setdefault("_pk_cache", {})
        key = (id(conv), str(device), dtype, stem, pad_cout)
        ver = version_key(conv.weight, conv.bias, *(() if bn is None else (bn.weight, bn.bias, bn.running_mean,
                                                                           bn.running_var))) + ((bn.eps,) if bn is not None else ())
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        k = conv.kernel_size[0]
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.groups != 1 or conv.dilation != (1, 1) or \
                conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]:
            raise L.UpaError(f"HIP conv supports square kernels, groups=1, dilation=1 only, got {conv}")
        w, b = fold_bn(conv, bn)
        if pad_cout > w.shape[0]:
            w = torch.cat([w, torch.zeros(pad_cout - w.shape[0], *w.shape[1:])], 0)
            b = torch.cat([b, torch.zeros(pad_cout - b.shape[0])], 0)
        pk = PackedConv(w, b, k, device, dtype, stem)
        cache[key] = (ver, pk)
        return pk

    def _packed_stack(self, pairs, device, dtype) -> PackedConv:
        """ONE packed conv for several 1x1 convs that read the same input (C3's cv1 / cv2, block.py:509-532; MHSA's query / key /
        value, block.py:6036-6062): the BN-folded filters and biases stacked along the output channels in the given order, so that a
        single launch reads the input once and writes every branch's channel slice of one buffer.  `pairs`: [(nn.Conv2d, bn | None)];
        same kernel size / stride / padding.  Cached like `_packed`, keyed on every member's parameters."""
        cache = self.__dict__.setdefault("_pk_cache", {})
        key = ("stack",) + tuple(id(c) for c, _ in pairs) + (str(device), dtype)
        ver = ()
        for conv, bn in pairs:
            ver += version_key(conv.weight, conv.bias, *(() if bn is None else (bn.weight, bn.bias, bn.running_mean, bn.running_var)))
            ver += ((bn.eps,) if bn is not None else ())
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        c0 = pairs[0][0]
        for conv, _ in pairs:
            if (conv.kernel_size, conv.stride, conv.padding, conv.in_channels) != (c0.kernel_size, c0.stride, c0.padding, c0.in_channels) \
                    or conv.groups != 1 or conv.dilation != (1, 1):
                raise L.UpaError("stacked convs must share kernel size / stride / padding / input channels (groups = 1)")
        folded = [fold_bn(conv, bn) for conv, bn in pairs]
        w = torch.cat([f[0] for f in folded], 0)
        b = torch.cat([f[1] for f in folded], 0)
        pk = PackedConv(w, b, c0.kernel_size[0], device, dtype, False)
        cache[key] = (ver, pk)
        return pk

    def invalidate_packed(self):
        self.__dict__.pop("_pk_cache", None)


def _is_model_input(x: torch.Tensor, cin: int) -> bool:
    """A raw image batch - NCHW float (<= 4 channels) or uint8 (N, H, W, 3) BGR frames: handled by the stem kernel."""
    if x.dtype == torch.uint8:
        return x.dim() == 4 and x.shape[-1] == 3 and cin == 3
    return cin <= 4 and x.dim() == 4 and x.is_contiguous() and (x.shape[2] > 1 or x.shape[3] > 1)


class Conv(nn.Module, _HipConvMixin):
    """Standard convolution: conv2d(bias=False) -> BatchNorm2d -> SiLU (conv.py:147-197)."""

    default_act = nn.SiLU()  # default activation

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        self.compute_dtype = None  # set by the model for the first layer (stem output dtype)

    def _act_code(self) -> int:
        if isinstance(self.act, nn.SiLU):
            return L.ACT_SILU
        if isinstance(self.act, nn.ReLU):
            return L.ACT_RELU
        if isinstance(self.act, nn.Identity):
            return L.ACT_NONE
        raise L.UpaError(f"activation {self.act} has no HIP epilogue")

    def forward(self, x, out=None, residual=None, up=None):
        """act(bn(conv(x))) with BN folded into the HIP conv epilogue (conv.py:177-197 compute the same function)."""
        if self.training:
            raise L.UpaError("training-mode Conv (batch-statistics BN) is not on the HIP path yet (SURVEY §8f rank 2)")
        stem = _is_model_input(x, self.conv.in_channels)
        # raw uint8 frames with no compute dtype chosen: float32 activations (the reference's `im.float()`, predictor.py:169)
        dt = (self.compute_dtype or (torch.float32 if x.dtype == torch.uint8 else x.dtype)) if stem else x.dtype
        pk = self._packed(self.conv, getattr(self, "bn", None), x.device, dt, stem)
        if up is not None and stem:
            up.materialize()
            up = None
        return hip_conv2d(x, pk, self.conv.stride[0], self.conv.padding[0], self._act_code(), out=out, residual=residual,
                          out_dtype=dt, key=(id(self), "y"), up=up)

    forward_fuse = forward  # BN is always folded on the HIP path (tasks.py:1134 rebinding is a no-op here)

    def forward_pool2(self, x):
        """MaxPool2d(2, 2, 0)(self(x)) as ONE launch when the layer has the fused form (bf16, k 3 s 1 p 1, SiLU: `upa_conv2d_pool2`,
        or the first layer: `upa_conv2d_stem_nchw_pool2`), else None - the caller then runs the two rows separately.  The result is
        bit-identical either way; the full-resolution activation is never written (yolov3-tiny.yaml rows 0-7)."""
        if self.training or not isinstance(self.act, nn.SiLU):
            return None
        cv = self.conv
        if (cv.kernel_size, cv.stride, cv.padding) != ((3, 3), (1, 1), (1, 1)):
            return None
        stem = _is_model_input(x, cv.in_channels)
        dt = (self.compute_dtype or (torch.float32 if x.dtype == torch.uint8 else x.dtype)) if stem else x.dtype
        if dt != torch.bfloat16:
            return None
        L.require_gpu(x, "conv2d_pool2")
        lib, stream = L.lib(), L.current_stream(x.device)
        pk = self._packed(cv, getattr(self, "bn", None), x.device, dt, stem)
        if stem:
            if x.dtype == torch.uint8:
                n, h, w, cin = x.shape
                xcode = L.UPA_U8_BGR_HWC
            else:
                if not (x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)):
                    return None
                n, cin, h, w = x.shape
                xcode = L.dtype_code(x.dtype)
            if h % 2 or w % 2:
                return None
            y = R.alloc_nhwc(n, pk.cout, h // 2, w // 2, dt, x.device, key=(id(self), "ypool"))
            vy = R.view_of(y)
            rc = lib.upa_conv2d_stem_nchw_pool2(x.data_ptr(), xcode, n, cin, h, w, pk.w.data_ptr(), pk.bias.data_ptr(), vy.ptr, pk.cout,
                                                vy.ld, 3, 1, 1, L.ACT_SILU, vy.dtype, R.opts_ptr(), stream)
        else:
            x = R.to_nhwc(x, x.dtype)
            vx = R.view_of(x)
            if vx.h % 2 or vx.w % 2:
                return None
            y = R.alloc_nhwc(vx.n, pk.cout, vx.h // 2, vx.w // 2, x.dtype, x.device, key=(id(self), "ypool"))
            vy = R.view_of(y)
            rc = lib.upa_conv2d_pool2(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, pk.w.data_ptr(), pk.bias.data_ptr(), vy.ptr, pk.cout, vy.ld,
                                      3, 1, 1, L.ACT_SILU, vx.dtype, R.opts_ptr(), stream)
        if rc == 0:
            return y
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "conv2d_pool2")
        return None

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)


class Concat(nn.Module):
    """Concatenate a list of tensors along `dimension` (conv.py:850-875): channel-slice copies into one NHWC buffer.
    (Inside a planned model the producers write into the slices directly and no copy is launched.)"""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        if self.d != 1:
            raise L.UpaError("HIP Concat supports the channel dimension only")
        xs = list(x)
        n, _, h, w = xs[0].shape
        ctot = sum(int(t.shape[1]) for t in xs)
        y = R.alloc_nhwc(n, ctot, h, w, xs[0].dtype, xs[0].device, key=(id(self), "y"))
        c0 = 0
        for t in xs:
            c = int(t.shape[1])
            dst = y[:, c0:c0 + c]
            if not (t.data_ptr() == dst.data_ptr() and t.stride() == dst.stride()):
                vs, vd = R.view_of(t), R.view_of(dst)
                L.check(L.lib().upa_copy_view(vs.ptr, vs.n, vs.h, vs.w, vs.c, vs.ld, vd.ptr, vd.ld, vs.dtype,
                                              L.current_stream(t.device)), "copy_view")
            c0 += c
        return y
