"""Block modules with the reference's operator API (ultralytics/nn/modules/block.py), composed from HIP kernels.

Concatenations never copy: a block allocates its concat buffer once and every producing conv writes its channel slice
in place (`out=` views); Bottleneck residual adds are fused in the second conv's epilogue.
"""

from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R
from .conv import Conv, PackedConv, _HipConvMixin, fold_bn, hip_conv2d, version_key

__all__ = ("DFL", "SPPF", "C2f", "C3", "Bottleneck", "MHSA", "BottleneckTransformer", "BoT3")


class DFL(nn.Module):
    """Integral module of Distribution Focal Loss (block.py:232-253).  On the HIP path the softmax-expectation is part of
    `upa_detect_decode`; this module only carries the arange(c1) weights for the state_dict."""

    def __init__(self, c1: int = 16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        raise L.UpaError("DFL is fused into upa_detect_decode on the HIP path; call Detect instead")


class Bottleneck(nn.Module):
    """Standard bottleneck: x + cv2(cv1(x)) when shortcut and c1 == c2 (block.py:644-668)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    fuse_pair = True  # bf16: both convs as one kernel, the intermediate tile in LDS (upa_bottleneck_pair)

    def _pair_ok(self, x) -> bool:
        a, b = self.cv1.conv, self.cv2.conv
        return bool(self.fuse_pair and x.dtype == torch.bfloat16 and not self.training
                    and a.kernel_size == (3, 3) and b.kernel_size == (3, 3) and a.stride == (1, 1) and b.stride == (1, 1)
                    and a.padding == (1, 1) and b.padding == (1, 1) and a.groups == 1 and b.groups == 1
                    and a.in_channels == b.out_channels and b.in_channels == a.out_channels
                    and (a.in_channels, a.out_channels) in ((32, 32), (64, 64), (64, 32))  # (64, 32): the e = 0.5 darknet block
                    and isinstance(self.cv1.act, nn.SiLU) and isinstance(self.cv2.act, nn.SiLU)
                    and hasattr(self.cv1, "bn") and hasattr(self.cv2, "bn"))

    def forward(self, x, out=None):
        x = R.to_nhwc(x, x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        if self._pair_ok(x):
            n, c, h, w = x.shape
            y = out if out is not None else R.alloc_nhwc(n, c, h, w, x.dtype, x.device, key=(id(self), "y"))
            p1 = self.cv1._packed(self.cv1.conv, self.cv1.bn, x.device, x.dtype, False)
            p2 = self.cv2._packed(self.cv2.conv, self.cv2.bn, x.device, x.dtype, False)
            vx, vy = R.view_of(x), R.view_of(y)
            rc = L.lib().upa_bottleneck_pair_e(vx.ptr, vx.n, vx.h, vx.w, vx.c, self.cv1.conv.out_channels, vx.ld, p1.w.data_ptr(),
                                               p1.bias.data_ptr(), p2.w.data_ptr(), p2.bias.data_ptr(), vy.ptr, vy.ld, int(self.add),
                                               L.ACT_SILU, vx.dtype, R.opts_ptr(), L.current_stream(x.device))
            if rc == 0:
                return y
            if rc != L.UPA_EUNSUPPORTED:
                L.check(rc, "bottleneck_pair")
        return self.cv2(self.cv1(x), out=out, residual=x if self.add else None)


class C2f(nn.Module):
    """CSP bottleneck with 2 convolutions (block.py:457-488): cv1 -> chunk(2) -> n chained Bottlenecks -> cat -> cv2.
    One (2+n)c-channel buffer: cv1 fills [0,2c), bottleneck i reads slice [(1+i)c,(2+i)c) and writes [(2+i)c,(3+i)c)."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    # bf16 C2f(32, 32, n=1, shortcut) / C2f(64, 64, n=1|2) (upa_c2f_fused) and C2f(c1 % 64 == 0, 128, n=1|2) with 64-channel halves
    # (upa_c2f64_fused): the whole block as one kernel
    fuse_block = True

    def _form64(self) -> bool:
        return self.c == 64 and self.cv2.conv.out_channels == 128 and self.cv1.conv.in_channels % 64 == 0 and len(self.m) in (1, 2)

    def _form32up(self) -> bool:
        """C2f(64 k >= 128, 64, n = 1): the 32-channel kernel with cv1 streamed over 64-channel chunks (yolov8n model.15)."""
        return self.c == 32 and self.cv2.conv.out_channels == 64 and self.cv1.conv.in_channels % 64 == 0 and \
            128 <= self.cv1.conv.in_channels <= 512 and len(self.m) == 1

    def _fused(self, x, out, up=None):
        """One launch for the whole block when it has a fused form; None otherwise.  `up`: a conv.VirtualUpsample for the
        leading channels of x (the 64-channel form reads the half-resolution tensor itself)."""
        import ctypes as C
        nb = len(self.m)
        c1, c2 = self.cv1.conv.in_channels, self.cv2.conv.out_channels
        form = (c1, self.c, c2)
        f64 = self._form64() or self._form32up()  # the forms that read a virtual Upsample + Concat themselves
        fn = L.lib().upa_c2f64_fused if self._form64() else L.lib().upa_c2f32_up_fused
        if not (self.fuse_block and x.dtype == torch.bfloat16 and not self.training
                and ((form == (32, 16, 32) and nb == 1 and self.m[0].add) or (form == (64, 32, 64) and nb in (1, 2)) or f64)):
            return None
        convs = [self.cv1, self.cv2] + [cv for m in self.m for cv in (m.cv1, m.cv2)]
        if not (all(isinstance(cv.act, nn.SiLU) and cv.conv.groups == 1 and cv.conv.stride == (1, 1) and hasattr(cv, "bn")
                    for cv in convs)
                and all(cv.conv.kernel_size == (3, 3) and cv.conv.padding == (1, 1) for cv in convs[2:])
                and all(m.add == self.m[0].add for m in self.m)):
            return None
        n, _, h, w = x.shape
        y = out if out is not None else R.alloc_nhwc(n, c2, h, w, x.dtype, x.device, key=(id(self), "y"))
        pk = [cv._packed(cv.conv, cv.bn, x.device, x.dtype, False) for cv in convs]
        wm = (C.c_void_p * (2 * nb))(*[q.w.data_ptr() for q in pk[2:]])
        bm = (C.c_void_p * (2 * nb))(*[q.bias.data_ptr() for q in pk[2:]])
        vx, vy = R.view_of(x), R.view_of(y)
        if f64:
            upp, upc, upld = None, 0, 0
            if up is not None and not up.done:
                vu = R.view_of(up.src)
                if vu.dtype == vx.dtype and vu.c == up.channels and up.channels % 64 == 0 and (vu.n, 2 * vu.h, 2 * vu.w) == (vx.n, vx.h, vx.w):
                    upp, upc, upld = vu.ptr, vu.c, vu.ld
                else:
                    up.materialize()
            rc = fn(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, upp, upc, upld, nb, int(self.m[0].add),
                    pk[0].w.data_ptr(), pk[0].bias.data_ptr(), C.cast(wm, C.c_void_p), C.cast(bm, C.c_void_p),
                    pk[1].w.data_ptr(), pk[1].bias.data_ptr(), vy.ptr, vy.c, vy.ld, L.ACT_SILU, vx.dtype,
                    R.opts_ptr(), L.current_stream(x.device))
            if rc == 0:
                return y
            if rc != L.UPA_EUNSUPPORTED:
                L.check(rc, "c2f64_fused / c2f32_up_fused")
            return None  # nothing was launched: the separate convolutions follow, cv1 still reading `up` virtually
        rc = L.lib().upa_c2f_fused(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, self.c, nb, int(self.m[0].add), pk[0].w.data_ptr(),
                                   pk[0].bias.data_ptr(), C.cast(wm, C.c_void_p), C.cast(bm, C.c_void_p), pk[1].w.data_ptr(),
                                   pk[1].bias.data_ptr(), vy.ptr, vy.c, vy.ld, L.ACT_SILU, vx.dtype, R.opts_ptr(),
                                   L.current_stream(x.device))
        if rc == 0:
            return y
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "c2f_fused")
        return None

    def forward_down(self, x, down, out=None):
        """This block AND the stride-2 `down` = Conv(32, 64, 3, 2) row behind it as one launch (`upa_c2f16_down_fused`: yolov8n rows 2-3,
        yolov8.yaml:18-19); returns down(self(x)) or None when the pair is outside the form (nothing was launched then)."""
        import ctypes as C
        if not (self.fuse_block and torch.is_tensor(x) and x.dtype == torch.bfloat16 and not self.training and len(self.m) == 1
                and self.m[0].add and (self.cv1.conv.in_channels, self.c, self.cv2.conv.out_channels) == (32, 16, 32)):
            return None
        convs = [self.cv1, self.cv2, self.m[0].cv1, self.m[0].cv2, down]
        dc = down.conv
        if not (all(isinstance(cv.act, nn.SiLU) and cv.conv.groups == 1 and cv.conv.dilation == (1, 1) and hasattr(cv, "bn") for cv in convs)
                and all(cv.conv.stride == (1, 1) for cv in convs[:4])
                and all(cv.conv.kernel_size == (3, 3) and cv.conv.padding == (1, 1) for cv in convs[2:])
                and all(cv.conv.kernel_size == (1, 1) for cv in convs[:2])
                and (dc.in_channels, dc.out_channels, dc.stride) == (32, 64, (2, 2)) and not down.training):
            return None
        x = R.to_nhwc(x, x.dtype)
        n, _, h, w = x.shape
        oh, ow = (h + 1) // 2, (w + 1) // 2
        y = out if out is not None else R.alloc_nhwc(n, 64, oh, ow, x.dtype, x.device, key=(id(down), "y"))
        pk = [cv._packed(cv.conv, cv.bn, x.device, x.dtype, False) for cv in convs]
        wm = (C.c_void_p * 2)(pk[2].w.data_ptr(), pk[3].w.data_ptr())
        bm = (C.c_void_p * 2)(pk[2].bias.data_ptr(), pk[3].bias.data_ptr())
        vx, vy = R.view_of(x), R.view_of(y)
        rc = L.lib().upa_c2f16_down_fused(vx.ptr, vx.n, vx.h, vx.w, vx.ld, pk[0].w.data_ptr(), pk[0].bias.data_ptr(), C.cast(wm, C.c_void_p),
                                          C.cast(bm, C.c_void_p), pk[1].w.data_ptr(), pk[1].bias.data_ptr(), pk[4].w.data_ptr(),
                                          pk[4].bias.data_ptr(), vy.ptr, vy.ld, vx.dtype, R.opts_ptr(), L.current_stream(x.device))
        if rc == 0:
            return y
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "c2f16_down_fused")
        return None

    def forward(self, x, out=None, up=None):
        """`up`: a conv.VirtualUpsample for the leading channels of x (see BaseModel._predict_once) - consumed by cv1."""
        x = R.to_nhwc(x, x.dtype)
        if up is not None and self.fuse_block and self.cv1.conv.in_channels in (32, 64) and not (self._form64() or self._form32up()):
            up.materialize()  # the narrow whole-block kernels read x themselves
            up = None
        y = self._fused(x, out, up)
        if y is not None:
            return y
        if up is not None and up.done:
            up = None
        n, _, h, w = x.shape
        c, nb = self.c, len(self.m)
        cat = R.alloc_nhwc(n, (2 + nb) * c, h, w, x.dtype, x.device, key=(id(self), "cat"))
        self.cv1(x, out=cat[:, : 2 * c], up=up)
        y = self._pair_cv2(cat, out)
        if y is not None:
            return y
        for i, m in enumerate(self.m):
            m(cat[:, (1 + i) * c: (2 + i) * c], out=cat[:, (2 + i) * c: (3 + i) * c])
        return self.cv2(cat, out=out)

    fuse_pair_cv2 = True  # bf16, one 32-channel Bottleneck, 64 outputs: Bottleneck + cv2 as one launch (upa_bottleneck_pair_cv2)

    def invalidate_packed(self):
        """Drop the split cv2 weights of `_pair_cv2` (utils/weights.py and `train()` call this on every module that has it)."""
        self.__dict__.pop("_pc_cache", None)

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def _pair_cv2(self, cat, out):
        """[Bottleneck + cv2] in one launch after cv1 has filled cat[:, :2c]; None when the block is outside that form."""
        m = self.m[0] if len(self.m) == 1 else None
        if not (self.fuse_pair_cv2 and m is not None and self.c == 32 and self.cv2.conv.out_channels == 64
                and cat.dtype == torch.bfloat16 and not self.training and m._pair_ok(cat[:, 32:64])
                and isinstance(self.cv2.act, nn.SiLU) and self.cv2.conv.kernel_size == (1, 1) and hasattr(self.cv2, "bn")):
            return None
        n, _, h, w = cat.shape
        dev = cat.device
        y = out if out is not None else R.alloc_nhwc(n, 64, h, w, cat.dtype, dev, key=(id(self), "y"))
        p1 = m.cv1._packed(m.cv1.conv, m.cv1.bn, dev, cat.dtype, False)
        p2 = m.cv2._packed(m.cv2.conv, m.cv2.bn, dev, cat.dtype, False)
        cache = self.__dict__.setdefault("_pc_cache", {})
        cv = self.cv2
        ver = version_key(cv.conv.weight, cv.conv.bias, cv.bn.weight, cv.bn.bias, cv.bn.running_mean, cv.bn.running_var) + (cv.bn.eps,)
        hit = cache.get(str(dev))
        if hit is None or hit[0] != ver:
            wf, bf = fold_bn(cv.conv, cv.bn)  # (64, 96, 1, 1), (64,)
            std = PackedConv(wf[:, :64].contiguous(), bf, 1, dev, cat.dtype, False)
            wb = wf[:, 64:96].reshape(64, 32).contiguous().float()
            host = torch.empty(L.lib().upa_tail_packed_weight_bytes(64, 32), dtype=torch.uint8)
            L.check(L.lib().upa_pack_tail_weight(wb.data_ptr(), 64, 32, host.data_ptr()), "pack_tail_weight")
            hit = (ver, std, host.to(dev))
            cache[str(dev)] = hit
        _, std, wb_d = hit
        vx, v0, vy = R.view_of(cat[:, 32:64]), R.view_of(cat[:, :32]), R.view_of(y)
        rc = L.lib().upa_bottleneck_pair_cv2(vx.ptr, v0.ptr, vx.n, vx.h, vx.w, vx.ld, p1.w.data_ptr(), p1.bias.data_ptr(),
                                             p2.w.data_ptr(), p2.bias.data_ptr(), int(m.add), std.w.data_ptr(), wb_d.data_ptr(),
                                             std.bias.data_ptr(), vy.ptr, vy.ld, L.ACT_SILU, vx.dtype, R.opts_ptr(),
                                             L.current_stream(dev))
        if rc == 0:
            return y
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "bottleneck_pair_cv2")
        return None


def _stacked_cv12(self, x, up):
    """cv1(x) and cv2(x) of a C3 / BoT3 (block.py:509-532, 6095-6109) as ONE launch: both are 1x1 Conv + SiLU on the same input, so
    their folded filters are stacked ([cv2 | cv1] along the output channels) and the launch writes channels [c_, 3c_) of one
    (n, 3c_, h, w) buffer laid out [m(cv1(x)) | cv2(x) | cv1(x)]: the concat that cv3 reads is its first 2c_ channels, the Bottleneck
    chain reads the last c_ and its last member writes the first c_.  x is read once instead of twice.  Returns (buffer, c_), or None
    outside the form (then the two convs run separately)."""
    a, b = self.cv1, self.cv2
    c_ = a.conv.out_channels
    ok = (self.stack_cv12 and not self.training and x.dtype in (torch.float32, torch.bfloat16) and b.conv.out_channels == c_
          and a.conv.kernel_size == (1, 1) and b.conv.kernel_size == (1, 1) and a.conv.stride == (1, 1) and b.conv.stride == (1, 1)
          and isinstance(a.act, nn.SiLU) and isinstance(b.act, nn.SiLU) and hasattr(a, "bn") and hasattr(b, "bn")
          and c_ % (8 if x.dtype == torch.bfloat16 else 4) == 0)
    if not ok:
        return None
    n, _, h, w = x.shape
    buf = R.alloc_nhwc(n, 3 * c_, h, w, x.dtype, x.device, key=(id(self), "cat3"))
    pk = a._packed_stack([(b.conv, b.bn), (a.conv, a.bn)], x.device, x.dtype)
    hip_conv2d(x, pk, 1, 0, L.ACT_SILU, out=buf[:, c_:], up=up)
    return buf, c_


class C3(nn.Module):
    """CSP bottleneck with 3 convolutions (block.py:509-532): cv3(cat(m(cv1(x)), cv2(x))).
    `up`: a conv.VirtualUpsample for the leading channels of x - both 1x1 convs that read x take it."""

    stack_cv12 = True  # cv1 and cv2 as one launch (`_stacked_cv12`); A/B and test switch

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def forward(self, x, out=None, up=None):
        x = R.to_nhwc(x, x.dtype)
        n, _, h, w = x.shape
        c_ = self.cv1.conv.out_channels
        st = _stacked_cv12(self, x, up) if len(self.m) else None
        if st is not None:
            buf, _ = st
            y = buf[:, 2 * c_:]
            for i, m in enumerate(self.m):
                y = m(y, out=buf[:, :c_] if i == len(self.m) - 1 else None)
            return self.cv3(buf[:, : 2 * c_], out=out)
        cat = R.alloc_nhwc(n, 2 * c_, h, w, x.dtype, x.device, key=(id(self), "cat"))
        self.cv2(x, out=cat[:, c_:], up=up)
        y = self.cv1(x, up=up) if len(self.m) else self.cv1(x, out=cat[:, :c_], up=up)
        for i, m in enumerate(self.m):
            y = m(y, out=cat[:, :c_] if i == len(self.m) - 1 else None)
        return self.cv3(cat, out=out)


class SPPF(nn.Module):
    """Spatial Pyramid Pooling - Fast (block.py:382-406): cv1 -> 3 chained MaxPool2d(5,1,2) -> cat 4 -> cv2.
    The three pools are one kernel (5/9/13 windows of the cv1 output) writing straight into the concat buffer."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    fuse_front = True  # cv1 and the three pools as one launch (A/B switch; upa_opts.no_sppf_front too)

    def _front(self, x, cat, c_) -> bool:
        """`upa_sppf_front`: cat[:, :c_] = cv1(x) and cat[:, c_:] = its three chained pools in ONE launch; False (nothing launched) outside
        the form (bf16 inference, cv1 = Conv(c1, c_, 1, 1) with SiLU, c1 in (128, 256, 512), maps of up to 1024 pixels)."""
        cv = self.cv1
        if not (self.fuse_front and x.dtype == torch.bfloat16 and not self.training and isinstance(cv.act, nn.SiLU) and hasattr(cv, "bn")
                and cv.conv.kernel_size == (1, 1) and cv.conv.stride == (1, 1) and cv.conv.groups == 1):
            return False
        pk = cv._packed(cv.conv, cv.bn, x.device, x.dtype, False)
        vx, vc = R.view_of(x), R.view_of(cat)
        rc = L.lib().upa_sppf_front(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, pk.w.data_ptr(), pk.bias.data_ptr(), vc.ptr, c_, vc.ld, vx.dtype,
                                    R.opts_ptr(), L.current_stream(x.device))
        if rc == 0:
            return True
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "sppf_front")
        return False

    def forward(self, x, out=None):
        if self.m.kernel_size != 5:
            raise L.UpaError("HIP SPPF implements k=5 (every reference YAML on the hot path)")
        x = R.to_nhwc(x, x.dtype)
        n, _, h, w = x.shape
        c_ = self.cv1.conv.out_channels
        cat = R.alloc_nhwc(n, 4 * c_, h, w, x.dtype, x.device, key=(id(self), "cat"))
        if self._front(x, cat, c_):  # cv1 and the three pools as ONE launch (bf16; upa_sppf_front)
            return self.cv2(cat, out=out)
        self.cv1(x, out=cat[:, :c_])
        v = R.view_of(cat[:, :c_])
        ptr = lambda i: R.view_of(cat[:, i * c_: (i + 1) * c_]).ptr  # noqa: E731
        L.check(L.lib().upa_sppf_pool3(v.ptr, v.n, v.h, v.w, v.c, v.ld, ptr(1), ptr(2), ptr(3), v.ld, v.dtype,
                                       L.current_stream(x.device)), "sppf_pool3")
        return self.cv2(cat, out=out)


class MHSA(nn.Module, _HipConvMixin):
    """Multi-head self attention over a feature map (block.py:6020-6062): q,k,v 1x1 convs with bias, energy = q^T k
    (unscaled), softmax over keys, out = v.attn^T, `view` back with (W,H) reinterpretation."""

    def __init__(self, n_dims, width=14, height=14, heads=4, pos_emb=False):
        super().__init__()
        if pos_emb:
            raise L.UpaError("MHSA(pos_emb=True) is not used on the hot path (block.py:6078) and is not implemented")
        self.heads = heads
        self.query = nn.Conv2d(n_dims, n_dims, kernel_size=1)
        self.key = nn.Conv2d(n_dims, n_dims, kernel_size=1)
        self.value = nn.Conv2d(n_dims, n_dims, kernel_size=1)
        self.pos = pos_emb
        self.softmax = nn.Softmax(dim=-1)

    stack_qkv = True

    def forward(self, x, out=None, residual=None):
        x = R.to_nhwc(x, x.dtype)
        n, c, h, w = x.shape
        qkv = R.alloc_nhwc(n, 3 * c, h, w, x.dtype, x.device, key=(id(self), "qkv"))
        if self.stack_qkv:  # query / key / value read the same x: one launch with the three filters stacked
            pk = self._packed_stack([(self.query, None), (self.key, None), (self.value, None)], x.device, x.dtype)
            hip_conv2d(x, pk, 1, 0, L.ACT_NONE, out=qkv)
        else:
            for i, conv in enumerate((self.query, self.key, self.value)):
                pk = self._packed(conv, None, x.device, x.dtype, False)
                hip_conv2d(x, pk, 1, 0, L.ACT_NONE, out=qkv[:, i * c: (i + 1) * c])
        y = out if out is not None else R.alloc_nhwc(n, c, h, w, x.dtype, x.device, key=(id(self), "y"))
        vq, vy = R.view_of(qkv[:, :c]), R.view_of(y)
        rp, rld = (None, 0)
        if residual is not None:
            vr = R.view_of(residual)
            rp, rld = vr.ptr, vr.ld
        esz = x.element_size()
        L.check(L.lib().upa_mhsa(vq.ptr, vq.ptr + c * esz, vq.ptr + 2 * c * esz, vq.ld, n, h * w, self.heads,
                                 c // self.heads, 1.0, rp, rld, vy.ptr, vy.ld, vq.dtype, L.current_stream(x.device)),
                "mhsa")
        return y


class BottleneckTransformer(nn.Module):
    """x + MHSA(cv1(x)) (block.py:6065-6092); `fc1` is a dead parameter kept for the state_dict (:6088)."""

    def __init__(self, c1, c2, stride=1, heads=4, mhsa=True, resolution=None, expansion=1):
        super().__init__()
        if not mhsa or stride != 1 or c1 != expansion * c2:
            raise L.UpaError("BottleneckTransformer: only the BoT3 configuration (mhsa, stride 1, c1 == c2) is built")
        c_ = int(c2 * expansion)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = nn.Sequential(MHSA(c2, width=int(resolution[0]), height=int(resolution[1]), heads=heads))
        self.shortcut = c1 == c2
        self.fc1 = nn.Linear(c2, c2)

    def forward(self, x, out=None):
        x = R.to_nhwc(x, x.dtype)
        return self.cv2[0](self.cv1(x), out=out, residual=x if self.shortcut else None)


class BoT3(nn.Module):
    """CSP bottleneck whose inner blocks are BottleneckTransformers (block.py:6095-6109)."""

    def __init__(self, c1, c2, n=1, e=0.5, e2=1, w=20, h=20):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(
            *[BottleneckTransformer(c_, c_, stride=1, heads=4, mhsa=True, resolution=(w, h), expansion=e2)
              for _ in range(n)])

    stack_cv12 = True

    def forward(self, x, out=None):
        x = R.to_nhwc(x, x.dtype)
        n, _, h, w = x.shape
        c_ = self.cv1.conv.out_channels
        st = _stacked_cv12(self, x, None) if len(self.m) else None
        if st is not None:
            buf, _ = st
            y = buf[:, 2 * c_:]
            for i, m in enumerate(self.m):
                y = m(y, out=buf[:, :c_] if i == len(self.m) - 1 else None)
            return self.cv3(buf[:, : 2 * c_], out=out)
        cat = R.alloc_nhwc(n, 2 * c_, h, w, x.dtype, x.device, key=(id(self), "cat"))
        self.cv2(x, out=cat[:, c_:])
        y = self.cv1(x)
        for i, m in enumerate(self.m):
            y = m(y, out=cat[:, :c_] if i == len(self.m) - 1 else None)
        return self.cv3(cat, out=out)
