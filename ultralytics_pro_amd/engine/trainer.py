"""Training step of the detection model on the HIP path (BASELINE config 3).

Mirrors the reference's step - forward in train mode, `v8DetectionLoss`, `loss.sum() * world_size`, backward, gradient
all-reduce, `optimizer_step` = clip_grad_norm_(10) + SGD(nesterov) + EMA (ultralytics/engine/trainer.py:416-432,
:674-682, :891-950; utils/torch_utils.py:606-650; utils/loss.py:415-528) - without torch autograd: every layer has an
explicit forward that keeps what its explicit backward needs, all of it upa_* kernels (include/upa.h, "training step").

  Conv (conv -> BN(batch statistics) -> SiLU):  z = conv(x, W);  (mean, var) = stats(z);  y = act(bn(z)) (+ residual)
      backward: dz, dgamma, dbeta = bn_act_bwd(z, dy);  dW += wgrad(x, dz);  dx (+)= conv(dz, W^T flipped)
  C2f / Bottleneck / SPPF / Upsample / Concat / Detect: compositions of the above + pool / upsample backward kernels.

torch is used for memory, streams, views and torch.distributed (the gradient all-reduce over RCCL).  Parameters live
in one flat f32 buffer (grouped: biases | decayed weights | norm weights, trainer.py:917-926) so the optimizer is three
fused launches; `model.parameters()` are views into it, the state_dict stays the reference's.
Only the yolov8 module set is differentiable here (Conv, C2f, Bottleneck, SPPF, nn.Upsample, Concat, Detect); other
modules raise.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

import torch
import torch.nn as nn

from .. import _lib as L
from ..nn.modules import block as B
from ..nn.modules import conv as CV
from ..nn.modules import head as H
from ..nn.modules import resample as RS
from . import runtime as R

HYP = dict(lr=0.01, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema_decay=0.9999, ema_tau=2000.0,
           box=7.5, cls=0.5, dfl=1.5)
# warm-up / accumulation schedule of the reference's training loop (engine/trainer.py:337-338, 392-413, 245; cfg/default.yaml)
SCHED = dict(nbs=64, warmup_epochs=3.0, warmup_momentum=0.8, warmup_bias_lr=0.1, lrf=0.01, epochs=100)
MAX_GT = 64        # initial gt-row capacity per image; grows with the data (the loss kernels take it at run time)
MAX_GT_CAP = 1024  # upa_detection_loss: LDS arrays of the assignment-resolve kernel


def _s(dev):
    return L.current_stream(dev)


class _Ctx:
    """Per-trainer scratch shared by all layers."""

    def __init__(self, device, dtype, wgrad_streams: int = 1):
        self.device, self.dtype = device, dtype
        self.code = L.dtype_code(dtype)
        self.es = 2 if dtype == torch.bfloat16 else 4
        self.E = 16 // self.es
        # channel-reduction scratch (per-block partial sums + combined sums), sized for c <= 1024
        self._ws = [torch.zeros(L.lib().upa_channel_reduce_workspace_bytes(1024) // 8, dtype=torch.float64, device=device) for _ in range(2)]
        # the Detect head's 40 x 40 / 20 x 20 levels run beside the 80 x 80 level on a second stream (`DetectT`): its own reduction
        # workspace (the reductions of one workspace must be stream-ordered)
        self.level_stream = torch.cuda.Stream(device=device)
        self._ws_sel = 0
        self.wgrad_ws = torch.empty(0, dtype=torch.uint8, device=device)  # weight-gradient partial sums (largest layer)
        # weight gradients run on a side stream (a parallel branch of the captured graph): a layer's dW only needs its
        # input and dz, so it overlaps the data-gradient / BN-backward chain of the layers in front of it
        # (wgrad_streams = 0: weight gradients on the main stream, no overlap - the A/B switch of that measurement)
        self.wgrad_streams = [torch.cuda.Stream(device=device) for _ in range(max(0, int(wgrad_streams)))]
        self.wgrad_wss = [self.wgrad_ws for _ in self.wgrad_streams]  # one partial-sum workspace per stream
        self.wgrad_rr = 0
        self.wgrad_pending = False


def _ctx_ws(self):
    return self._ws[self._ws_sel]


def _ctx_wgrad_ws_here(self):
    """The weight-gradient partial-sum workspace for a weight gradient launched on the CURRENT stream (`wgrad_streams = 0`): the main
    stream's, or - while `DetectT._fork_levels` runs the small Detect levels on the level stream - a second buffer of the same size:
    two streams must never write and reduce split-K partials in one buffer at the same time."""
    if self._ws_sel == 0:
        return self.wgrad_ws
    lv = self.__dict__.get("_wgrad_ws_level")
    if lv is None or lv.numel() < self.wgrad_ws.numel():
        lv = self.__dict__["_wgrad_ws_level"] = torch.empty(self.wgrad_ws.numel(), dtype=torch.uint8, device=self.device)
    return lv


_Ctx.ws = property(_ctx_ws)
_Ctx.wgrad_ws_here = property(_ctx_wgrad_ws_here)


def _new(n, c, h, w, dtype, dev, key):
    return R.alloc_nhwc(n, c, h, w, dtype, dev, key)


class ConvT:
    """Train-mode forward/backward of one conv (+ BatchNorm2d + activation, or + bias for the plain head outputs)."""

    def __init__(self, ctx: _Ctx, conv: nn.Conv2d, bn, act_code: int, name: str):
        if conv.groups != 1 or conv.dilation != (1, 1) or conv.kernel_size[0] != conv.kernel_size[1]:
            raise L.UpaError(f"training supports square kernels, groups=1, dilation=1 only ({name})")
        self.ctx, self.conv, self.bn, self.act, self.name = ctx, conv, bn, act_code, name
        self.k, self.s, self.p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if self.k not in (1, 3) or self.s not in (1, 2):
            raise L.UpaError(f"training supports k in (1,3), stride in (1,2) ({name}: k={self.k} s={self.s})")
        self.cout, self.cin = conv.out_channels, conv.in_channels
        dev, code = ctx.device, ctx.code
        nb = L.lib().upa_conv_packed_weight_bytes(self.cout, self.cin, self.k, code)
        self.wp = torch.empty(nb, dtype=torch.uint8, device=dev)          # forward layout
        nbt = L.lib().upa_conv_packed_weight_bytes(self.cin, self.cout, self.k, code)
        self.wpt = torch.empty(nbt, dtype=torch.uint8, device=dev)        # transposed + flipped (data gradient)
        if bn is not None:
            self.mean = torch.empty(self.cout, dtype=torch.float32, device=dev)
            self.var = torch.empty(self.cout, dtype=torch.float32, device=dev)
        self.phase = None
        if self.s == 2 and self.k == 3 and self.p == 1:
            # data gradient by output parity: four 2x2 kernels, stacked along the output channels into ONE cout -> 4 * cin conv
            # (V is [phase][cin][cout][2][2] = a (4 cin, cout, 2, 2) weight): dz is read once and the four phases are one launch
            nph = L.lib().upa_conv_packed_weight_bytes(4 * self.cin, self.cout, 2, code)
            self.phase_v = torch.empty(4 * self.cin * self.cout * 4, dtype=torch.float32, device=dev)
            self.phase = torch.empty(nph, dtype=torch.uint8, device=dev)
        self.x = self.z = None
        nws = L.lib().upa_conv2d_wgrad_workspace_bytes(self.cin, self.cout, self.k)
        if nws > ctx.wgrad_ws.numel():
            ctx.wgrad_ws = torch.empty(nws, dtype=torch.uint8, device=dev)  # shared by all layers (stream ordered)
            ctx.wgrad_wss = [ctx.wgrad_ws if j == 0 else torch.empty(nws, dtype=torch.uint8, device=dev)
                             for j in range(len(ctx.wgrad_streams))]

    def pack(self):
        lib, c = L.lib(), self.ctx
        w = self.conv.weight
        L.check(lib.upa_pack_conv_weight_dev(w.data_ptr(), self.cout, self.cin, self.k, c.code, 0, self.wp.data_ptr(),
                                             _s(c.device)), "pack_dev")
        if self.phase is None:
            L.check(lib.upa_pack_conv_weight_dev(w.data_ptr(), self.cout, self.cin, self.k, c.code, 1, self.wpt.data_ptr(),
                                                 _s(c.device)), "pack_dev_t")
        else:
            L.check(lib.upa_dgrad_s2_phase_weights(w.data_ptr(), self.cout, self.cin, self.phase_v.data_ptr(), _s(c.device)),
                    "phase_weights")
            L.check(lib.upa_pack_conv_weight_dev(self.phase_v.data_ptr(), 4 * self.cin, self.cout, 2, c.code, 0,
                                                 self.phase.data_ptr(), _s(c.device)), "pack_phase")

    def pack_descs(self):
        """(w_ptr, out_ptr, cout, cin, k, dtype, transpose_flip) of every repack `pack()` launches, for the batched form
        (upa_pack_conv_weights_batched); a stride-2 conv's phase kernels are read from `phase_v`, which
        `pack_phase_weights()` must have refreshed first."""
        c, w = self.ctx, self.conv.weight
        d = [(w.data_ptr(), self.wp.data_ptr(), self.cout, self.cin, self.k, c.code, 0)]
        if self.phase is None:
            d.append((w.data_ptr(), self.wpt.data_ptr(), self.cout, self.cin, self.k, c.code, 1))
        else:
            d.append((self.phase_v.data_ptr(), self.phase.data_ptr(), 4 * self.cin, self.cout, 2, c.code, 0))
        return d

    def pack_phase_weights(self):
        if self.phase is not None:
            L.check(L.lib().upa_dgrad_s2_phase_weights(self.conv.weight.data_ptr(), self.cout, self.cin,
                                                       self.phase_v.data_ptr(), _s(self.ctx.device)), "phase_weights")

    def _conv(self, x, wp, cout, k, s, p, out, bias=None, residual=None):
        vx, vy = R.view_of(x), R.view_of(out)
        rp, rld = (None, 0)
        if residual is not None:
            vr = R.view_of(residual)
            rp, rld = vr.ptr, vr.ld
        L.check(L.lib().upa_conv2d_bias_act(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, wp.data_ptr(),
                                            None if bias is None else bias.data_ptr(), vy.ptr, cout, vy.ld, rp, rld, k, s, p,
                                            L.ACT_NONE, vx.dtype, R.opts_ptr(), _s(x.device)), f"conv2d[{self.name}]")

    @staticmethod
    def _momentum(bn) -> float:
        if bn.momentum is None:
            raise L.UpaError("BatchNorm2d(momentum=None) (cumulative moving average) is not supported by the training step: the running "
                             "statistics kernel takes a fixed momentum (nn.BatchNorm2d default 0.1; the reference's YAMLs set 0.03)")
        return float(bn.momentum)

    def forward(self, x, out=None, residual=None):
        c, lib = self.ctx, L.lib()
        n, _, h, w = x.shape
        oh, ow = (h + 2 * self.p - self.k) // self.s + 1, (w + 2 * self.p - self.k) // self.s + 1
        self.x = x
        if self.bn is None:  # plain nn.Conv2d with bias: y = conv(x) + b
            y = out if out is not None else _new(n, self.cout, oh, ow, c.dtype, c.device, (id(self), "y"))
            self._conv(x, self.wp, self.cout, self.k, self.s, self.p, y, bias=self.conv.bias)
            return y
        z = _new(n, self.cout, oh, ow, c.dtype, c.device, (id(self), "z"))
        vx, vz = R.view_of(x), R.view_of(z)
        npix = vz.n * vz.h * vz.w
        bn = self.bn
        # conv + batch statistics + normalisation / activation in ONE call: on the kernels with a statistics epilogue the sums come from
        # the convolution's own workgroups (no pass over z), elsewhere the library runs conv -> reduce -> combine itself
        y = out if out is not None else _new(n, self.cout, oh, ow, c.dtype, c.device, (id(self), "y"))
        vy = R.view_of(y)
        rp, rld = (None, 0)
        if residual is not None:
            vr = R.view_of(residual)
            rp, rld = vr.ptr, vr.ld
        L.check(lib.upa_conv2d_bn_act_fwd(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, self.wp.data_ptr(), vz.ptr, self.cout, vz.ld,
                                          self.k, self.s, self.p, self._momentum(bn), self.mean.data_ptr(), self.var.data_ptr(),
                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.weight.data_ptr(),
                                          bn.bias.data_ptr(), float(bn.eps), self.act, vy.ptr, vy.ld, rp, rld, c.ws.data_ptr(),
                                          vx.dtype, R.opts_ptr(), _s(c.device)), f"conv2d_bn_act_fwd[{self.name}]")
        self.z = z
        return y

    def backward(self, dy, dx=None, accumulate=False):
        """dy: gradient of this layer's output (NHWC view). dx: view receiving / accumulating the input gradient."""
        c, lib = self.ctx, L.lib()
        vdy = R.view_of(dy)
        npix = vdy.n * vdy.h * vdy.w
        st = _s(c.device)
        if self.bn is not None and self.s == 1 and self.phase is None:
            # the common form in ONE library call: BN + activation backward, the weight gradient (on the side stream, behind an event),
            # the data gradient
            vz, vx = R.view_of(self.z), R.view_of(self.x)
            dz = _new(vz.n, vz.c, vz.h, vz.w, c.dtype, c.device, (id(self), "dz"))
            vdz = R.view_of(dz)
            bn = self.bn
            side_p, ws = None, c.wgrad_ws_here
            if c.wgrad_streams:
                j = c.wgrad_rr % len(c.wgrad_streams)
                c.wgrad_rr += 1
                side_p, ws = c.wgrad_streams[j].cuda_stream, c.wgrad_wss[j]
                c.wgrad_pending = True
            vdx = None if dx is None else R.view_of(dx)
            L.check(lib.upa_conv_bn_act_bwd(vx.ptr, vx.n, vx.h, vx.w, self.cin, vx.ld, vz.ptr, vdy.ptr, self.cout, vz.ld, vdy.ld,
                                            self.mean.data_ptr(), self.var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            float(bn.eps), self.act, vdz.ptr, vdz.ld, bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(),
                                            c.ws.data_ptr(), self.conv.weight.grad.data_ptr(), ws.data_ptr(), ws.numel(), side_p,
                                            None if dx is None else self.wpt.data_ptr(), None if dx is None else vdx.ptr,
                                            0 if dx is None else vdx.ld, int(accumulate), self.k, self.p, vz.dtype, R.opts_ptr(), st),
                    f"conv_bn_act_bwd[{self.name}]")
            return
        if self.bn is None:
            dz = dy
            L.check(lib.upa_channel_sum(vdy.ptr, npix, vdy.c, vdy.ld, self.conv.bias.grad.data_ptr(), 1, c.ws.data_ptr(),
                                        vdy.dtype, st), "bias_grad")
        else:
            vz = R.view_of(self.z)
            dz = _new(vz.n, vz.c, vz.h, vz.w, c.dtype, c.device, (id(self), "dz"))
            vdz = R.view_of(dz)
            bn = self.bn
            L.check(lib.upa_bn_act_bwd(vz.ptr, vdy.ptr, npix, vz.c, vz.ld, vdy.ld, self.mean.data_ptr(), self.var.data_ptr(),
                                       bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), self.act, vdz.ptr, vdz.ld,
                                       bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), 1, c.ws.data_ptr(), vz.dtype, st),
                    "bn_act_bwd")
        vx, vdz = R.view_of(self.x), R.view_of(dz)
        if not c.wgrad_streams:
            L.check(lib.upa_conv2d_wgrad(vx.ptr, vx.n, vx.h, vx.w, self.cin, vx.ld, vdz.ptr, self.cout, vdz.ld,
                                         self.conv.weight.grad.data_ptr(), self.k, self.s, self.p, 1, vx.dtype,
                                         c.wgrad_ws_here.data_ptr(), c.wgrad_ws_here.numel(), st), f"wgrad[{self.name}]")
        else:
            j = c.wgrad_rr % len(c.wgrad_streams)
            c.wgrad_rr += 1
            side, ws = c.wgrad_streams[j], c.wgrad_wss[j]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(c.device))  # dz is ready
            side.wait_event(ev)
            L.check(lib.upa_conv2d_wgrad(vx.ptr, vx.n, vx.h, vx.w, self.cin, vx.ld, vdz.ptr, self.cout, vdz.ld,
                                         self.conv.weight.grad.data_ptr(), self.k, self.s, self.p, 1, vx.dtype,
                                         ws.data_ptr(), ws.numel(), side.cuda_stream), f"wgrad[{self.name}]")
            c.wgrad_pending = True
        if dx is None:
            return
        if self.phase is not None:
            # stride 2: the four 2x2 stride-1 correlations over dz (one per output parity) as one conv with 4 * cin output
            # channels + one interleave pass over its four channel blocks
            vdx = R.view_of(dx)
            rc = lib.upa_conv2d_dgrad_s2(vdz.ptr, vdz.n, vdz.h, vdz.w, self.cout, vdz.ld, self.phase.data_ptr(), vdx.ptr, vdx.h, vdx.w,
                                         self.cin, vdx.ld, int(accumulate), vdx.dtype, R.opts_ptr(), st)
            if rc != L.UPA_EUNSUPPORTED:  # one launch: the phase conv's epilogue writes dx's interleaved pixels itself
                L.check(rc, "conv2d_dgrad_s2")
                return
            t = _new(vdz.n, 4 * self.cin, vdz.h + 1, vdz.w + 1, c.dtype, c.device, (id(self), "phases"))
            self._conv(dz, self.phase, 4 * self.cin, 2, 1, 1, t)
            ts = [R.view_of(t[:, ph * self.cin:(ph + 1) * self.cin]) for ph in range(4)]  # raw pointers: `t` outlives the launch below
            vdx = R.view_of(dx)
            L.check(lib.upa_interleave2x(ts[0].ptr, ts[1].ptr, ts[2].ptr, ts[3].ptr, vdz.n, vdz.h + 1, vdz.w + 1, self.cin, ts[0].ld,
                                         vdx.ptr, vdx.h, vdx.w, vdx.ld, int(accumulate), vdx.dtype, st), "interleave2x")
            return
        src = dz
        if self.s == 2:  # other stride-2 shapes: zero-insert dz, then a stride-1 correlation with the flipped weights
            up = _new(vx.n, self.cout, vx.h, vx.w, c.dtype, c.device, (id(self), "dz_up"))
            vu = R.view_of(up)
            L.check(lib.upa_dilate2x(vdz.ptr, vdz.n, vdz.h, vdz.w, vdz.c, vdz.ld, vu.ptr, vu.h, vu.w, vu.ld, vdz.dtype, st),
                    "dilate2x")
            src = up
        self._conv(src, self.wpt, self.cin, self.k, 1, self.k - 1 - self.p, dx, residual=dx if accumulate else None)


class _Seq:
    """Helpers shared by the composite layers."""

    def __init__(self, ctx):
        self.ctx = ctx


class BottleneckT(_Seq):
    def __init__(self, ctx, m: B.Bottleneck, name):
        super().__init__(ctx)
        self.cv1 = ConvT(ctx, m.cv1.conv, m.cv1.bn, m.cv1._act_code(), name + ".cv1")
        self.cv2 = ConvT(ctx, m.cv2.conv, m.cv2.bn, m.cv2._act_code(), name + ".cv2")
        self.add = m.add

    def convs(self):
        return [self.cv1, self.cv2]

    def forward(self, x, out=None):
        t = self.cv1.forward(x)
        return self.cv2.forward(t, out=out, residual=x if self.add else None)

    def backward(self, dy, dx, accumulate):
        c = self.ctx
        vt = R.view_of(self.cv2.x)
        dt = _new(vt.n, vt.c, vt.h, vt.w, c.dtype, c.device, (id(self), "dt"))
        self.cv2.backward(dy, dt, False)
        self.cv1.backward(dt, dx, accumulate)
        if self.add:  # y = x + f(x): the shortcut passes dy straight through
            add_into(c, dy, dx)


def add_into(ctx, src, dst):
    """dst += src (NHWC views)."""
    vs, vd = R.view_of(src), R.view_of(dst)
    L.check(L.lib().upa_add_view(vd.ptr, vd.ld, vs.ptr, vs.ld, vd.ptr, vd.ld, vd.n, vd.h, vd.w, vd.c, vd.dtype, _s(ctx.device)),
            "add_view")


def copy_into(ctx, src, dst):
    vs, vd = R.view_of(src), R.view_of(dst)
    L.check(L.lib().upa_copy_view(vs.ptr, vs.n, vs.h, vs.w, vs.c, vs.ld, vd.ptr, vd.ld, vs.dtype, _s(ctx.device)), "copy_view")


class C2fT(_Seq):
    def __init__(self, ctx, m: B.C2f, name):
        super().__init__(ctx)
        self.c = m.c
        self.cv1 = ConvT(ctx, m.cv1.conv, m.cv1.bn, m.cv1._act_code(), name + ".cv1")
        self.cv2 = ConvT(ctx, m.cv2.conv, m.cv2.bn, m.cv2._act_code(), name + ".cv2")
        self.m = [BottleneckT(ctx, b, f"{name}.m.{i}") for i, b in enumerate(m.m)]

    def convs(self):
        return [self.cv1, self.cv2] + [cv for b in self.m for cv in b.convs()]

    def forward(self, x, out=None):
        c, k = self.ctx, self.c
        n, _, h, w = x.shape
        cat = _new(n, (2 + len(self.m)) * k, h, w, c.dtype, c.device, (id(self), "cat"))
        self.cv1.forward(x, out=cat[:, :2 * k])
        for i, b in enumerate(self.m):
            b.forward(cat[:, (1 + i) * k:(2 + i) * k], out=cat[:, (2 + i) * k:(3 + i) * k])
        self.cat = cat
        return self.cv2.forward(cat, out=out)

    def backward(self, dy, dx, accumulate):
        c, k = self.ctx, self.c
        n, ct, h, w = self.cat.shape
        g = _new(n, ct, h, w, c.dtype, c.device, (id(self), "gcat"))
        self.cv2.backward(dy, g, False)
        for i in reversed(range(len(self.m))):
            self.m[i].backward(g[:, (2 + i) * k:(3 + i) * k], g[:, (1 + i) * k:(2 + i) * k], True)
        self.cv1.backward(g[:, :2 * k], dx, accumulate)


class SPPFT(_Seq):
    def __init__(self, ctx, m: B.SPPF, name):
        super().__init__(ctx)
        self.cv1 = ConvT(ctx, m.cv1.conv, m.cv1.bn, m.cv1._act_code(), name + ".cv1")
        self.cv2 = ConvT(ctx, m.cv2.conv, m.cv2.bn, m.cv2._act_code(), name + ".cv2")
        self.k = m.m.kernel_size if isinstance(m.m.kernel_size, int) else m.m.kernel_size[0]

    def convs(self):
        return [self.cv1, self.cv2]

    def forward(self, x, out=None):
        c = self.ctx
        n, _, h, w = x.shape
        cm = self.cv1.cout
        cat = _new(n, 4 * cm, h, w, c.dtype, c.device, (id(self), "cat"))
        self.cv1.forward(x, out=cat[:, :cm])
        if self.k == 5:  # the three chained MaxPool2d(5, 1, 2) (block.py:402-406) as one launch (5 / 9 / 13 windows)
            v = [R.view_of(cat[:, i * cm:(i + 1) * cm]) for i in range(4)]
            L.check(L.lib().upa_sppf_pool3(v[0].ptr, v[0].n, v[0].h, v[0].w, v[0].c, v[0].ld, v[1].ptr, v[2].ptr, v[3].ptr, v[0].ld,
                                           v[0].dtype, _s(c.device)), "sppf_pool3")
        else:
            for i in range(3):
                vs, vd = R.view_of(cat[:, i * cm:(i + 1) * cm]), R.view_of(cat[:, (i + 1) * cm:(i + 2) * cm])
                L.check(L.lib().upa_maxpool2d(vs.ptr, vs.n, vs.h, vs.w, vs.c, vs.ld, vd.ptr, vd.h, vd.w, vd.ld, self.k, 1,
                                              self.k // 2, 0, vs.dtype, _s(c.device)), "maxpool2d")
        self.cat = cat
        return self.cv2.forward(cat, out=out)

    def backward(self, dy, dx, accumulate):
        c = self.ctx
        n, ct, h, w = self.cat.shape
        cm = ct // 4
        g = _new(n, ct, h, w, c.dtype, c.device, (id(self), "gcat"))
        self.cv2.backward(dy, g, False)
        for i in (2, 1, 0):
            vx, vdy, vdx = R.view_of(self.cat[:, i * cm:(i + 1) * cm]), R.view_of(g[:, (i + 1) * cm:(i + 2) * cm]), \
                R.view_of(g[:, i * cm:(i + 1) * cm])
            nws = L.lib().upa_maxpool2d_bwd_workspace_bytes(vx.n, vx.h, vx.w, vx.c, self.k, 1, self.k // 2)
            ws = R.alloc_plain((nws,), torch.uint8, c.device, (id(self), "argmax"))
            L.check(L.lib().upa_maxpool2d_bwd(vx.ptr, vdy.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, vdy.ld, self.k, 1, self.k // 2,
                                              vdx.ptr, vdx.ld, 1, vx.dtype, ws.data_ptr(), nws, _s(c.device)), "maxpool2d_bwd")
        self.cv1.backward(g[:, :cm], dx, accumulate)


class _Node:
    """One row of the model YAML inside the training graph."""

    def __init__(self, i, f, kind, op):
        self.i, self.f, self.kind, self.op = i, f, kind, op
        self.y = None       # forward output
        self.g = None       # gradient buffer of the output
        self.g_ready = False


class GradScaler:
    """`torch.amp.GradScaler("cuda", enabled=amp)` of the reference (engine/trainer.py:301-302; used at :429 scale(loss).backward(),
    :676 unscale_, :678 step, :679 update, :593 / :824-825 state_dict) with its state ON THE DEVICE: {scale, growth tracker, found_inf}.
    The loss kernels multiply their gradients by the scale (upa_detection_loss_scaled), the optimizer kernel divides it out again, clips
    on the unscaled norm and skips an overflowing step (upa_sgd_nesterov_ema_scaled), `upa_grad_scaler_update` adjusts the scale - the
    host never reads it, so a scaled step stays one hipGraph.  Defaults = torch's (init 2**16, growth 2, backoff 0.5, interval 2000).
    bf16 has f32's exponent range and needs no loss scaling; the scaler exists because the reference's AMP loop has one
    (`DetectionTrainer(..., amp_scaler=True)`)."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def ptr(self):
        return self.state.data_ptr() if self.enabled else None

    def update(self, sumsq, stream):
        if self.enabled:
            L.check(L.lib().upa_grad_scaler_update(self.state.data_ptr(), sumsq.data_ptr(), self.growth_factor, self.backoff_factor,
                                                   self.growth_interval, stream), "grad_scaler_update")

    def get_scale(self) -> float:  # (host synchronisation: logging / tests only)
        return float(self.state[0].item()) if self.enabled else 1.0

    def carried_scale(self, after_update: bool) -> float:
        """The scale the gradients in the flat buffer were multiplied by: the current one before `update()` ran for them, the one
        `update()` saved (state[3]) after - on a step where the scale grew or backed off the two differ by that factor."""
        if not self.enabled:
            return 1.0
        return float(self.state[3 if after_update else 0].item())

    def found_inf(self) -> bool:
        return bool(self.state[2].item() != 0.0) if self.enabled else False

    def state_dict(self):
        if not self.enabled:
            return {}
        st = self.state.cpu()
        return {"scale": float(st[0]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[1])}

    def load_state_dict(self, sd):
        if not self.enabled or not sd:
            return
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state.copy_(torch.tensor([float(sd["scale"]), float(sd["_growth_tracker"]), 0.0, 0.0]))


class DetectionTrainer:
    """One-process-per-GPU trainer for a DetectionModel (reference: engine/trainer.py BaseTrainer._do_train inner loop)."""

    def __init__(self, model, dtype=torch.bfloat16, hyp=None, device=None, world_size=1, ema=True, wgrad_streams: int = 1,
                 amp_scaler: bool = False):
        self.model = model
        self.hyp = dict(HYP, **(hyp or {}))
        self.device = torch.device(device or "cuda:0")
        self.dtype = dtype
        self.world_size = world_size
        self.ctx = _Ctx(self.device, dtype, wgrad_streams)
        self.scaler = GradScaler(self.device, enabled=amp_scaler)  # trainer.py:301-302
        self.pool = R.BufferPool()
        self.updates = 0
        self.first_step = True
        self._scaler_updated = False
        self._graphs = None
        self._capturing = False
        self._issue_buckets = False
        self._imgsz = None
        self.gt_d = self.ngt_d = None
        self.schedule = None      # set_schedule(): warm-up interpolation + gradient accumulation of the reference's loop
        self.ni = 0               # iteration counter (the reference's `ni = i + nb * epoch`)
        self.last_opt_step = -1   # trainer.py:386
        self.max_gt = MAX_GT  # rows per image of the padded gt tensor (the reference pads to counts.max(), loss.py:445-461)
        model.to(self.device)
        self._flatten_parameters(ema)
        self._build_graph()

    # ---- parameters ---------------------------------------------------------------------------------------------------
    def _flatten_parameters(self, ema):
        """Flat f32 buffers [biases | decayed weights | norm weights] with model.parameters() viewing into them."""
        from ..parallel import flat_parameter_layout
        layout, meta = flat_parameter_layout(self.model)
        total = sum(n for _, n, _ in layout)
        dev = self.device
        self.P = torch.empty(total, dtype=torch.float32, device=dev)
        self.G = torch.zeros(total, dtype=torch.float32, device=dev)
        self.M = torch.zeros(total, dtype=torch.float32, device=dev)
        for _name, off, n, p in meta:
            self.P[off:off + n].copy_(p.detach().reshape(-1))  # plumbing: one-time gather of the initial weights
            p.data = self.P[off:off + n].view(p.shape)
            p.grad = self.G[off:off + n].view(p.shape)
        self.groups = [(start, n, self.hyp["weight_decay"] if wd else 0.0) for start, n, wd in layout]
        self._param_meta = [(name, off, n) for name, off, n, _ in meta]  # flat order: gradient buckets are cut by layer from it
        self.E = self.P.clone() if ema else None  # ModelEMA copy of the parameters
        bufs = [b for b in self.model.buffers() if b.dtype.is_floating_point]
        nb = sum(b.numel() for b in bufs)
        self.RB = torch.empty(max(nb, 1), dtype=torch.float32, device=dev)
        off = 0
        for b in bufs:
            n = b.numel()
            self.RB[off:off + n].copy_(b.detach().reshape(-1))
            b.data = self.RB[off:off + n].view(b.shape)
            off += n
        self.nbuf = nb
        self.ERB = self.RB.clone() if ema else None
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.sumsq_ws = torch.zeros(L.lib().upa_sumsq_workspace_bytes() // 8, dtype=torch.float64, device=dev)
        self.ema_d_dev = torch.zeros(1, dtype=torch.float32, device=dev)

    # ---- graph --------------------------------------------------------------------------------------------------------
    def _build_graph(self):
        ctx = self.ctx
        self.nodes = []
        self.convs = []
        self._pack_table, self._pack_n = None, 0
        for m in self.model.model:
            name = f"model.{m.i}"
            if isinstance(m, H.Detect):
                op = DetectT(ctx, m, name, self)
                kind = "detect"
            elif isinstance(m, B.C2f):
                op, kind = C2fT(ctx, m, name), "c2f"
            elif isinstance(m, B.SPPF):
                op, kind = SPPFT(ctx, m, name), "sppf"
            elif isinstance(m, CV.Conv):
                op, kind = ConvT(ctx, m.conv, m.bn, m._act_code(), name), "conv"
            elif isinstance(m, CV.Concat):
                op, kind = None, "concat"
            elif isinstance(m, RS.Upsample):
                op, kind = None, "upsample"
            else:
                raise L.UpaError(f"{type(m).__name__} (layer {m.i}) has no training path on HIP yet")
            if op is not None:
                self.convs += op.convs() if hasattr(op, "convs") else [op]
            self.nodes.append(_Node(m.i, m.f, kind, op))
        self.detect = self.nodes[-1].op
        self._plan_concats()

    def _plan_concats(self):
        """nn.Concat without copies: a layer whose output feeds exactly one Concat writes it straight into its channel slice of the
        Concat's buffer (every kernel takes a pixel stride), and in backward that slice of the Concat's gradient IS the layer's
        gradient buffer when the Concat is the first consumer the reverse walk meets.  {producer index: (concat index, first
        channel, channels, channels of the concat)}."""
        cout = {}
        for nd in self.nodes[:-1]:
            src = (lambda j: j if j != -1 else nd.i - 1)
            if nd.kind == "conv":
                cout[nd.i] = nd.op.cout
            elif nd.kind in ("c2f", "sppf"):
                cout[nd.i] = nd.op.cv2.cout
            elif nd.kind == "upsample":
                cout[nd.i] = cout[src(nd.f)]
            elif nd.kind == "concat":
                cout[nd.i] = sum(cout[src(j)] for j in nd.f)
        self._cat_slot = {}
        for nd in self.nodes[:-1]:
            if nd.kind != "concat":
                continue
            idx = [j if j != -1 else nd.i - 1 for j in nd.f]
            c0 = 0
            for j in idx:
                if j not in self._cat_slot and idx.count(j) == 1 and self.nodes[j].i == j and \
                        self.nodes[j].kind in ("conv", "c2f", "sppf", "upsample"):
                    self._cat_slot[j] = (nd.i, c0, cout[j], cout[nd.i])
                c0 += cout[j]

    # ---- one step -----------------------------------------------------------------------------------------------------
    def _input_nhwc(self, img):
        """(N,3,H,W) f32 NCHW image batch -> NHWC view padded to one 16-byte channel group (pad channels stay zero)."""
        n, c, h, w = img.shape
        E = self.ctx.E
        zeroed = self.__dict__.setdefault("_img_zeroed", set())
        buf = self.pool.get(("img", n, h, w), (n, h, w, E), self.dtype, self.device)
        if (n, h, w) not in zeroed:
            buf.zero_()  # one-time: the pad channels are never written again
            zeroed.add((n, h, w))
        L.check(L.lib().upa_nchw_to_nhwc(img.data_ptr(), n, c, h, w, buf.data_ptr(), E, self.ctx.code, _s(self.device)),
                "nchw_to_nhwc")
        return buf.permute(0, 3, 1, 2)

    def _pack_weights(self):
        """Repack every conv's master weights into the MFMA fragment layouts (forward, data gradient) - one launch for the
        whole model plus one per stride-2 conv (its four phase kernels).  The descriptor table is built once: the master
        weights are views of the flat parameter buffer and the packed buffers are owned by the layers."""
        for cv in self.convs:
            cv.pack_phase_weights()
        if self._pack_table is None:
            rows = [d for cv in self.convs for d in cv.pack_descs()]
            t = np.zeros(len(rows), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("k", "<i4"),
                                                    ("dtype", "<i4"), ("tf", "<i4"), ("reserved", "<i4")]))
            for i, r in enumerate(rows):
                t[i] = r + (0,)
            self._pack_table = torch.from_numpy(t.view(np.uint8).copy()).to(self.device)
            self._pack_n = len(rows)
        L.check(L.lib().upa_pack_conv_weights_batched(self._pack_table.data_ptr(), self._pack_n, _s(self.device)), "pack_batched")

    def forward_backward(self, img, labels):
        """img: (N,3,H,W) float32 NCHW on the device; labels: the reference's batch dict (batch_idx, cls, bboxes).
        Fills the flat gradient buffer; returns loss_items (3,) on the device."""
        ctx, dev = self.ctx, self.device
        L.require_gpu(img, "train")
        self.model.train()
        self._scaler_updated = False  # the gradients this call produces carry the scaler's CURRENT scale
        if labels is not None:
            # the labels go up BEFORE the forward launches (stream-ordered behind the previous step's loss kernels): packing them on the
            # host overlaps the GPU's work on the previous step instead of standing between this step's forward and its loss
            self._imgsz = (int(img.shape[2]), int(img.shape[3]))
            self._upload_labels(labels, int(img.shape[0]))
            labels = None
        with R.static_buffers(self.pool):
            self._pack_weights()
            x = self._input_nhwc(img.float().contiguous() if img.dtype != torch.float32 else img.contiguous())
            # the stem sees E input channels: 3 real + zero padding (the packed weights are zero there as well)
            ys = []
            for nd in self.nodes:
                xin = x if nd.f == -1 else (ys[nd.f] if isinstance(nd.f, int) else [x if j == -1 else ys[j] for j in nd.f])
                nd.xin = xin
                out = None
                slot = self._cat_slot.get(nd.i)
                if slot is not None:  # this layer's output lives in its slice of the Concat's buffer
                    cat_i, c0, cj, ctot = slot
                    n, _, h, w = xin.shape
                    if nd.kind == "conv":
                        h, w = (h + 2 * nd.op.p - nd.op.k) // nd.op.s + 1, (w + 2 * nd.op.p - nd.op.k) // nd.op.s + 1
                    elif nd.kind == "upsample":
                        h, w = 2 * h, 2 * w
                    out = _new(n, ctot, h, w, ctx.dtype, dev, (id(self.nodes[cat_i]), "y"))[:, c0:c0 + cj]
                if nd.kind in ("conv", "c2f", "sppf"):
                    y = nd.op.forward(xin, out=out)
                elif nd.kind == "upsample":
                    n, c, h, w = xin.shape
                    y = out if out is not None else _new(n, c, 2 * h, 2 * w, ctx.dtype, dev, (id(nd), "y"))
                    vs, vd = R.view_of(xin), R.view_of(y)
                    L.check(L.lib().upa_upsample2x(vs.ptr, vs.n, vs.h, vs.w, vs.c, vs.ld, vd.ptr, vd.ld, vs.dtype, _s(dev)),
                            "upsample2x")
                elif nd.kind == "concat":
                    n, _, h, w = xin[0].shape
                    y = _new(n, sum(int(t.shape[1]) for t in xin), h, w, ctx.dtype, dev, (id(nd), "y"))
                    c0 = 0
                    for j, t in zip(nd.f, xin):
                        j = j if j != -1 else nd.i - 1
                        if self._cat_slot.get(j, (None,))[0] != nd.i:  # not written in place by its producer
                            copy_into(ctx, t, y[:, c0:c0 + t.shape[1]])
                        c0 += t.shape[1]
                else:  # detect
                    y = nd.op.forward(xin)
                nd.y = y
                nd.g = None
                nd.g_ready = False
                x = y
                ys.append(y)
            items = self.detect.loss_backward(labels, img.shape[0])
            self._bucket_hook(self.nodes[-1].i)
            # ---- backward over the layer list in reverse
            for nd in reversed(self.nodes[:-1]):
                if nd.g is None:
                    self._bucket_hook(nd.i)
                    continue  # output unused by the loss
                dy = nd.g
                if nd.kind in ("conv", "c2f", "sppf"):
                    if nd.i == 0:
                        nd.op.backward(dy, None, False)
                    else:
                        src = self.nodes[nd.f] if nd.f != -1 else self.nodes[nd.i - 1]
                        dx, acc = self._grad_of(src)
                        nd.op.backward(dy, dx, acc)
                elif nd.kind == "upsample":
                    src = self.nodes[nd.f] if nd.f != -1 else self.nodes[nd.i - 1]
                    dx, acc = self._grad_of(src)
                    vdy, vdx = R.view_of(dy), R.view_of(dx)
                    L.check(L.lib().upa_upsample2x_bwd(vdy.ptr, vdx.n, vdx.h, vdx.w, vdx.c, vdy.ld, vdx.ptr, vdx.ld, int(acc),
                                                       vdx.dtype, _s(dev)), "upsample2x_bwd")
                elif nd.kind == "concat":
                    c0 = 0
                    for j in nd.f:
                        src = self.nodes[j] if j != -1 else self.nodes[nd.i - 1]
                        cj = int(src.y.shape[1])
                        if src.g is None:   # the Concat is the first consumer met: its slice IS the gradient buffer, later ones accumulate
                            src.g = dy[:, c0:c0 + cj]
                        else:
                            add_into(ctx, dy[:, c0:c0 + cj], src.g)
                        c0 += cj
                self._bucket_hook(nd.i)
            self._join_wgrad()
        return items

    def _join_wgrad(self):
        ctx = self.ctx
        if ctx.wgrad_streams and ctx.wgrad_pending:
            for side in ctx.wgrad_streams:
                ev = torch.cuda.Event()
                ev.record(side)
                torch.cuda.current_stream(self.device).wait_event(ev)
            ctx.wgrad_pending = False

    def _grad_of(self, node):
        """(gradient buffer of node's output, accumulate?) - the first producer writes, later ones accumulate."""
        if node.g is None:
            n, c, h, w = node.y.shape
            node.g = _new(n, c, h, w, self.ctx.dtype, self.device, (id(node), "g"))
            return node.g, False
        return node.g, True

    def grad_sumsq(self):
        off_end = self.groups[-1][0] + self.groups[-1][1]
        L.check(L.lib().upa_sumsq(self.G.data_ptr(), off_end, self.sumsq.data_ptr(), 0, self.sumsq_ws.data_ptr(),
                                  _s(self.device)), "sumsq")
        return self.sumsq

    # ---- gradient exchange -------------------------------------------------------------------------------------------
    def gradient_buckets(self, target_bytes: int = 16 << 20):
        """Layer spans, last layer first, of at least `target_bytes` of f32 gradients each (DDP's bucket_cap_mb idea; yolov8s:
        three spans of its 44.7 MB), as (first layer of the span, [flat ranges]).  Within each optimizer group the parameters
        lie in layer order, so a span of layers is ONE contiguous range per group.  A span is complete - every gradient kernel
        of its layers enqueued - once the backward walk has passed its first layer."""
        from ..parallel import gradient_bucket_table
        return gradient_bucket_table(self._param_meta, self.groups, target_bytes)

    def enable_overlapped_allreduce(self, target_bytes: int = 16 << 20):
        """Issue the gradient all-reduce per bucket DURING backward on a communication stream (eager steps only: a collective
        cannot sit inside the captured backward graph, so `compile()`d multi-GPU steps keep the single all-reduce between their
        two graphs).  Returns the bucket table [(first layer, [ranges])]."""
        from ..parallel import BucketedAllReduce
        table = self.gradient_buckets(target_bytes)
        self._bucket_first = {first: k for k, (first, _) in enumerate(table)}
        self._buckets = BucketedAllReduce(self.G, [r for _, r in table])
        ntot = self.groups[-1][0] + self.groups[-1][1]
        assert self._buckets.covered() == ntot, "gradient buckets must cover the flat buffer exactly once"
        self._exposed = []
        return table

    def _bucket_hook(self, layer_i: int):
        """Called by the backward walk after layer `layer_i` has enqueued its gradient kernels."""
        b = self.__dict__.get("_buckets")
        if b is None or not self._issue_buckets or layer_i not in self._bucket_first:
            return
        evs = []
        for st in [torch.cuda.current_stream(self.device)] + list(self.ctx.wgrad_streams):
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        b.issue(self._bucket_first[layer_i], evs)

    def all_reduce_gradients(self):
        """Batch-DP exchange step: SUM all-reduce of the flat f32 gradient buffer over RCCL - one collective, or
        the buckets `enable_overlapped_allreduce` issued during backward (then only the wait is left here, and the time the
        optimizer had to wait for them is recorded: `allreduce_exposed_ms`).
        The reference multiplies the loss by world_size and lets DDP average the gradients (trainer.py:424-425), i.e.
        every rank ends up with sum_over_ranks d(loss_rank) - exactly the SUM of the unscaled per-rank gradients."""
        if self.world_size > 1:
            b = self.__dict__.get("_buckets")
            if b is not None and b.issued:
                main = torch.cuda.current_stream(self.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                # a span the walk never hooked (parameters outside `model.<N>.*`: layer -1, or a layer without gradient) goes now:
                # backward has finished on the HOST side only, so the communication stream is ordered behind the main stream and
                # every weight-gradient stream first, exactly as `_bucket_hook` does
                late = [k for k in range(len(b.buckets)) if k not in b.issued]
                if late:
                    evs = []
                    for st in [main] + list(self.ctx.wgrad_streams):
                        ev = torch.cuda.Event()
                        ev.record(st)
                        evs.append(ev)
                    for k in late:
                        b.issue(k, evs)
                b.wait()
                e1.record(main)
                self._exposed.append((e0, e1))
                return
            from ..parallel import allreduce_gradients_
            n = self.groups[-1][0] + self.groups[-1][1]
            allreduce_gradients_(self.G[:n])

    def allreduce_exposed_ms(self):
        """Mean time (ms) the main stream spent waiting for the overlapped all-reduce after backward had finished, over the
        steps since the last call (None if nothing was recorded).  Synchronises."""
        ex = self.__dict__.get("_exposed") or []
        if not ex:
            return None
        torch.cuda.synchronize(self.device)
        ms = [a.elapsed_time(b) for a, b in ex]
        self._exposed = []
        return sum(ms) / len(ms)

    def set_schedule(self, batches_per_epoch: int, global_batch: int | None = None, **overrides):
        """Enable the reference's warm-up and gradient accumulation (engine/trainer.py:337-338, 392-413):
        nw = max(round(warmup_epochs * nb), 100) iterations over which the bias learning rate falls from warmup_bias_lr to
        lr0 * lf(epoch), every other learning rate rises from 0, the momentum rises from warmup_momentum and `accumulate`
        grows from 1 to nbs / batch; afterwards the optimizer steps every max(round(nbs / batch), 1) iterations with
        weight_decay * batch * accumulate / nbs.  `global_batch` = the reference's `batch_size` (all ranks together)."""
        sc = dict(SCHED, **overrides)
        sc["nb"] = int(batches_per_epoch)
        sc["global_batch"] = global_batch
        self.schedule = sc
        if self._graphs is not None and len(self._graphs) == 1:
            raise L.UpaError("set_schedule() before compile(): the optimizer has to stay outside the captured graph")
        return self

    def schedule_at(self, ni: int, batch_size: int):
        """(accumulate, [lr biases, lr decayed weights, lr norm weights], momentum, weight_decay) of iteration ni."""
        h, sc = self.hyp, self.schedule
        if sc is None:
            return 1, [h["lr"]] * 3, h["momentum"], h["weight_decay"]
        gb = sc["global_batch"] or batch_size * self.world_size
        acc0 = max(round(sc["nbs"] / gb), 1)
        wd = h["weight_decay"] * gb * acc0 / sc["nbs"]
        lf = max(1 - (ni // sc["nb"]) / sc["epochs"], 0) * (1.0 - sc["lrf"]) + sc["lrf"]
        nw = max(round(sc["warmup_epochs"] * sc["nb"]), 100) if sc["warmup_epochs"] > 0 else -1
        lr = h["lr"] * lf
        if ni <= nw:
            xi = [0, nw]
            acc = max(1, int(np.interp(ni, xi, [1, sc["nbs"] / gb]).round()))
            lrs = [float(np.interp(ni, xi, [sc["warmup_bias_lr"] if j == 0 else 0.0, lr])) for j in range(3)]
            return acc, lrs, float(np.interp(ni, xi, [sc["warmup_momentum"], h["momentum"]])), wd
        return acc0, [lr, lr, lr], h["momentum"], wd

    def optimizer_step(self, lrs=None, momentum=None, weight_decay=None):
        h = self.hyp
        lib, st = L.lib(), _s(self.device)
        self.grad_sumsq()
        self.updates += 1
        d = h["ema_decay"] * (1 - math.exp(-self.updates / h["ema_tau"]))
        dp = None
        if self._capturing:  # a replayed graph reads the decay of the current step from device memory
            dp = self.ema_d_dev.data_ptr()
        lrs = [h["lr"]] * 3 if lrs is None else lrs
        mom = h["momentum"] if momentum is None else momentum
        for gi, (start, n, wd) in enumerate(self.groups):
            if n == 0:
                continue
            if wd and weight_decay is not None:
                wd = weight_decay
            o = 4 * start
            if self.scaler.enabled:  # unscale_ + clip + scaler.step in the same kernel (the momentum buffers start at zero, so the
                # first-step form b = g equals momentum * 0 + g: a skipped first step needs no special case)
                L.check(lib.upa_sgd_nesterov_ema_scaled(self.P.data_ptr() + o, self.G.data_ptr() + o, self.M.data_ptr() + o,
                                                        (self.E.data_ptr() + o) if self.E is not None else None, n,
                                                        self.sumsq.data_ptr(), h["max_norm"], lrs[gi], mom, wd, 0, d, dp, 1,
                                                        self.scaler.ptr(), st), "sgd_scaled")
                continue
            L.check(lib.upa_sgd_nesterov_ema(self.P.data_ptr() + o, self.G.data_ptr() + o, self.M.data_ptr() + o,
                                             (self.E.data_ptr() + o) if self.E is not None else None, n,
                                             self.sumsq.data_ptr(), h["max_norm"], lrs[gi], mom, wd,
                                             int(self.first_step), d, dp, 1, st), "sgd")
        if self.ERB is not None and self.nbuf:
            L.check(lib.upa_ema_update(self.ERB.data_ptr(), self.RB.data_ptr(), self.nbuf, d, dp, st), "ema_buffers")
        self.scaler.update(self.sumsq, st)  # trainer.py:679
        self._scaler_updated = True  # (`grad_norm()` now divides by the scale the update saved, not by the new one)
        self.first_step = False
        # parameters and BN buffers were just rewritten through raw pointers: no `_version` moved, so every packed-weight
        # cache of the inference path (Conv / Detect / C2f) is told explicitly
        CV.bump_weights_generation()

    def step(self, img, labels):
        """One training step. After `compile()` the step is replayed as hipGraphs (one graph on a single GPU; forward +
        backward | all-reduce | optimizer on several)."""
        if self._graphs is not None:
            return self._replay(img, labels)
        # buckets go out during backward only on an iteration that ends in an optimizer step (gradient accumulation adds
        # several backward passes into the flat buffer before it is exchanged once)
        acc = self.schedule_at(self.ni, img.shape[0])[0]
        self._issue_buckets = self.world_size > 1 and self.ni - self.last_opt_step >= acc
        items = self.forward_backward(img, labels)
        self._issue_buckets = False
        self._maybe_optimize(img.shape[0])
        return items

    def _maybe_optimize(self, batch_size: int, graph=None):
        """trainer.py:399-413 + 428-431: warm-up state of this iteration; the optimizer steps once `accumulate` iterations
        have added their gradients to the flat buffer (each backward accumulates; the step zeroes it).  The gradient
        all-reduce happens once per optimizer step: the sum of the per-iteration all-reduces DDP would do."""
        acc, lrs, mom, wd = self.schedule_at(self.ni, batch_size)
        if self.ni - self.last_opt_step >= acc:
            self.all_reduce_gradients()
            if graph is not None:
                graph.replay(self.device)
            else:
                self.optimizer_step(lrs, mom, wd if self.schedule is not None else None)
            self.last_opt_step = self.ni
        self.ni += 1

    def compile(self, img, labels, warm_steps=2):
        """Capture the step (fixed shapes) once eager steps have allocated every static buffer and passed the first-step
        branch of the optimizer (`warm_steps` of them are run here on the given batch; pass 0 if >= 1 step already ran).
        Host-side inputs (image batch, labels, EMA decay) are copied into static device buffers before each replay;
        everything else is upa_* launches recorded once."""
        for _ in range(warm_steps):
            self.step(img, labels)
        if self.first_step:
            raise L.UpaError("compile(): run at least one eager step first (the optimizer's first step differs)")
        torch.cuda.synchronize(self.device)
        self._static_img = self.pool.get(("img_in",) + tuple(img.shape), tuple(img.shape), torch.float32, self.device)
        self._static_img.copy_(img)
        self._upload_labels(labels, img.shape[0])
        self._capturing = True
        try:
            g1 = R.HipGraph()
            if self.schedule is not None:
                # warm-up changes lr / momentum every iteration and accumulation skips optimizer steps: the forward +
                # backward is the graph, the three fused optimizer launches stay eager
                self._items = g1.capture(lambda: self.forward_backward(self._static_img, None), device=self.device)
                self._graphs = (g1, None)
            elif self.world_size == 1:
                def whole():
                    it = self.forward_backward(self._static_img, None)
                    self.optimizer_step()
                    return it
                self._items = g1.capture(whole, device=self.device)
                self._graphs = (g1,)
            else:
                g2 = R.HipGraph()
                self._items = g1.capture(lambda: self.forward_backward(self._static_img, None), device=self.device)
                g2.capture(self.optimizer_step, device=self.device)
                self._graphs = (g1, g2)
        finally:
            self._capturing = False
        if self._graphs[-1] is not None:
            self.updates -= 1  # a captured optimizer_step counted an update, but the capture itself executed nothing
        return self

    def _replay(self, img, labels):
        h = self.hyp
        self._static_img.copy_(img, non_blocking=True)
        self._upload_labels(labels, img.shape[0])

        def stage_decay():
            # the decay of THIS optimizer step travels as a kernel argument of a stream-ordered fill (no pinned host slot
            # that a later step could overwrite while an earlier copy is still queued when the host runs replays ahead)
            self.updates += 1
            self.ema_d_dev.fill_(h["ema_decay"] * (1 - math.exp(-self.updates / h["ema_tau"])))

        CV.bump_weights_generation()  # a replayed optimizer writes the weights through raw pointers as well
        if len(self._graphs) == 1:  # forward + backward + optimizer in one graph (single GPU, no schedule)
            stage_decay()
            self._graphs[0].replay(self.device)
            self.last_opt_step = self.ni
            self.ni += 1
            return self._items
        self._graphs[0].replay(self.device)
        if self._graphs[1] is None:  # schedule active: eager optimizer (it counts the update itself)
            self._maybe_optimize(img.shape[0])
        else:
            acc, _, _, _ = self.schedule_at(self.ni, img.shape[0])
            if self.ni - self.last_opt_step >= acc:
                stage_decay()
            self._maybe_optimize(img.shape[0], graph=self._graphs[1])
        return self._items

    def _upload_labels(self, labels, batch_size):
        imgsz_h, imgsz_w = self._imgsz
        gt, ngt = pack_targets(labels, batch_size, imgsz_h, imgsz_w)
        need = int(gt.shape[1])
        if self.gt_d is None or self.gt_d.shape[0] != batch_size or need > self.gt_d.shape[1]:
            if self._graphs is not None or self._capturing:
                raise L.UpaError(f"a batch needs {need} gt rows per image but the compiled step holds {self.max_gt}: set "
                                 f"trainer.max_gt >= {need} (<= {MAX_GT_CAP}) before compile()")
            self.max_gt = max(self.max_gt, need)
            self.gt_d = torch.zeros(batch_size, self.max_gt, 5, dtype=torch.float32, device=self.device)
            self.ngt_d = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        # through PINNED staging buffers (a ring of four, each guarded by the event of its last copy): a copy from pageable memory makes the
        # host wait for the stream, and the step's launches then trail the GPU by the time `pack_targets` + the copy took (0.45 ms of idle
        # GPU per yolov8s step in the kernel trace, tools/experiments/r05_train_gaps.py)
        rows = int(self.gt_d.shape[1])
        ring = self.__dict__.setdefault("_label_ring", [None] * 4)
        self._label_slot = (self.__dict__.get("_label_slot", -1) + 1) % len(ring)
        ent = ring[self._label_slot]
        if ent is None or ent[0].shape != (batch_size, rows, 5):
            ent = [torch.zeros(batch_size, rows, 5, dtype=torch.float32).pin_memory(), torch.zeros(batch_size, dtype=torch.int32).pin_memory(), None]
            ring[self._label_slot] = ent
        if ent[2] is not None:
            ent[2].synchronize()
        ent[0][:, :need].copy_(gt)
        ent[1].copy_(ngt)
        self.gt_d.copy_(ent[0], non_blocking=True)  # (rows past an image's count are never read: n_gt bounds them)
        self.ngt_d.copy_(ent[1], non_blocking=True)
        ent[2] = torch.cuda.Event()
        ent[2].record(torch.cuda.current_stream(self.device))

    def grad_norm(self) -> float:
        """Norm of the (unscaled) gradients in the flat buffer."""
        return float(torch.sqrt(self.grad_sumsq())[0]) / self.scaler.carried_scale(self._scaler_updated)

    # ---- the replicate steps of the multi-rank loop ------------------------------------------------------------------------
    def sync_ema_buffers(self, src: int = 0):
        """Rank `src`'s EMA float buffers (BatchNorm running mean / variance) on every rank, to be called before a validate that is
        sharded over the ranks (engine/trainer.py:695-698).  The parameters need nothing: after the gradient all-reduce every rank
        applies the same update, so P, M and the EMA parameters E are equal by construction; the BatchNorm running statistics are
        NOT (every rank normalises with its own shard's batch statistics, there is no SyncBN - trainer.py:253-258 only wraps in DDP,
        which broadcasts buffers rank 0 -> all at each forward), so without this the ranks would validate different models.  The
        reference loops over `ema.buffers()`; the EMA buffers live in ONE flat f32 tensor here (`ERB`), so it is one broadcast.
        The live model's buffers (`RB`) get the same treatment - DDP's `broadcast_buffers` (default True) keeps them rank 0's."""
        from ..parallel import broadcast_
        if self.nbuf:
            if self.ERB is not None:
                broadcast_(self.ERB[:self.nbuf], src)
            broadcast_(self.RB[:self.nbuf], src)

    def broadcast_stop(self, stop: bool, src: int = 0) -> bool:
        """The early-stopping decision of rank `src` on every rank (engine/trainer.py:505-508): all ranks must leave the epoch loop
        together or the next collective hangs."""
        from ..parallel import broadcast_flag
        return broadcast_flag(stop, src, self.device)

    def ema_state_dict(self):
        """The EMA weights under the reference's state_dict keys (for validation / checkpoints)."""
        out = {}
        for k, p in self.model.named_parameters():
            if p.requires_grad and self.E is not None:
                off = (p.data_ptr() - self.P.data_ptr()) // 4
                out[k] = self.E[off:off + p.numel()].view(p.shape)
            else:
                out[k] = p.detach()
        for k, b in self.model.named_buffers():
            if b.dtype.is_floating_point and self.ERB is not None:
                off = (b.data_ptr() - self.RB.data_ptr()) // 4
                out[k] = self.ERB[off:off + b.numel()].view(b.shape)
            else:
                out[k] = b
        return out


class DetectT(_Seq):
    """Train-mode Detect (head.py:116-126 returns the raw per-level maps) + v8DetectionLoss + its gradient."""

    def __init__(self, ctx, m: H.Detect, name, trainer):
        super().__init__(ctx)
        self.m, self.tr = m, trainer
        self.nl, self.nc, self.reg_max, self.no = m.nl, m.nc, m.reg_max, m.no
        self.br = []
        for i in range(m.nl):
            row = []
            for tag, seq in (("cv2", m.cv2[i]), ("cv3", m.cv3[i])):
                row.append([ConvT(ctx, seq[0].conv, seq[0].bn, seq[0]._act_code(), f"{name}.{tag}.{i}.0"),
                            ConvT(ctx, seq[1].conv, seq[1].bn, seq[1]._act_code(), f"{name}.{tag}.{i}.1"),
                            ConvT(ctx, seq[2], None, L.ACT_NONE, f"{name}.{tag}.{i}.2")])
            self.br.append(row)

    def convs(self):
        return [cv for row in self.br for seq in row for cv in seq]

    def forward(self, xs):
        c = self.ctx
        dev = c.device
        nb = 4 * self.reg_max
        self.xs = list(xs)
        self.raw, self.raw32 = [None] * len(xs), [None] * len(xs)

        def level(i, x):
            n, _, h, w = x.shape
            raw = _new(n, self.no, h, w, c.dtype, dev, (id(self), "raw", i))
            for k, (seq, out) in enumerate(zip(self.br[i], (raw[:, :nb], raw[:, nb:]))):
                t = seq[1].forward(seq[0].forward(x))
                seq[2].forward(t, out=out)
            self.raw[i] = raw
            if c.dtype == torch.float32:
                self.raw32[i] = raw
            else:  # the loss reads f32 maps
                r32 = _new(n, self.no, h, w, torch.float32, dev, (id(self), "raw32", i))
                vs, vd = R.view_of(raw), R.view_of(r32)
                L.check(L.lib().upa_cast_view(vs.ptr, vs.dtype, vs.ld, vd.ptr, vd.dtype, vd.ld, vs.n * vs.h * vs.w, vs.c,
                                              _s(dev)), "cast_view")
                self.raw32[i] = r32

        # The levels are independent chains (head.py:116-126).  The 40 x 40 / 20 x 20 levels are strings of 5-15 us launches (a few
        # hundred workgroups each): on a second stream they run beside the 80 x 80 level's chip-filling launches instead of after them.
        self._fork_levels(lambda: level(0, xs[0]), lambda: [level(i, xs[i]) for i in range(1, len(xs))], len(xs) > 1)
        return self.raw

    def _fork_levels(self, first, rest, fork: bool):
        """`first()` on the current stream, `rest()` on the context's level stream (own reduction workspace), joined at the end."""
        c = self.ctx
        if not (fork and self.fork_levels):
            first()
            rest()
            return
        main = torch.cuda.current_stream(c.device)
        ev = torch.cuda.Event()
        ev.record(main)
        c.level_stream.wait_event(ev)
        c._ws_sel = 1
        try:
            with torch.cuda.stream(c.level_stream):
                rest()
                done = torch.cuda.Event()
                done.record(c.level_stream)
        finally:
            c._ws_sel = 0
        first()
        main.wait_event(done)

    fork_levels = True  # (False: every level on the main stream, as before round 5 - the A/B switch)

    def loss_backward(self, labels, batch_size):
        """v8DetectionLoss (utils/loss.py:471-528) + gradient; then backward through the head into the neck outputs."""
        c, tr = self.ctx, self.tr
        dev = c.device
        lib = L.lib()
        h = tr.hyp
        imgsz_h = int(self.raw[0].shape[2] * float(self.m.stride[0]))
        imgsz_w = int(self.raw[0].shape[3] * float(self.m.stride[0]))
        tr._imgsz = (imgsz_h, imgsz_w)
        if labels is not None:  # None: already uploaded into the static buffers (graph replay)
            tr._upload_labels(labels, batch_size)
        gt_d, ngt_d = tr.gt_d, tr.ngt_d
        max_gt = int(gt_d.shape[1])
        grads32 = []
        for i, r in enumerate(self.raw32):
            n, ch, hh, ww = r.shape
            grads32.append(_new(n, ch, hh, ww, torch.float32, dev, (id(self), "graw32", i)))
        nl = self.nl
        FP = C.c_void_p * nl
        feats = FP(*[R.view_of(r).ptr for r in self.raw32])
        grads = FP(*[R.view_of(g).ptr for g in grads32])
        IA = C.c_int * nl
        hs = IA(*[int(r.shape[2]) for r in self.raw32])
        ws = IA(*[int(r.shape[3]) for r in self.raw32])
        lds = IA(*[R.view_of(r).ld for r in self.raw32])
        strides = (C.c_float * nl)(*[float(s) for s in self.m.stride])
        A = sum(int(r.shape[2]) * int(r.shape[3]) for r in self.raw32)
        nbytes = lib.upa_detection_loss_workspace_bytes(batch_size, A, max_gt)
        wsb = tr.pool.get(("loss_ws", batch_size, A, max_gt), (nbytes,), torch.uint8, dev)
        items = tr.pool.get(("loss_items",), (3,), torch.float32, dev)
        L.check(lib.upa_detection_loss_scaled(C.cast(feats, C.c_void_p), C.cast(grads, C.c_void_p), C.cast(hs, C.c_void_p),
                                              C.cast(ws, C.c_void_p), C.cast(lds, C.c_void_p), C.cast(strides, C.c_void_p), nl,
                                              batch_size, self.nc, self.reg_max, gt_d.data_ptr(), ngt_d.data_ptr(), max_gt,
                                              h["box"], h["cls"], h["dfl"], 1.0, tr.scaler.ptr(), items.data_ptr(), wsb.data_ptr(),
                                              nbytes, _s(dev)), "detection_loss")
        nb = 4 * self.reg_max

        def level_bwd(i):
            g32 = grads32[i]
            if c.dtype == torch.float32:
                graw = g32
            else:
                n, ch, hh, ww = g32.shape
                graw = _new(n, ch, hh, ww, c.dtype, dev, (id(self), "graw", i))
                vs, vd = R.view_of(g32), R.view_of(graw)
                L.check(lib.upa_cast_view(vs.ptr, vs.dtype, vs.ld, vd.ptr, vd.dtype, vd.ld, vs.n * vs.h * vs.w, vs.c, _s(dev)),
                        "cast_view")
            # feature map feeding level i: route the gradient into the producing node
            src = tr.nodes[self.m.f[i]]
            dx, acc = tr._grad_of(src)
            for k, (seq, dy) in enumerate(zip(self.br[i], (graw[:, :nb], graw[:, nb:]))):
                vt1, vt0 = R.view_of(seq[2].x), R.view_of(seq[1].x)
                d1 = _new(vt1.n, vt1.c, vt1.h, vt1.w, c.dtype, dev, (id(self), "d1", i, k))
                d0 = _new(vt0.n, vt0.c, vt0.h, vt0.w, c.dtype, dev, (id(self), "d0", i, k))
                seq[2].backward(dy, d1, False)
                seq[1].backward(d1, d0, False)
                seq[0].backward(d0, dx, acc or k > 0)

        # as in the forward pass: the small levels' chains beside the 80 x 80 level's (they write the gradients of different neck outputs)
        self._fork_levels(lambda: level_bwd(0), lambda: [level_bwd(i) for i in range(1, nl)], nl > 1)
        return items


def pack_targets(labels, batch_size, imgsz_h, imgsz_w, min_rows=MAX_GT):
    """loss.py:445-461 on the host: (n,) image index, (n,) class, (n,4) normalised xywh -> padded (B, rows, 5)
    [cls, x1, y1, x2, y2] in pixels + per-image counts (host tensors; a few hundred bytes per step).  `rows` = the largest
    per-image count rounded up to a multiple of 64 (at least `min_rows`), as the reference pads to counts.max().  Boxes
    whose xyxy coordinates sum to zero are dropped: the reference masks them (`mask_gt = gt_bboxes.sum(2) > 0`,
    loss.py:489), so they never take part in the assignment."""
    bi = labels["batch_idx"].view(-1).cpu().long()
    cls = labels["cls"].view(-1).cpu().float()
    bb = labels["bboxes"].view(-1, 4).cpu().float()
    scale = torch.tensor([imgsz_w, imgsz_h, imgsz_w, imgsz_h], dtype=torch.float32)
    xywh = bb * scale
    xy, wh = xywh[:, :2], xywh[:, 2:] / 2
    xyxy = torch.cat([xy - wh, xy + wh], 1)
    keep = xyxy.sum(1) > 0
    per = [((bi == j) & keep).nonzero().view(-1) for j in range(batch_size)]
    most = max([int(ix.numel()) for ix in per] + [1])
    rows = max(min_rows, (most + 63) // 64 * 64)
    if rows > MAX_GT_CAP:
        raise L.UpaError(f"an image has {most} boxes; the loss kernels hold at most {MAX_GT_CAP} per image")
    gt = torch.zeros(batch_size, rows, 5)
    ngt = torch.zeros(batch_size, dtype=torch.int32)
    for j, ix in enumerate(per):
        k = int(ix.numel())
        if k:
            gt[j, :k, 0] = cls[ix]
            gt[j, :k, 1:] = xyxy[ix]
        ngt[j] = k
    return gt, ngt
