"""-m gpu: training-step kernels (include/upa.h "training step") against torch autograd on the CPU (per operator) and
against the reference-pinned oracle / goldens (loss, whole step).  f32 = parity mode; bf16 = the AMP-like perf mode."""

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from ultralytics_pro_amd.utils import procedural as P

pytestmark = pytest.mark.gpu


def _ctx(dtype):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine import trainer as T
    return T._Ctx(DEV, dtype)


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


CONV_TRAIN_CASES = [
    # cin, cout, k, s, N, H, W
    (16, 32, 3, 1, 2, 12, 20),
    (32, 64, 3, 2, 2, 16, 16),
    (64, 64, 1, 1, 3, 10, 10),
    (48, 80, 3, 1, 1, 9, 13),
    (128, 64, 3, 2, 2, 13, 11),
    (80, 144, 1, 1, 2, 8, 8),
    (256, 128, 1, 1, 3, 9, 11),   # pointwise bf16 MFMA weight gradient: 2 x 1 blocks of 128, ragged pixel count
    (192, 136, 1, 1, 1, 16, 16),  # channel counts that are not multiples of the 128 block
    (64, 128, 3, 2, 2, 64, 64),   # stride 2, large enough for the one-launch data gradient (phase conv with the interleaving epilogue)
    (64, 128, 3, 2, 3, 46, 62),   # ... odd output map (23 x 31): phase pixels past dx's last row / column are dropped
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_TRAIN_CASES, ids=[f"c{c[0]}-{c[1]}k{c[2]}s{c[3]}_{c[5]}x{c[6]}" for c in CONV_TRAIN_CASES])
def test_conv_bn_silu_forward_backward(case, dtype):
    """Conv = conv2d -> BatchNorm2d (batch statistics) -> SiLU (nn/modules/conv.py:177-186): output, running statistics,
    dx, dW, dgamma, dbeta of the HIP layer vs torch autograd on the CPU."""
    from tests.hip_utils import DEV, bf16_round, to_cpu_nchw, to_dev_nhwc
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine import trainer as T
    cin, cout, k, s, N, H, W = case
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
    bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        conv.weight.copy_(P.uniform(f"tw{case}", tuple(conv.weight.shape), -1, 1) * (2.0 / (cin * k * k)) ** 0.5)
        bn.weight.copy_(P.uniform(f"tg{case}", (cout,), 0.5, 1.5))
        bn.bias.copy_(P.uniform(f"tb{case}", (cout,), -0.3, 0.3))
    x = P.uniform(f"tx{case}", (N, cin, H, W), -1, 1)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
        with torch.no_grad():
            conv.weight.copy_(bf16_round(conv.weight))
    xr = x.clone().requires_grad_(True)
    ref = F.silu(bn(conv(xr)))
    dy = P.uniform(f"tdy{case}", tuple(ref.shape), -1, 1)
    if dtype == torch.bfloat16:
        dy = bf16_round(dy)
    ref.backward(dy)
    # HIP
    ctx = _ctx(dtype)
    dconv, dbn = nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(DEV), nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03).to(DEV)
    with torch.no_grad():
        dconv.weight.copy_(conv.weight)
        dbn.weight.copy_(bn.weight)
        dbn.bias.copy_(bn.bias)
    for p in (dconv.weight, dbn.weight, dbn.bias):
        p.grad = torch.zeros_like(p)
    op = T.ConvT(ctx, dconv, dbn, 1, "t")
    op.pack()
    xd = to_dev_nhwc(x, dtype)
    y = op.forward(xd)
    dx = R.alloc_nhwc(N, cin, H, W, dtype, DEV)
    op.backward(to_dev_nhwc(dy, dtype), dx, False)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert _rel(to_cpu_nchw(y), ref.detach()) <= tol
    assert _rel(dbn.running_mean.cpu(), bn.running_mean) <= tol and _rel(dbn.running_var.cpu(), bn.running_var) <= tol
    assert _rel(to_cpu_nchw(dx), xr.grad) <= tol
    assert _rel(dconv.weight.grad.cpu(), conv.weight.grad) <= tol
    assert _rel(dbn.weight.grad.cpu(), bn.weight.grad) <= tol
    assert _rel(dbn.bias.grad.cpu(), bn.bias.grad) <= tol
    # accumulate: a second backward doubles the parameter gradients and adds into dx
    op.backward(to_dev_nhwc(dy, dtype), dx, True)
    torch.cuda.synchronize()  # the weight gradient runs on the trainer context's side stream (the trainer joins it before the optimizer)
    assert _rel(dconv.weight.grad.cpu(), 2 * conv.weight.grad) <= tol
    assert _rel(to_cpu_nchw(dx), 2 * xr.grad) <= tol


WGRAD_CASES = [
    # cin, cout, k, s, N, H, W, ldx (channels of the tensor x is a slice of), accumulate
    (96, 64, 1, 1, 2, 20, 20, 96, 0),      # pointwise ring: one 128 x 128 block, half of it outside the tensors
    (768, 512, 1, 1, 3, 5, 7, 768, 1),     # 24 blocks, 105 pixels (less than two 64-pixel stages per workgroup), accumulate
    (40, 200, 1, 1, 1, 9, 9, 64, 0),       # ragged channel counts, x a channel slice of a wider tensor
    (128, 128, 1, 1, 2, 80, 80, 128, 0),   # more stages than the ring holds
    (64, 64, 3, 1, 2, 20, 20, 64, 0),      # 3x3 ring: image edge inside every tile, 20 columns = one tile and a quarter
    (72, 136, 3, 1, 3, 13, 17, 96, 1),     # ragged channels (2 x 3 blocks), odd map, slice view, accumulate
    (128, 64, 3, 1, 2, 5, 5, 128, 0),      # map smaller than a tile
    (64, 64, 3, 1, 4, 40, 48, 64, 0),      # interior tiles, several stages per workgroup
    (32, 64, 3, 2, 2, 32, 32, 32, 0),      # stride 2: de-interleaved halo columns
    (64, 128, 3, 2, 2, 15, 21, 64, 1),     # stride 2, odd sizes
    (3, 32, 3, 2, 2, 40, 56, 8, 0),        # the stem: 3 channels stored as 8, narrow ring form
    (16, 32, 3, 1, 2, 12, 20, 16, 0),      # 16 input channels: two 16-byte groups, register-staged narrow form
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[f"c{c[0]}-{c[1]}k{c[2]}s{c[3]}_{c[5]}x{c[6]}" for c in WGRAD_CASES])
def test_weight_gradient_bf16_matches_torch(case):
    """upa_conv2d_wgrad (bf16 operands, f32 sums: LDS-DMA ring kernels + partial-sum reduce) against torch's conv2d_weight on
    the same bf16-rounded operands in f32.  Products of bf16 values are exact in f32, only the summation order differs:
    tolerance 2e-5 of the largest |dW|."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    cin, cout, k, s, N, H, W, ldx, accumulate = case
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    g = torch.Generator().manual_seed(cin * 1000 + cout + H)
    xw = torch.zeros(N, H, W, ldx, dtype=torch.bfloat16)
    xw[..., :cin] = torch.randn(N, H, W, cin, generator=g).to(torch.bfloat16)
    if ldx > cin and cin >= 8:
        xw[..., cin:] = 7.0  # neighbours of the slice must not leak into dW
    dzc = torch.randn(N, OH, OW, cout, generator=g).to(torch.bfloat16)
    x = xw.to(DEV)[..., :cin].permute(0, 3, 1, 2)
    dz = dzc.to(DEV).permute(0, 3, 1, 2)
    dw0 = torch.randn(cout, cin, k, k, generator=g)
    dw = dw0.clone().to(DEV) if accumulate else torch.full((cout, cin, k, k), float("nan"), device=DEV)
    vx, vz = R.view_of(x), R.view_of(dz)
    assert vx.ld == ldx
    lib = L.lib()
    ws = torch.empty(lib.upa_conv2d_wgrad_workspace_bytes(cin, cout, k), dtype=torch.uint8, device=DEV)
    L.check(lib.upa_conv2d_wgrad(vx.ptr, N, H, W, cin, vx.ld, vz.ptr, cout, vz.ld, dw.data_ptr(), k, s, pad, accumulate, vx.dtype,
                                 ws.data_ptr(), ws.numel(), L.current_stream(DEV)), "wgrad")
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(xw[..., :cin].permute(0, 3, 1, 2).float(), (cout, cin, k, k), dzc.permute(0, 3, 1, 2).float(),
                                      stride=s, padding=pad)
    if accumulate:
        ref = ref + dw0
    got = dw.cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # the same call again gives the same bits (fixed summation order)
    dw2 = dw0.clone().to(DEV) if accumulate else torch.zeros_like(dw)
    L.check(lib.upa_conv2d_wgrad(vx.ptr, N, H, W, cin, vx.ld, vz.ptr, cout, vz.ld, dw2.data_ptr(), k, s, pad, accumulate, vx.dtype,
                                 ws.data_ptr(), ws.numel(), L.current_stream(DEV)), "wgrad")
    assert torch.equal(dw2.cpu(), got)


EPI_STATS_CASES = [
    # cin, cout, k, s, N, H, W: shapes of the yolov8s training step that run on kernels with a statistics epilogue, small batches
    (128, 128, 3, 1, 4, 40, 40),   # conv_big 256 px x 128 ch (4 x 4 tiles per wave), 25 workgroups
    (128, 128, 3, 1, 2, 20, 20),   # 128-px workgroups, ragged tile (20 x 20 = 3.1 tiles of 128)
    (256, 256, 3, 1, 3, 20, 20),   # two column blocks of 128 channels: rows are filled by blockIdx.y slices
    (64, 128, 3, 2, 2, 80, 80),    # stride 2
    (128, 64, 3, 1, 2, 40, 40),    # 64-channel columns (8 x 1 waves)
    (768, 512, 1, 1, 2, 20, 20),   # pointwise on conv_big (cin >= 512): flattened pixel row, ragged last tile
    (256, 64, 3, 1, 1, 40, 40),    # 64 channels, 128-px workgroups (4 x 2 waves x 2 x 2 tiles)
    (384, 256, 1, 1, 2, 40, 40),   # streaming pointwise kernel, two column blocks of eight n-tiles
    (96, 64, 1, 1, 3, 32, 32),     # ... four n-tiles, three k-tiles
    (64, 32, 1, 1, 2, 24, 24),     # ... two n-tiles
    (64, 96, 1, 1, 1, 80, 80),     # ... six n-tiles
    (256, 128, 1, 1, 32, 20, 20),  # ... persistent workgroups with several pixel groups per wave
    (64, 64, 3, 1, 6, 40, 40),     # weights-stationary 3x3 (conv_ws3): every wave owns its channels
    (64, 64, 3, 1, 9, 44, 52),     # ... ragged tiles (44 = 5.5 x 8 rows, 52 = 3.25 x 16 columns), more tiles than workgroups
    (128, 80, 1, 1, 2, 20, 20),    # five n-tiles: no such epilogue - the library's conv -> reduce -> combine sequence
    (32, 64, 3, 2, 2, 32, 32),     # ... and a shape on the generic kernel
]


@pytest.mark.parametrize("case", EPI_STATS_CASES, ids=[f"c{c[0]}-{c[1]}k{c[2]}s{c[3]}_{c[5]}x{c[6]}" for c in EPI_STATS_CASES])
def test_conv_batch_statistics_from_the_convolution_epilogue(case):
    """upa_conv2d_bn_stats (training forward of Conv up to the normalisation, conv.py:177-186): z bit-identical to the plain convolution
    and mean / var / running statistics equal to a reduction pass over z (upa_opts.no_epi_stats = 1: the same call on the three-launch
    sequence) to f32 rounding, against an f64 torch reduction of the stored z, and bit-reproducible."""
    from tests.hip_utils import DEV, to_dev_nhwc
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    cin, cout, k, s, N, H, W = case
    lib = L.lib()
    st = L.current_stream(DEV)
    w = P.uniform(f"ew{case}", (cout, cin, k, k), -1, 1) * (2.0 / (cin * k * k)) ** 0.5
    wp = torch.empty(lib.upa_conv_packed_weight_bytes(cout, cin, k, L.UPA_BF16), dtype=torch.uint8, device=DEV)
    wd = w.to(DEV).contiguous()
    L.check(lib.upa_pack_conv_weight_dev(wd.data_ptr(), cout, cin, k, L.UPA_BF16, 0, wp.data_ptr(), st))
    x = to_dev_nhwc(P.uniform(f"ex{case}", (N, cin, H, W), -1, 1), torch.bfloat16)
    vx = R.view_of(x)
    oh, ow = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    npix = N * oh * ow
    ws = torch.zeros(lib.upa_channel_reduce_workspace_bytes(cout) // 8, dtype=torch.float64, device=DEV)
    rm0, rv0 = P.uniform(f"erm{case}", (cout,), -0.1, 0.1).to(DEV), P.uniform(f"erv{case}", (cout,), 0.5, 1.5).to(DEV)

    def run(no_epi):
        z = R.alloc_nhwc(N, cout, oh, ow, torch.bfloat16, DEV)
        vz = R.view_of(z)
        m, v, rm, rv = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), rm0.clone(), rv0.clone()
        with R.use_opts(L.Opts(no_epi_stats=int(no_epi))):
            L.check(lib.upa_conv2d_bn_stats(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, wp.data_ptr(), vz.ptr, cout, vz.ld, k, s, k // 2, 0.03,
                                            m.data_ptr(), v.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws.data_ptr(), L.UPA_BF16,
                                            R.opts_ptr(), st))
        torch.cuda.synchronize()
        return z, (m, v, rm, rv)

    z_sep, st_sep = run(True)
    z_epi, st_epi = run(False)
    assert torch.equal(z_sep, z_epi)
    for a, b in zip(st_epi, st_sep):
        assert _rel(a.double(), b.double()) <= 2e-6
    z64 = z_epi.permute(0, 2, 3, 1).reshape(-1, cout).double() if z_epi.dim() == 4 else z_epi.double()
    assert z64.shape[0] == npix
    m_ref, v_ref = z64.mean(0), z64.var(0, unbiased=False)
    assert _rel(st_epi[0].double(), m_ref) <= 1e-5 and _rel(st_epi[1].double(), v_ref) <= 1e-5
    assert _rel(st_epi[3].double(), 0.97 * rv0.double() + 0.03 * v_ref * npix / (npix - 1)) <= 1e-5
    for _ in range(3):
        z2, st2 = run(False)
        assert torch.equal(z2, z_epi) and all(torch.equal(a, b) for a, b in zip(st2, st_epi))


def test_head_conv_with_bias_and_stem_input():
    """Plain nn.Conv2d(+bias) head outputs (head.py:98-100) and the 3-channel stem (padded NHWC input, no dx)."""
    from tests.hip_utils import DEV, to_cpu_nchw, to_dev_nhwc
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine import trainer as T
    ctx = _ctx(torch.float32)
    conv = nn.Conv2d(64, 80, 1)
    with torch.no_grad():
        conv.weight.copy_(P.uniform("hw", tuple(conv.weight.shape), -0.2, 0.2))
        conv.bias.copy_(P.uniform("hb", (80,), -1, 1))
    x = P.uniform("hx", (2, 64, 8, 8), -1, 1)
    xr = x.clone().requires_grad_(True)
    ref = conv(xr)
    dy = P.uniform("hdy", tuple(ref.shape), -1, 1)
    ref.backward(dy)
    d = nn.Conv2d(64, 80, 1).to(DEV)
    with torch.no_grad():
        d.weight.copy_(conv.weight)
        d.bias.copy_(conv.bias)
    for p in d.parameters():
        p.grad = torch.zeros_like(p)
    op = T.ConvT(ctx, d, None, 0, "h")
    op.pack()
    y = op.forward(to_dev_nhwc(x))
    dx = R.alloc_nhwc(2, 64, 8, 8, torch.float32, DEV)
    op.backward(to_dev_nhwc(dy), dx, False)
    assert _rel(to_cpu_nchw(y), ref.detach()) <= 1e-4
    assert _rel(to_cpu_nchw(dx), xr.grad) <= 1e-4
    assert _rel(d.weight.grad.cpu(), conv.weight.grad) <= 1e-4 and _rel(d.bias.grad.cpu(), conv.bias.grad) <= 1e-4
    # stem: cin = 3, stride 2, input padded to 4 channels
    conv = nn.Conv2d(3, 16, 3, 2, 1, bias=False)
    bn = nn.BatchNorm2d(16, eps=1e-3, momentum=0.03)
    x = P.uniform("sx", (2, 3, 32, 48), 0, 1)
    ref = F.silu(bn(conv(x)))
    dy = P.uniform("sdy", tuple(ref.shape), -1, 1)
    ref.backward(dy)
    dc, db = nn.Conv2d(3, 16, 3, 2, 1, bias=False).to(DEV), nn.BatchNorm2d(16, eps=1e-3, momentum=0.03).to(DEV)
    with torch.no_grad():
        dc.weight.copy_(conv.weight)
    for p in list(dc.parameters()) + list(db.parameters()):
        p.grad = torch.zeros_like(p)
    op = T.ConvT(ctx, dc, db, 1, "s")
    op.pack()
    buf = torch.zeros(2, 32, 48, 4, device=DEV)
    buf[..., :3] = x.permute(0, 2, 3, 1).to(DEV)  # test-side layout plumbing
    y = op.forward(buf.permute(0, 3, 1, 2))
    op.backward(to_dev_nhwc(dy), None, False)
    assert _rel(to_cpu_nchw(y), ref.detach()) <= 2e-4
    assert _rel(dc.weight.grad.cpu(), conv.weight.grad) <= 2e-4
    # the same stem in bf16 mode: input padded to 8 channels, narrow-input bf16 MFMA weight gradient (16-channel block)
    from tests.hip_utils import bf16_round
    ctxb = _ctx(torch.bfloat16)
    convb = nn.Conv2d(3, 32, 3, 2, 1, bias=False)
    with torch.no_grad():
        convb.weight.copy_(bf16_round(convb.weight))
    bnb = nn.BatchNorm2d(32, eps=1e-3, momentum=0.03)
    xb = bf16_round(P.uniform("sxb", (2, 3, 40, 64), 0, 1))
    refb = F.silu(bnb(convb(xb)))
    dyb = bf16_round(P.uniform("sdyb", tuple(refb.shape), -1, 1))
    refb.backward(dyb)
    dcb, dbb = nn.Conv2d(3, 32, 3, 2, 1, bias=False).to(DEV), nn.BatchNorm2d(32, eps=1e-3, momentum=0.03).to(DEV)
    with torch.no_grad():
        dcb.weight.copy_(convb.weight)
    for p in list(dcb.parameters()) + list(dbb.parameters()):
        p.grad = torch.zeros_like(p)
    opb = T.ConvT(ctxb, dcb, dbb, 1, "sb")
    opb.pack()
    bufb = torch.zeros(2, 40, 64, 8, device=DEV, dtype=torch.bfloat16)
    bufb[..., :3] = xb.permute(0, 2, 3, 1).to(DEV).to(torch.bfloat16)
    yb = opb.forward(bufb.permute(0, 3, 1, 2))
    opb.backward(to_dev_nhwc(dyb, torch.bfloat16), None, False)
    assert _rel(to_cpu_nchw(yb), refb.detach()) <= 3e-2
    assert _rel(dcb.weight.grad.cpu(), convb.weight.grad) <= 3e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_pool_upsample_backward(dtype):
    from tests.hip_utils import DEV, bf16_round, to_cpu_nchw, to_dev_nhwc
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    x = P.uniform("px", (2, 16, 9, 11), -1, 1)
    x = (x * 4).round() / 4  # plenty of ties: the first-maximum rule matters
    st = L.current_stream(DEV)
    for k, pad in ((5, 2), (3, 1)):  # 5 / 1 / 2 in bf16: the unrolled SPPF kernels; everything else: the generic window walk
        xr = x.clone().requires_grad_(True)
        ref = F.max_pool2d(xr, k, 1, pad)
        dy = P.uniform(f"pdy{k}", tuple(ref.shape), -1, 1)
        if dtype == torch.bfloat16:
            dy = bf16_round(dy)
        ref.backward(dy)
        xd, dyd = to_dev_nhwc(x, dtype), to_dev_nhwc(dy, dtype)  # keep the device tensors alive: views are raw pointers
        vx, vdy = R.view_of(xd), R.view_of(dyd)
        dx = R.alloc_nhwc(2, 16, 9, 11, dtype, DEV)
        vdx = R.view_of(dx)
        nws = L.lib().upa_maxpool2d_bwd_workspace_bytes(2, 9, 11, 16, k, 1, pad)
        wsb = torch.empty(nws, dtype=torch.uint8, device=DEV)
        L.check(L.lib().upa_maxpool2d_bwd(vx.ptr, vdy.ptr, 2, 9, 11, 16, vx.ld, vdy.ld, k, 1, pad, vdx.ptr, vdx.ld, 0, vx.dtype,
                                          wsb.data_ptr(), nws, st))
        assert _rel(to_cpu_nchw(dx), xr.grad) <= (1e-6 if dtype == torch.float32 else 2e-2)
        if dtype == torch.bfloat16 and k == 5:  # ... and accumulating into an existing gradient
            L.check(L.lib().upa_maxpool2d_bwd(vx.ptr, vdy.ptr, 2, 9, 11, 16, vx.ld, vdy.ld, k, 1, pad, vdx.ptr, vdx.ld, 1, vx.dtype,
                                              wsb.data_ptr(), nws, st))
            assert _rel(to_cpu_nchw(dx), 2 * xr.grad) <= 3e-2
    # nearest 2x upsample backward
    u = P.uniform("ux", (2, 16, 5, 7), -1, 1).requires_grad_(True)
    up = F.interpolate(u, scale_factor=2, mode="nearest")
    dyu = P.uniform("udy", tuple(up.shape), -1, 1)
    if dtype == torch.bfloat16:
        dyu = bf16_round(dyu)
    up.backward(dyu)
    dyud = to_dev_nhwc(dyu, dtype)
    vdy = R.view_of(dyud)
    dxu = R.alloc_nhwc(2, 16, 5, 7, dtype, DEV)
    vdx = R.view_of(dxu)
    L.check(L.lib().upa_upsample2x_bwd(vdy.ptr, 2, 5, 7, 16, vdy.ld, vdx.ptr, vdx.ld, 0, vdy.dtype, st))
    assert _rel(to_cpu_nchw(dxu), u.grad) <= (1e-6 if dtype == torch.float32 else 2e-2)


def test_sgd_nesterov_clip_ema_matches_torch():
    """engine/trainer.py:674-682 on a flat buffer vs clip_grad_norm_ + torch.optim.SGD(nesterov) + the EMA formula, 3 steps."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd import _lib as L
    n = 10000
    p = P.uniform("sp", (n,), -1, 1)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4)
    ema_r = p.clone()
    P_, M_, E_ = p.to(DEV), torch.zeros(n, device=DEV), p.to(DEV)
    sumsq = torch.zeros(1, dtype=torch.float64, device=DEV)
    sumsq_ws = torch.zeros(L.lib().upa_sumsq_workspace_bytes() // 8, dtype=torch.float64, device=DEV)
    st = L.current_stream(DEV)
    for step in range(3):
        g = P.uniform(f"sg{step}", (n,), -1, 1) * (50.0 if step == 1 else 0.01)  # step 1 clips
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 10.0)
        opt.step()
        d = 0.9999 * (1 - np.exp(-(step + 1) / 2000.0))
        ema_r = ema_r * d + (1 - d) * pr.detach()
        G_ = g.to(DEV)
        L.check(L.lib().upa_sumsq(G_.data_ptr(), n, sumsq.data_ptr(), 0, sumsq_ws.data_ptr(), st))
        L.check(L.lib().upa_sgd_nesterov_ema(P_.data_ptr(), G_.data_ptr(), M_.data_ptr(), E_.data_ptr(), n, sumsq.data_ptr(), 10.0,
                                             0.01, 0.9, 5e-4, int(step == 0), float(d), None, 1, st))
        assert float(G_.abs().max()) == 0.0
        assert float((P_.cpu() - pr.detach()).abs().max()) <= 2e-6
        assert float((E_.cpu() - ema_r).abs().max()) <= 2e-6


def _loss_inputs(B, hw, nc=80, seed=0):
    feats = [P.uniform(f"lf{seed}:{i}", (B, 64 + nc, h, w), -2, 2) for i, (h, w) in enumerate(hw)]
    for f in feats:  # class logits mostly negative (as in a real head), a few positive
        f[:, 64:] = f[:, 64:] * 2 - 3
    return feats


@pytest.mark.parametrize("case", [(2, [(16, 16), (8, 8), (4, 4)]), (4, [(20, 12), (10, 6), (5, 3)])], ids=["sq128", "rect160x96"])
def test_detection_loss_and_gradient_match_oracle(case):
    """v8DetectionLoss + TaskAlignedAssigner (utils/loss.py:471-528, utils/tal.py:12-316): loss items and the gradient
    wrt the head maps from the HIP kernels vs the reference-pinned oracle (autograd) on identical inputs."""
    from oracle.loss import v8_detection_loss
    from tests.hip_utils import DEV, to_cpu_nchw, to_dev_nhwc
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine import trainer as T
    import ctypes as C
    B, hw = case
    _check_loss_vs_oracle(B, hw, P.synthetic_labels(B, seed=3))


def test_detection_loss_class_count_not_a_multiple_of_four():
    """6 classes: the class term takes the element-per-thread kernel (the four-classes-per-thread form needs nc % 4 == 0 and
    16-byte aligned rows), vs the oracle."""
    labels = P.synthetic_labels(2, seed=3)
    labels = dict(labels, cls=labels["cls"] % 6)
    _check_loss_vs_oracle(2, [(16, 16), (8, 8), (4, 4)], labels, nc=6)


def _crowded_labels(B, per_image, seed=5):
    """`per_image[j]` boxes in image j (mosaic-like crowding, > 64 per image), plus two all-zero boxes that the reference
    masks out (mask_gt, loss.py:489)."""
    bi, cl, bb = [], [], []
    for j, k in enumerate(per_image):
        u = P.hash_uniform(f"crowd:{seed}:{j}", 5 * k).reshape(k, 5)
        bi.append(np.full(k, j, dtype=np.float32))
        cl.append(np.floor(u[:, 0] * 80).astype(np.float32))
        bb.append(np.stack([0.1 + 0.8 * u[:, 1], 0.1 + 0.8 * u[:, 2], 0.03 + 0.3 * u[:, 3], 0.03 + 0.3 * u[:, 4]], 1).astype(np.float32))
    bi.append(np.array([0, 1], dtype=np.float32))
    cl.append(np.array([3, 4], dtype=np.float32))
    bb.append(np.zeros((2, 4), dtype=np.float32))
    return {"batch_idx": torch.from_numpy(np.concatenate(bi)), "cls": torch.from_numpy(np.concatenate(cl)),
            "bboxes": torch.from_numpy(np.concatenate(bb))}


def test_detection_loss_more_than_64_boxes_per_image_matches_oracle():
    """the gt capacity is a run-time argument (the reference pads to counts.max(), loss.py:445-461): 150 / 70 / 3
    boxes per image plus zero-sum boxes, vs the oracle."""
    from ultralytics_pro_amd.engine import trainer as T
    labels = _crowded_labels(3, [150, 70, 3])
    gt, ngt = T.pack_targets(labels, 3, 160, 96)
    assert tuple(gt.shape) == (3, 192, 5) and ngt.tolist() == [150, 70, 3]
    _check_loss_vs_oracle(3, [(20, 12), (10, 6), (5, 3)], labels)


def _check_loss_vs_oracle(B, hw, labels, nc=80):
    from oracle.loss import v8_detection_loss
    from tests.hip_utils import DEV, to_cpu_nchw, to_dev_nhwc
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine import trainer as T
    import ctypes as C
    feats = _loss_inputs(B, hw, nc=nc)
    strides = torch.tensor([8.0, 16.0, 32.0])
    fr = [f.clone().requires_grad_(True) for f in feats]
    loss, items = v8_detection_loss(fr, labels, strides, nc=nc)
    loss.sum().backward()
    # HIP
    fd = [to_dev_nhwc(f) for f in feats]
    gd = [R.alloc_nhwc(*f.shape[:1], f.shape[1], f.shape[2], f.shape[3], torch.float32, DEV) for f in feats]
    gt, ngt = T.pack_targets(labels, B, hw[0][0] * 8, hw[0][1] * 8)
    gt_d, ngt_d = gt.to(DEV), ngt.to(DEV)
    nl = 3
    FP, IA = C.c_void_p * nl, C.c_int * nl
    fp = FP(*[R.view_of(t).ptr for t in fd])
    gp = FP(*[R.view_of(t).ptr for t in gd])
    hs, ws = IA(*[h for h, _ in hw]), IA(*[w for _, w in hw])
    lds = IA(*[R.view_of(t).ld for t in fd])
    st_ = (C.c_float * nl)(8.0, 16.0, 32.0)
    A = sum(h * w for h, w in hw)
    max_gt = int(gt.shape[1])
    nbytes = L.lib().upa_detection_loss_workspace_bytes(B, A, max_gt)
    wsb = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    out = torch.zeros(3, device=DEV)
    L.check(L.lib().upa_detection_loss(C.cast(fp, C.c_void_p), C.cast(gp, C.c_void_p), C.cast(hs, C.c_void_p), C.cast(ws, C.c_void_p),
                                       C.cast(lds, C.c_void_p), C.cast(st_, C.c_void_p), nl, B, nc, 16, gt_d.data_ptr(),
                                       ngt_d.data_ptr(), max_gt, 7.5, 0.5, 1.5, 1.0, out.data_ptr(), wsb.data_ptr(), nbytes,
                                       L.current_stream(DEV)), "detection_loss")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), items.numpy(), rtol=2e-4)
    for g, f in zip(gd, fr):
        got, ref = to_cpu_nchw(g), f.grad
        assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_training_step_yolov8n_matches_reference_golden(dtype, golden_dir):
    """Two full training steps of yolov8n (train-mode forward, loss, backward, clip, SGD nesterov, EMA) on the HIP path vs
    what the imported reference recorded (tests/golden/train_yolov8n.npz). f32: parity mode; bf16: AMP-like bound."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    G = np.load(golden_dir / "train_yolov8n.npz")
    bs, imgsz, steps = int(G["bs"][0]), int(G["imgsz"][0]), int(G["steps"][0])
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=dtype, device=DEV)
    keys = [str(k) for k in G["param_keys"]]
    named = dict(m.named_parameters())
    f32 = dtype == torch.float32
    for step in range(steps):
        x = P.synthetic_images(bs, h=imgsz, w=imgsz, seed=step).to(DEV)
        lab = P.synthetic_labels(bs, seed=step)
        items = tr.forward_backward(x, lab)
        norm = tr.grad_norm()
        torch.cuda.synchronize()
        ref_items, ref_norm = G[f"loss_items_{step}"], float(G[f"grad_norm_{step}"][0])
        # bf16: a statistical bound (see the note at the end), per step.  Step 0 starts from the reference's weights: measured on
        # MI355X (tools/experiments/train_bf16_err.py, round 3) loss items 1.4-4.7 %, gradient norm 8 % off the f32 reference.  Step 1
        # runs on weights that already differ by one bf16-gradient update and train-mode BatchNorm renormalises every layer: 9-14.5 % /
        # 14.7 % (three kernel configurations that differ only in the f32 summation order inside the convs landed 10, 13 and 14.5 %).
        # The step-1 bound is the step-0 bound (8 % / 12 %) compounded once: step 1 starts from weights that are already one
        # bf16-gradient update (up to 12 % off in norm) away from the reference's, so (1.12 x 1.08 - 1) = 21 % is what the same
        # per-step error allows; 20 % sits on it.  A regression of the kernels shows in step 0, whose bound did not move.
        rt_items, rt_norm = (2e-3, 3e-3) if f32 else ((0.08, 0.12) if step == 0 else (0.2, 0.2))
        np.testing.assert_allclose(items.cpu().numpy(), ref_items, rtol=rt_items)
        assert abs(norm - ref_norm) <= rt_norm * ref_norm
        # per-parameter gradient norms (the golden stores them after clipping)
        coef = min(1.0, 10.0 / (ref_norm + 1e-6))
        l2 = np.array([float(named[k].grad.double().norm()) * coef for k in keys])
        ref_l2 = G[f"grad_l2_{step}"]
        big = ref_l2 > 1e-3 * ref_l2.max()
        if f32:
            np.testing.assert_allclose(l2[big], ref_l2[big], rtol=2e-2)
        else:
            assert float(np.median(np.abs(l2[big] - ref_l2[big]) / ref_l2[big])) <= 0.15
        if f32:
            np.testing.assert_allclose(named["model.22.cv3.0.2.bias"].grad.cpu().numpy() * coef, G[f"grad_cls_bias_{step}"],
                                       rtol=5e-3, atol=1e-5 * float(np.abs(G[f"grad_cls_bias_{step}"]).max()))
        tr.optimizer_step()
        torch.cuda.synchronize()
        sd = m.state_dict()
        np.testing.assert_allclose(sd["model.0.conv.weight"].cpu().numpy(), G[f"w_stem_{step}"], rtol=0, atol=2e-5 if f32 else 1.5e-2)
        np.testing.assert_allclose(sd["model.2.cv1.bn.running_var"].cpu().numpy(), G[f"bn_rv_{step}"], rtol=2e-3 if f32 else 0.1)
        fk = [str(k) for k in G["state_keys"]]
        ema = tr.ema_state_dict()
        got = np.array([float(ema[k].double().sum()) for k in fk])
        if f32:  # sums of signed weights cancel: only meaningful at f32 accuracy
            np.testing.assert_allclose(got, G[f"ema_sum_{step}"], rtol=1e-3, atol=1e-2)
    # bf16 bounds are statistical: 8 mantissa bits through ~60 layers with random weights move individual
    # TaskAlignedAssigner decisions, so single gradients differ by tens of percent while losses / norms stay within ~10 %


@pytest.mark.parametrize("world", [1, 2], ids=["one_graph", "two_graphs_around_the_allreduce"])
def test_compiled_training_step_equals_eager(world):
    """DetectionTrainer.compile(): the hipGraph replay of the step (static buffers, device-side EMA decay, labels uploaded
    outside the graph) follows the eager step: same loss items per step and the same weights / EMA after four steps.
    world = 2 exercises the multi-GPU structure (forward+backward graph | gradient all-reduce | optimizer graph) in one
    process: without an initialised process group the all-reduce helper is the identity."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 4, 192
    runs = []
    for compiled in (False, True):
        m = DetectionModel("yolov8n.yaml")
        P.apply_procedural_weights(m)
        tr = DetectionTrainer(m, dtype=torch.bfloat16, device=DEV, world_size=world)
        batches = [(P.synthetic_images(bs, h=sz, w=sz, seed=s).to(DEV), P.synthetic_labels(bs, seed=s)) for s in range(4)]
        items = []
        if compiled:
            # two eager steps, then capture without further warm-up steps
            items.append(tr.step(*batches[0]).cpu().clone())
            items.append(tr.step(*batches[1]).cpu().clone())
            tr.compile(*batches[1], warm_steps=0)
            assert len(tr._graphs) == world
            for b in batches[2:]:
                items.append(tr.step(*b).cpu().clone())
        else:
            for b in batches:
                items.append(tr.step(*b).cpu().clone())
        torch.cuda.synchronize()
        runs.append((items, m.state_dict()["model.0.conv.weight"].cpu().clone(), tr.E.cpu().clone(), tr.updates))
    (ia, wa, ea, ua), (ib, wb, eb, ub) = runs
    assert ua == ub == 4
    for a, b in zip(ia, ib):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5)
    np.testing.assert_allclose(wa.numpy(), wb.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ea.numpy(), eb.numpy(), rtol=1e-5, atol=1e-7)


def test_training_step_with_statistics_epilogues_follows_reduction_passes():
    """A yolov8s training step (bf16, 4 x 256 x 256) with the BatchNorm batch statistics taken from the convolutions' own workgroups
    (default) against the same step with a reduction pass over every z (`upa_opts.no_epi_stats = 1`).  Per layer z is bit-identical
    for identical inputs and the statistics differ in f32 summation order only (2e-6, `test_conv_batch_statistics_from_the_convolution_
    epilogue`): the first layer whose kernel has the epilogue (model.2.cv1, same input in both runs) updates its running statistics to
    1e-5; further down a 1e-6 shift of a mean flips single bf16 roundings, which this weight family amplifies (as it does between any
    two summation orders: the bf16 gates against the golden are 0.2) - loss items and the gradient norm agree to a few percent."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 4, 256
    batch = (P.synthetic_images(bs, h=sz, w=sz, seed=0).to(DEV), P.synthetic_labels(bs, seed=0))
    runs = []
    for off in (1, 0):
        m = DetectionModel("yolov8s.yaml")
        P.apply_procedural_weights(m)
        tr = DetectionTrainer(m, dtype=torch.bfloat16, device=DEV)
        with R.use_opts(L.Opts(no_epi_stats=off)):
            items = tr.forward_backward(*batch).cpu().clone()
            tr._join_wgrad()
            gn = tr.grad_norm()
        torch.cuda.synchronize()
        sd = m.state_dict()
        runs.append((items, gn, sd["model.2.cv1.bn.running_var"].cpu().clone(), sd["model.2.cv1.bn.running_mean"].cpu().clone()))
    (ia, ga, va, ma), (ib, gb, vb, mb) = runs
    np.testing.assert_allclose(va.numpy(), vb.numpy(), rtol=1e-5)
    np.testing.assert_allclose(ma.numpy(), mb.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ia.numpy(), ib.numpy(), rtol=5e-2)
    assert abs(ga - gb) <= 0.1 * gb


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_amp_grad_scaler_follows_torch_semantics(dtype):
    """DetectionTrainer(amp_scaler=True) = the reference's `torch.amp.GradScaler` loop (engine/trainer.py:301-302, 429, 676-679;
    torch/amp/grad_scaler.py) with the scaler state on the device:
      * a scaled step lands where the unscaled one does (the scale is a power of two: loss gradients x 2**16, divided out in the
        optimizer - exact in f32 / bf16 short of overflow) and the reported gradient norm is the unscaled one;
      * `growth_interval` clean steps in a row double the scale;
      * a step whose gradients hold an inf is skipped - parameters and momentum untouched, gradients zeroed, EMA still updated
        (optimizer_step calls ema.update regardless) - the scale halves and the growth tracker restarts;
      * state_dict / load_state_dict carry torch's keys."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 2, 128
    batches = [(P.synthetic_images(bs, h=sz, w=sz, seed=s).to(DEV), P.synthetic_labels(bs, seed=s)) for s in range(3)]

    def make(scaled):
        m = DetectionModel("yolov8n.yaml")
        P.apply_procedural_weights(m)
        tr = DetectionTrainer(m, dtype=dtype, device=DEV, amp_scaler=scaled)
        if scaled:
            tr.scaler.growth_interval = 2
        return m, tr

    (m0, t0), (m1, t1) = make(False), make(True)
    assert t1.scaler.get_scale() == 65536.0 and t0.scaler.ptr() is None
    for k, b in enumerate(batches[:2]):
        i0, i1 = t0.step(*b).cpu(), t1.step(*b).cpu()
        torch.cuda.synchronize()
        np.testing.assert_allclose(i1.numpy(), i0.numpy(), rtol=1e-6)  # loss items are unscaled
        rtol = 1e-5 if dtype == torch.float32 else 2e-3  # bf16: 2**16 x is exact, but sums of scaled bf16 products round elsewhere
        np.testing.assert_allclose(t1.P.cpu().numpy(), t0.P.cpu().numpy(), rtol=rtol, atol=1e-6)
        np.testing.assert_allclose(t1.E.cpu().numpy(), t0.E.cpu().numpy(), rtol=rtol, atol=1e-6)
        assert not t1.scaler.found_inf()
        assert t1.scaler.get_scale() == (65536.0 if k == 0 else 131072.0)  # two clean steps: x growth_factor
    sd = t1.scaler.state_dict()
    assert sd == {"scale": 131072.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2, "_growth_tracker": 0}
    # an overflowing step: forward + backward, one gradient made inf, then the reference's optimizer_step
    t1.forward_backward(*batches[2])
    t1._join_wgrad() if hasattr(t1, "_join_wgrad") else None
    torch.cuda.synchronize()
    t1.G[12345 % t1.G.numel()] = float("inf")
    p_before, m_before, e_before = t1.P.clone(), t1.M.clone(), t1.E.clone()
    t1.optimizer_step()
    torch.cuda.synchronize()
    assert torch.equal(t1.P, p_before) and torch.equal(t1.M, m_before)
    assert float(t1.G.abs().max()) == 0.0
    assert not torch.equal(t1.E, e_before)  # EMA of the (unchanged) weights still moves towards them
    assert t1.scaler.found_inf() and t1.scaler.get_scale() == 65536.0 and t1.scaler.state_dict()["_growth_tracker"] == 0
    # the next clean step trains again with the smaller scale
    t1.step(*batches[2])
    torch.cuda.synchronize()
    assert not torch.equal(t1.P, p_before) and not t1.scaler.found_inf() and t1.scaler.state_dict()["_growth_tracker"] == 1
    t2 = make(True)[1]
    t2.scaler.load_state_dict(sd)
    assert t2.scaler.state_dict() == sd
    # the reported gradient norm is that of the unscaled gradients
    t0.forward_backward(*batches[0]); t1.forward_backward(*batches[0])
    for t in (t0, t1):
        if hasattr(t, "_join_wgrad"):
            t._join_wgrad()
    torch.cuda.synchronize()
    # (t0 and t1 differ by the skipped step; compare t1's norm with its own scaled buffer instead)
    assert abs(t1.grad_norm() - float(t1.G.double().norm()) / t1.scaler.get_scale()) <= 1e-3 * t1.grad_norm()


def test_amp_grad_scaler_inside_the_compiled_step():
    """The scaler's state lives on the device, so a scaled step is still ONE hipGraph (the reference's `scaler.step` reads `found_inf`
    on the host): the replayed steps land where the eager scaled steps do, and the scale doubles after `growth_interval` replays."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 2, 128
    batches = [(P.synthetic_images(bs, h=sz, w=sz, seed=s).to(DEV), P.synthetic_labels(bs, seed=s)) for s in range(4)]
    runs = []
    for compiled in (False, True):
        m = DetectionModel("yolov8n.yaml")
        P.apply_procedural_weights(m)
        tr = DetectionTrainer(m, dtype=torch.bfloat16, device=DEV, amp_scaler=True)
        tr.scaler.growth_interval = 3
        tr.step(*batches[0])
        if compiled:
            tr.compile(*batches[0], warm_steps=0)
            assert len(tr._graphs) == 1
        for b in batches[1:]:
            tr.step(*b)
        torch.cuda.synchronize()
        runs.append((tr.P.cpu().clone(), tr.scaler.get_scale(), tr.scaler.state_dict()["_growth_tracker"]))
    (pa, sa, ta), (pb, sb, tb) = runs
    assert sa == sb == 131072.0 and ta == tb == 1  # four clean steps at interval 3: one doubling, tracker at 1
    np.testing.assert_allclose(pa.numpy(), pb.numpy(), rtol=1e-5, atol=1e-7)


def test_training_step_yolov8s_f32_matches_reference_golden(golden_dir):
    """BASELINE config 3's model (yolov8s), one f32 training step vs the imported reference's record
    (tests/golden/train_yolov8s.npz: bs 2, 256 x 256): loss items, gradient norm, per-parameter gradient norms, updated
    stem weights and BN running variance."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    G = np.load(golden_dir / "train_yolov8s.npz")
    bs, imgsz = int(G["bs"][0]), int(G["imgsz"][0])
    m = DetectionModel("yolov8s.yaml")
    P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=torch.float32, device=DEV)
    x = P.synthetic_images(bs, h=imgsz, w=imgsz, seed=0).to(DEV)
    items = tr.forward_backward(x, P.synthetic_labels(bs, seed=0))
    norm = tr.grad_norm()
    ref_norm = float(G["grad_norm_0"][0])
    np.testing.assert_allclose(items.cpu().numpy(), G["loss_items_0"], rtol=2e-3)
    assert abs(norm - ref_norm) <= 3e-3 * ref_norm
    keys = [str(k) for k in G["param_keys"]]
    named = dict(m.named_parameters())
    coef = min(1.0, 10.0 / (ref_norm + 1e-6))
    l2 = np.array([float(named[k].grad.double().norm()) * coef for k in keys])
    ref_l2 = G["grad_l2_0"]
    big = ref_l2 > 1e-3 * ref_l2.max()
    np.testing.assert_allclose(l2[big], ref_l2[big], rtol=2e-2)
    tr.optimizer_step()
    torch.cuda.synchronize()
    sd = m.state_dict()
    np.testing.assert_allclose(sd["model.0.conv.weight"].cpu().numpy(), G["w_stem_0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sd["model.2.cv1.bn.running_var"].cpu().numpy(), G["bn_rv_0"], rtol=2e-3)


def test_training_step_full_size_is_deterministic_and_descends():
    """BASELINE config 3 at its own size (yolov8s, batch 32 per GPU, 640 x 640, bf16) through size-independent
    properties: two trainers from the same initial weights take bit-identical steps (every reduction - BN statistics,
    weight gradients, loss sums, gradient norm - runs in a fixed order, no atomics); loss items and the clipped gradient
    norm are finite and positive; the EMA moved towards the weights; and on a repeated batch the loss goes down."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    x = P.synthetic_images(32, h=640, w=640).to(DEV)
    labels = P.synthetic_labels(32)
    runs = []
    for _ in range(2):
        m = DetectionModel("yolov8s.yaml")
        P.apply_procedural_weights(m)
        tr = DetectionTrainer(m, dtype=torch.bfloat16, device=DEV)
        e0 = tr.E.clone()
        items = [tr.step(x, labels).cpu().clone() for _ in range(3)]
        tr.forward_backward(x, labels)  # the optimizer step zeroes the gradients: a fourth backward for their norm
        torch.cuda.synchronize()
        runs.append((items, tr.P.cpu().clone(), tr.E.cpu().clone(), float(tr.grad_norm()), (tr.E - e0).abs().max().item()))
    (ia, pa, ea, ga, da), (ib, pb, eb, gb, _) = runs
    for a, b in zip(ia, ib):
        assert torch.equal(a, b)
    assert torch.equal(pa, pb) and torch.equal(ea, eb) and ga == gb
    assert all(torch.isfinite(t).all() and (t > 0).all() for t in ia)
    assert np.isfinite(ga) and ga > 0 and da > 0
    assert ia[-1].sum().item() < ia[0].sum().item(), "three SGD steps on one batch must lower the loss"


def test_batched_weight_repack_equals_per_layer_and_host_pack():
    """upa_pack_conv_weights_batched (one launch for every conv of the model, csrc/train.hip) writes the same bytes as
    upa_pack_conv_weight_dev per layer and - forward layout - as the host packer used for inference
    (upa_pack_conv_weight); covers 3x3 / 1x1, Cin not a multiple of the k-tile, transposed + flipped data-gradient
    layouts and both dtypes."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd import _lib as L
    lib = L.lib()
    st = L.current_stream(DEV)
    cases = [(64, 48, 3, 0), (64, 48, 3, 1), (80, 64, 1, 0), (16, 8, 3, 1), (128, 96, 1, 1), (32, 32, 2, 0)]
    desc = np.dtype([("w", "<u8"), ("out", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("k", "<i4"), ("dtype", "<i4"),
                     ("tf", "<i4"), ("reserved", "<i4")])
    for code, tdt in ((1, torch.bfloat16), (0, torch.float32)):
        ws, outs_b, outs_s, table = [], [], [], np.zeros(len(cases), dtype=desc)
        for i, (co, ci, k, tf) in enumerate(cases):
            w = P.uniform(f"pk{i}", (co, ci, k, k), -1, 1).to(DEV).contiguous()
            lco, lci = (ci, co) if tf else (co, ci)
            nb = lib.upa_conv_packed_weight_bytes(lco, lci, k, code)
            ob = torch.zeros(nb, dtype=torch.uint8, device=DEV)
            os_ = torch.zeros(nb, dtype=torch.uint8, device=DEV)
            L.check(lib.upa_pack_conv_weight_dev(w.data_ptr(), co, ci, k, code, tf, os_.data_ptr(), st))
            table[i] = (w.data_ptr(), ob.data_ptr(), co, ci, k, code, tf, 0)
            ws.append(w); outs_b.append(ob); outs_s.append(os_)
            if not tf:
                host = torch.zeros(nb, dtype=torch.uint8)
                wc = w.cpu().contiguous()
                L.check(lib.upa_pack_conv_weight(wc.data_ptr(), co, ci, k, code, host.data_ptr()))
                torch.cuda.synchronize()
                assert torch.equal(os_.cpu(), host)
        tdev = torch.from_numpy(table.view(np.uint8).copy()).to(DEV)
        L.check(lib.upa_pack_conv_weights_batched(tdev.data_ptr(), len(cases), st))
        torch.cuda.synchronize()
        for ob, os_ in zip(outs_b, outs_s):
            assert torch.equal(ob, os_)


def test_warmup_and_accumulation_schedule_matches_oracle():
    """engine/trainer.py:337-338, 392-413, 428-431 of the reference: per-iteration learning rates (bias group falling from
    warmup_bias_lr, the others rising from 0), momentum rising from warmup_momentum, `accumulate` growing from 1 to
    nbs / batch, the optimizer stepping only every `accumulate` iterations with the gradients of the iterations in between
    summed, weight decay scaled by batch * accumulate / nbs.  Six iterations of yolov8n (bs 8 at 160 px -> accumulate
    reaches 2 inside a 100-iteration warm-up when nbs = 64... shortened here with nbs = 16, nb = 2, warmup_epochs = 2)
    on the HIP trainer (f32) vs the oracle loop built from oracle.train.schedule / train_step."""
    from oracle import tasks as ot
    from oracle import train as otr
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz, iters = 8, 160, 6
    # nw = max(round(2 * 2), 100) = 100; small learning rates keep the two f32 trajectories of this random-weight model
    # comparable over six iterations (with warmup_bias_lr 0.1 the class loss runs into the thousands within three steps
    # and amplifies last-bit differences to percents)
    sched = dict(otr.SCHED, nbs=16, warmup_epochs=2.0, warmup_bias_lr=0.004)
    hyp = dict(otr.HYP, lr=0.002)
    # known answers of the schedule itself
    assert otr.schedule(0, 2, bs, sched=dict(sched, warmup_bias_lr=0.1))[:3] == (1, [0.1, 0.0, 0.0], 0.8)
    a, lrs, mom, wd = otr.schedule(100, 2, bs, sched=sched)
    assert a == 2 and abs(mom - 0.9) < 1e-12 and abs(wd - 5e-4 * bs * 2 / 16) < 1e-12
    x = P.synthetic_images(bs, h=sz, w=sz)
    lab = P.synthetic_labels(bs)
    ref = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(ref)
    st = otr.TrainState(ref)
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=torch.float32, device=DEV, hyp=dict(lr=hyp["lr"]))
    tr.set_schedule(2, **{k: sched[k] for k in ("nbs", "warmup_epochs", "warmup_bias_lr")})
    start = 48  # accumulate is 1 up to ni = 49 and 2 from ni = 50 on: optimizer steps at ni = 48, 49, 51, 53
    tr.ni, tr.last_opt_step = start, start - 1
    last = start - 1
    batch = {"img": x, **lab}
    for it in range(iters):
        ni = start + it
        acc, lrs, mom, wd = otr.schedule(ni, 2, bs, hyp=hyp, sched=sched)
        do = ni - last >= acc
        items_ref, _ = otr.train_step(ref, st, batch, hyp=hyp, lrs=lrs, momentum=mom, weight_decay=wd, optimize=do,
                                      zero_grad=(ni - 1 == last))
        if do:
            last = ni
        items = tr.step(x.to(DEV), lab)
        torch.cuda.synchronize()
        np.testing.assert_allclose(items.cpu().numpy(), items_ref.detach().numpy(), rtol=5e-3)
    assert tr.last_opt_step == last and tr.updates == st.updates and st.updates >= 3
    named = dict(m.named_parameters())
    for k, p in ref.named_parameters():
        if p.requires_grad:
            d = float((named[k].detach().cpu() - p.detach()).abs().max())
            assert d <= 5e-5 * max(1.0, float(p.detach().abs().max())), (k, d)


def test_eval_after_training_step_sees_the_new_weights():
    """The optimizer writes parameters and BN buffers through raw pointers (`upa_sgd_nesterov_ema` on the trainer's flat buffers), so no
    tensor `_version` moves; the inference path's packed-weight caches (nn/modules/conv.py `version_key`, bumped by
    `bump_weights_generation`, engine/trainer.py optimizer_step) must still repack.  eval -> `trainer.step` -> eval on ONE model object
    against the oracle doing the same: after the step the HIP eval output must follow the oracle's (new weights, new running
    statistics) and differ from its own pre-step output - a stale cache would reproduce the pre-step output exactly."""
    from oracle import tasks as ot
    from oracle import train as otr
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 4, 256
    x = P.synthetic_images(bs, h=sz, w=sz)
    lab = P.synthetic_labels(bs)
    o = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(o)
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m = m.to(DEV).eval()
    m.set_compute_dtype(torch.float32)

    def eval_pair():
        o.eval()
        m.eval()
        with torch.no_grad():
            yo = o(x)[0]
            ym = m(x.to(DEV))[0]
        torch.cuda.synchronize()
        return yo, ym.cpu()

    yo0, ym0 = eval_pair()
    d0 = (ym0 - yo0).abs()
    assert d0[:, :4].max().item() <= 2e-3 and d0[:, 4:].max().item() <= 1e-3  # unfused oracle eval vs BN-folded HIP eval
    tr = DetectionTrainer(m, dtype=torch.float32, device=DEV)
    st = otr.TrainState(o)
    for _ in range(2):
        tr.step(x.to(DEV), lab)
        otr.train_step(o, st, {"img": x, **lab})
    torch.cuda.synchronize()
    yo1, ym1 = eval_pair()
    moved = (yo1 - yo0).abs()
    d1 = (ym1 - yo1).abs()
    print(f"eval/step/eval: the oracle's output moved by {moved[:, :4].max():.3f} px / {moved[:, 4:].max():.4f}; HIP vs oracle after the "
          f"steps: {d1[:, :4].max():.3e} px / {d1[:, 4:].max():.3e} (before: {d0[:, :4].max():.3e} / {d0[:, 4:].max():.3e})")
    assert moved[:, 4:].max().item() > 20 * 1e-3, "the two steps were expected to move the scores visibly"
    assert not torch.equal(ym1, ym0)
    # two f32 training steps reproduce the reference's weights to ~1e-5 (test_training_step_yolov8n_matches_reference_golden);
    # through the eval forward that is a few 1e-3 on 256-px boxes
    assert d1[:, :4].max().item() <= 0.1 * moved[:, :4].max().item() + 5e-3
    assert d1[:, 4:].max().item() <= 0.1 * moved[:, 4:].max().item() + 2e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_gradients_do_not_depend_on_the_stream_layout(dtype):
    """The backward walk may run the Detect head's 40 x 40 / 20 x 20 levels on a second stream (`DetectT.fork_levels`) and the weight
    gradients on 0, 1 or 2 side streams (`DetectionTrainer(wgrad_streams=...)`).  None of that changes the arithmetic, so the flat
    gradient buffer must come out BIT-IDENTICAL for all six layouts.  With `wgrad_streams = 0` the weight gradients of the forked
    levels are launched on the level stream itself - beside the main stream's - and need their own split-K partial-sum workspace
    (round-5 advisor finding: they shared one buffer and corrupted each other silently)."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine import trainer as TR
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    bs, sz = 4, 256
    batch = (P.synthetic_images(bs, h=sz, w=sz, seed=3).to(DEV), P.synthetic_labels(bs, seed=3))
    grads = {}
    saved = TR.DetectT.fork_levels
    try:
        for fork in (True, False):
            for ns in (0, 1, 2):
                TR.DetectT.fork_levels = fork
                m = DetectionModel("yolov8s.yaml")
                P.apply_procedural_weights(m)
                tr = TR.DetectionTrainer(m, dtype=dtype, device=DEV, wgrad_streams=ns)
                for _ in range(2):  # the second pass runs with every workspace already allocated and the streams warm
                    items = tr.forward_backward(*batch)
                    tr._join_wgrad()
                torch.cuda.synchronize()
                grads[(fork, ns)] = (tr.G.cpu().clone(), items.cpu().clone())
                del tr, m
    finally:
        TR.DetectT.fork_levels = saved
    g0, i0 = grads[(False, 0)]
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    for key, (g, it) in grads.items():
        assert torch.equal(it, i0), key
        assert torch.equal(g, g0), (key, float((g - g0).abs().max()))
