"""bench.py's measurement legs outside the timed region: the live kernel profile behind `roofline` / `critical_path` (HIP events on the
launch stream + the committed PMC summaries), the whole-step roofline, the per-layer weight-gradient profile of `--workload train`."""
from __future__ import annotations

import json

import torch

from .common import GFLOP_OTHER, GFLOP_PER_IMG, PEAK_BF16_TFLOPS, PEAK_F32_TFLOPS, PEAK_HBM_GBS, ROOT


def wgrad_profile(tr, L, R, dev, dtype, reps=5, cfg_key=None):
    """Dominant kernel of the training step = the weight-gradient MFMA kernel: every layer's launch re-issued `reps` times
    back to back on the current stream between HIP events (its operands are still resident from the last step)."""
    fam = {}
    lib = L.lib()
    st = L.current_stream(dev)
    scratch = {}
    for cv in tr.convs:
        if cv.x is None:
            continue
        vx = R.view_of(cv.x)
        oh, ow = (vx.h + 2 * cv.p - cv.k) // cv.s + 1, (vx.w + 2 * cv.p - cv.k) // cv.s + 1
        dz = scratch.setdefault((vx.n, cv.cout, oh, ow), torch.zeros(vx.n, oh, ow, cv.cout, dtype=dtype, device=dev))
        dw = torch.zeros(cv.cout, cv.cin, cv.k, cv.k, device=dev)
        ws = tr.ctx.wgrad_ws

        def call():
            L.check(lib.upa_conv2d_wgrad(vx.ptr, vx.n, vx.h, vx.w, cv.cin, vx.ld, dz.data_ptr(), cv.cout, cv.cout, dw.data_ptr(),
                                         cv.k, cv.s, cv.p, 1, vx.dtype, ws.data_ptr(), ws.numel(), st), "wgrad")
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        bf = dtype == torch.bfloat16
        # the dispatch of upa_conv2d_wgrad (csrc/train.hip): bf16 MFMA kernels where the channel counts allow
        if bf and cv.k == 3 and cv.cout >= 16 and (cv.cin % 8 == 0 or cv.cin < 8):
            # LDS-DMA ring kernels; 9 - 16 input channels keep the register-staged narrow form
            fam_ = "wgrad_bf16_k3_kernel" if 8 < cv.cin <= 16 else "wgrad_k3_ring_kernel"
            name, peak = "void (anonymous namespace)::%s<%d, %d>((anonymous namespace)::WgradParams)" % (
                fam_, cv.s, 16 if cv.cin <= 16 else 64), PEAK_BF16_TFLOPS
        elif bf and cv.k == 1 and cv.s == 1 and cv.p == 0 and cv.cin >= 32 and cv.cout >= 32 and cv.cin % 8 == 0:
            name, peak = "(anonymous namespace)::wgrad_k1_ring_kernel((anonymous namespace)::WgradParams)", PEAK_BF16_TFLOPS
        else:
            small = cv.cin <= 32 or cv.cout <= 32
            mt = 4 if (cv.k == 1 and cv.cin >= 128 and cv.cout >= 128) else (1 if small else 2)
            name = "void (anonymous namespace)::wgrad_kernel<%s, %d, %d, %d>((anonymous namespace)::WgradParams)" % (
                "unsigned short" if bf else "float", mt, mt, cv.k)
            peak = PEAK_F32_TFLOPS
        d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, peak=peak))
        d["launches"] += 1
        d["ms"] += ms
        d["flops"] += 2.0 * vx.n * oh * ow * cv.cout * cv.cin * cv.k * cv.k
        # algorithmic bytes: the layer input and the output gradient read once (activation dtype), the f32 weight gradient written once
        d["bytes"] += vx.n * (vx.h * vx.w * cv.cin + oh * ow * cv.cout) * (2 if bf else 4) + 4.0 * cv.cout * cv.cin * cv.k * cv.k
    name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    avg_s = d["ms"] / d["launches"] * 1e-3
    tf = d["flops"] / d["launches"] / avg_s / 1e12
    # HBM bytes per launch of the dominant family from the committed PMC passes (tools/pmc_wgrad.sh: FETCH_SIZE x 2 + WRITE_SIZE,
    # separate passes); only valid for the configuration they were collected on
    traffic, traffic_src = None, None
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_wgrad_summary.json"), reverse=True):
        try:
            pmc = json.loads(pf.read_text())
            if pmc.get("config") == cfg_key and name in pmc["kernels"]:
                traffic = round(pmc["kernels"][name]["hbm_bytes_per_launch"])
                traffic_src = f"profiles/{pf.name} (rocprofv3 --pmc, separate passes)"
                break
        except (OSError, KeyError, ValueError):
            pass
    return {"kernel": name, "bound": "mfma", "achieved": round(tf, 2), "peak": d["peak"], "unit": "TFLOP/s",
            "frac": round(tf / d["peak"], 4), "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": d["launches"],
            "avg_launch_us": round(avg_s * 1e6, 1),
            "algorithmic_flops_per_launch": d["flops"] / d["launches"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "note": "weight gradient: bf16 MFMA (v_mfma_f32_16x16x32_bf16, operands DMAed into an LDS ring and read with "
                    "ds_read_b64_tr_b16) where channel counts allow, exact-f32 MFMA otherwise; the time includes the "
                    "partial-sum reduction kernel",
            "wgrad_ms_per_step": round(sum(v["ms"] for v in fam.values()), 3),
            "families": {k: dict(launches=v["launches"], avg_us=round(v["ms"] / v["launches"] * 1e3, 1),
                                 tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in sorted(fam.items())}}


def step_roofline(img_per_s, ms_per_step, args, kernels):
    """The whole step against the chip: MFMA (algorithmic FLOPs of every conv / peak), HBM (algorithmic bytes of every conv,
    each reading its input and writing its output once, / 8 TB/s) and the ratio of the conv HBM floor to the measured step."""
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    tf = img_per_s * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG) / 1e3
    out = {"model_tflops": round(tf, 2), "frac_of_mfma_peak": round(tf / peak, 4)}
    if kernels:
        gb = kernels["conv_algorithmic_bytes"] / 1e9
        out.update({"algorithmic_GB_per_step": round(gb, 4), "algorithmic_GBs": round(gb / (ms_per_step * 1e-3), 1),
                    "frac_of_hbm_peak": round(gb / (ms_per_step * 1e-3) / PEAK_HBM_GBS, 4),
                    "conv_hbm_floor_over_step": round(kernels["conv_hbm_floor_ms"] / ms_per_step, 4)})
    return out


def kernel_profile(model, x, dtype, dev, args, pconv, L, post, reps=10):
    """Device time of every conv launch of one step, measured live with HIP events on the launch stream.

    One eager forward records the launch list; every launch is then captured `reps` times back to back into its own
    hipGraph and the replay is bracketed by events on the stream it runs on, so the figure is kernel time (no host
    launch gaps) and is comparable with rocprofv3's per-kernel AverageNs.  Launches are grouped by the exact kernel
    instantiation name rocprofv3 reports; the roofline is given for the instantiation with the largest total time."""
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.nn.modules import block as pblock
    from ultralytics_pro_amd.nn.modules import head as phead

    code = L.dtype_code(dtype)
    es = 2 if code == L.UPA_BF16 else 4
    tname = "unsigned short" if es == 2 else "float"
    calls = []  # (kernel name, flops, algorithmic bytes, replay callable)
    orig = pconv.hip_conv2d
    orig_tail = phead.Detect._tail_call

    def conv_name(n, h, w, cin, pk, stride, pad, act, residual):
        if pk.stem:
            return (f"void stem_mfma_kernel<{pk.cout // 16}, {pk.k}, {stride}, {'true' if act == 1 else 'false'}>(StemParams)" if es == 2 else
                    f"void stem_conv_kernel<{tname}, 16, {'true' if act == 1 else 'false'}>(StemParams)")
        var = L.lib().upa_conv_variant(n, h, w, cin, pk.cout, pk.k, stride, pad, code, R.opts_ptr())
        if (var >> 26) & 1:  # 8-wave two-group phased kernel for the MFMA-bound 3x3 stride-1 layers (conv_p8.hip)
            return "conv_p8_kernel(BigParams)"
        if (var >> 25) & 1:  # 4-wave 32x32x16-MFMA kernel for the MFMA-bound 3x3 layers (conv_mm.hip): <ACT, RES>
            return "void conv_mm_kernel<%d, %s>(MmParams)" % (act, "true" if residual is not None else "false")
        if (var >> 24) & 1:  # persistent weights-stationary 3x3 (conv_ws3.hip): <NT, MT>
            return "void conv_ws3_kernel<%d, %d>(BigParams)" % ((var >> 4) & 15, var & 15)
        if (var >> 23) & 1:  # large-tile LDS-shared-operand kernel (conv_big.hip): <KS, STRIDE, WM, WN, MT, NT>
            ntb, mt = (var >> 4) & 15, 4 if (var & 15) == 2 else 2
            wm, wn, nt = (8, 1, 4) if (ntb == 4 and mt == 4) else (4, 2, ntb // 2)
            if ntb == 4:
                mt = 2
            if ntb == 5:  # 80 output channels: 8 x 1 waves, 5 tiles each, MT = 2 (256 px) or 1 (128 px)
                wm, wn, nt, mt = 8, 1, 5, (2 if (var & 15) == 2 else 1)
            return "void conv_big_kernel<%d, %d, %d, %d, %d, %d, 0>(BigParams)" % (pk.k, stride, wm, wn, mt, nt)
        if (var >> 22) & 1:  # streaming pointwise kernel (conv1x1.hip): <NTW, MT, WAVES, EPI>
            return "void conv1x1_stream_kernel<%d, %d, %d, 0>(C1Params)" % (var & 15, (var >> 4) & 15, (var >> 8) & 31)
        if (var >> 21) & 1 and (var >> 8) & 1:  # 16 -> 16 channel variant of the pipelined kernel: <act, residual>
            return "void conv3x3_c16_kernel<%d, %s>(PipeParams)" % (act, "true" if residual is not None else "false")
        if (var >> 21) & 1:  # software-pipelined 3x3 (conv_pipe.hip): <NTW, act, residual>
            return "void conv3x3_pipe_kernel<%d, %d, %s>(PipeParams)" % (var & 15, act, "true" if residual is not None else "false")
        return "void %s<%s, %d, %d, %d, %d, %d>(ConvParams)" % (
            "conv_ws_kernel" if (var >> 20) & 1 else "conv_igemm_kernel", tname, (var >> 12) & 15,
            (var >> 8) & 15, (var >> 4) & 15, var & 15, (var >> 16) & 15)

    def rec(xx, pk, stride, pad, act, out=None, residual=None, out_dtype=None, key=None, up=None):
        y = orig(xx, pk, stride, pad, act, out=out, residual=residual, out_dtype=out_dtype, key=key, up=up)
        n, cin, h, w = xx.shape
        oh, ow = y.shape[2], y.shape[3]
        flops = 2.0 * n * oh * ow * pk.cout * cin * pk.k * pk.k
        nbytes = n * h * w * cin * xx.element_size() + n * oh * ow * pk.cout * es * (2 if residual is not None else 1) \
            + pk.cout * cin * pk.k * pk.k * es
        if up is not None:  # virtual Upsample + Concat: the leading channels are read at quarter size
            nbytes -= n * h * w * up.channels * xx.element_size() * 3 // 4
        calls.append((conv_name(n, h, w, cin, pk, stride, pad, act, residual), flops, nbytes,
                      lambda: orig(xx, pk, stride, pad, act, out=y, residual=residual, out_dtype=out_dtype, up=up)))
        return y

    def rec_tail(self, t, conv, raw, kind, i, plan):
        orig_tail(self, t, conv, raw, kind, i, plan)
        n, cin, h, w = t.shape
        cout = conv.out_channels
        plan_keep = dict(plan)
        # the fused 1x1 + decode launch (conv1x1.hip EPI 1 / 2): reads t once, writes 4 or nc f32 rows per anchor
        var = L.lib().upa_conv_variant(n, h, w, cin, 64 if kind == 1 else max(16, (cout + 7) // 8 * 8), 1, 1, 0, code, R.opts_ptr())
        name = "void conv1x1_stream_kernel<%d, %d, %d, %d>(C1Params)" % (var & 15, (var >> 4) & 15, (var >> 8) & 31, kind)
        flops = 2.0 * n * h * w * cout * cin
        nbytes = n * h * w * cin * 2 + n * h * w * (4 if kind == 1 else self.nc) * 4 + cout * cin * 2 + \
            (n * h * w * cout * 2 if raw is not None else 0)
        calls.append((name, flops, nbytes, lambda: orig_tail(self, t, conv, raw, kind, i, plan_keep)))

    orig_pair = L.lib().upa_bottleneck_pair
    orig_c2f = L.lib().upa_c2f_fused
    orig_btail = L.lib().upa_detect_branch_tail
    orig_paircv2 = L.lib().upa_bottleneck_pair_cv2
    orig_c2f64 = L.lib().upa_c2f64_fused
    orig_c2f32up = L.lib().upa_c2f32_up_fused
    orig_c2f16down = L.lib().upa_c2f16_down_fused
    orig_dstream = L.lib().upa_detect_level_stream
    orig_sppf = L.lib().upa_sppf_front
    c2f32up_calls, c2f16down_calls, dstream_calls, sppf_calls = [], [], [], []
    pair_calls, c2f_calls, btail_calls, paircv2_calls, c2f64_calls = [], [], [], [], []

    class _LibProxy:
        """Forwards every C entry to the real library, recording the fused-block launches (Bottleneck / C2f / Detect call
        them directly, not through hip_conv2d)."""

        def __getattr__(self, name):
            return getattr(real_lib, name)

        def upa_bottleneck_pair(self, *a):
            rc = orig_pair(*a)
            if rc == 0:
                pair_calls.append(a)
            return rc

        def upa_bottleneck_pair_cv2(self, *a):
            rc = orig_paircv2(*a)
            if rc == 0:
                paircv2_calls.append(a)
            return rc

        def upa_c2f_fused(self, *a):
            rc = orig_c2f(*a)
            if rc == 0:
                c2f_calls.append(a)
            return rc

        def upa_sppf_front(self, *a):
            rc = orig_sppf(*a)
            if rc == 0:
                sppf_calls.append(a)
            return rc

        def upa_detect_level_stream(self, *a):
            rc = orig_dstream(*a)
            if rc == 0:  # the two upa_detect_branch structs are passed by reference and die with the caller: keep copies for the replay
                import ctypes as C_
                bx, cl = type(a[6]._obj)(), type(a[7]._obj)()
                C_.memmove(C_.byref(bx), C_.byref(a[6]._obj), C_.sizeof(bx))
                C_.memmove(C_.byref(cl), C_.byref(a[7]._obj), C_.sizeof(cl))
                dstream_calls.append((a, bx, cl))
            return rc

        def upa_c2f16_down_fused(self, *a):
            rc = orig_c2f16down(*a)
            if rc == 0:
                c2f16down_calls.append(a)
            return rc

        def upa_c2f64_fused(self, *a):
            rc = orig_c2f64(*a)
            if rc == 0:
                c2f64_calls.append(a)
            return rc

        def upa_c2f32_up_fused(self, *a):
            rc = orig_c2f32up(*a)
            if rc == 0:
                c2f32up_calls.append(a)
            return rc

        def upa_detect_branch_tail(self, *a):
            rc = orig_btail(*a)
            if rc == 0:
                btail_calls.append(a)
            return rc

    real_lib = L.lib()
    proxy = _LibProxy()
    mods = (pconv, pblock, phead)
    pool = R.BufferPool()
    orig_libfn = L.lib
    try:
        for m in mods:
            m.hip_conv2d = rec
        phead.Detect._tail_call = rec_tail
        L.lib = lambda: proxy
        with torch.no_grad(), R.static_buffers(pool):
            post(model._predict_once(x))
    finally:
        for m in mods:
            m.hip_conv2d = orig
        phead.Detect._tail_call = orig_tail
        L.lib = orig_libfn
    for a in pair_calls:  # (x, n, h, w, c, ldx, w1, b1, w2, b2, y, ldy, residual, act, dtype, opts, stream)
        n_, h_, w_, c_ = a[1], a[2], a[3], a[4]
        flops = 2 * 2.0 * n_ * h_ * w_ * c_ * c_ * 9
        nbytes = 2 * (n_ * h_ * w_ * c_ * 2 * (2 + (0.5 if a[12] else 0)) + c_ * c_ * 9 * 2)  # two convs, each in + out (+ residual)
        calls.append(("void conv_pair_kernel<%d, %s, false>(PairParams)" % (c_ // 32, "true" if a[12] else "false"), flops, nbytes,
                      (lambda a=a: orig_pair(*a[:16], L.current_stream(dev)))))
    for a in paircv2_calls:  # (x, y0, n, h, w, ldx, w1, b1, w2, b2, residual, wc_std, wc_b, bc, out, ldout, act, dtype, opts, stream)
        npx = a[2] * a[3] * a[4]
        flops = 2.0 * npx * (2 * 9 * 32 * 32 + 96 * 64)
        nbytes = npx * (64 + 64) * 2 + (18 * 32 * 32 + 96 * 64) * 2  # y0 | y1 in, 64 channels out, weights
        calls.append(("void conv_pair_kernel<1, %s, true>(PairParams)" % ("true" if a[10] else "false"), flops, nbytes,
                      (lambda a=a: orig_paircv2(*a[:19], L.current_stream(dev)))))
    for a in c2f_calls:  # (x, n, h, w, c1, ldx, c, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, c_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[6], a[7], a[16]
        wts = c1_ * 2 * c_ + nb_ * 18 * c_ * c_ + (2 + nb_) * c_ * c2_
        flops = 2.0 * npx * wts
        nbytes = npx * (c1_ + c2_) * 2 + wts * 2  # block input + block output + weights
        o_ = R.current_opts()
        th = 10 if (c_ != 16 and nb_ == 2 and o_ is not None and o_.c2f32_th == 10) else 16
        stream_form = c_ == 32 and th == 16 and (o_ is None or o_.c2f_stream != 1)  # the line-buffer kernels (csrc/c2f_stream.hip)
        tile16 = o_ is not None and (o_.c2f16_waves in (4, 8) or o_.c2f_stream == 1)   # else the line-buffer form (csrc/c2f16_stream.hip)
        name = (("void c2f16_fused_kernel<%d>(C2fParams)" % (8 if o_.c2f16_waves == 8 else 4) if tile16 else "c2f16_stream_kernel(C16Params)") if c_ == 16 else
                ("c2f32_stream2_kernel(C2fsParams)" if nb_ == 2 and (o_ is None or o_.c2f_stream != 2) else
                 "void c2f32_stream_kernel<2>(C2fsParams)" if nb_ == 2 else "void c2f32_stream1_kernel<1>(C2fsParams)") if stream_form else
                "void c2f32_fused_kernel<%d, %d>(C2f32Params)" % (nb_, th))
        calls.append((name, flops, nbytes, (lambda a=a: orig_c2f(*a[:21], L.current_stream(dev)))))
    for a in c2f16down_calls:  # (x, n, h, w, ldx, w1, b1, wm, bm, w2, b2, wd, bd, y, ldy, dtype, opts, stream): yolov8n rows 2-3 as one launch
        npx = a[1] * a[2] * a[3]
        wts = 32 * 32 + 18 * 16 * 16 + 48 * 32      # the block's weights per pixel of its map; the stride-2 conv: 9 x 32 x 64 per OUTPUT pixel
        flops = 2.0 * npx * wts + 2.0 * (npx // 4) * 9 * 32 * 64
        nbytes = npx * 32 * 2 + (npx // 4) * 64 * 2 + (wts + 9 * 32 * 64) * 2   # block input + stride-2 output + weights (the block's output stays in LDS)
        calls.append(("c2f16_down_kernel(C16Params)", flops, nbytes, (lambda a=a: orig_c2f16down(*a[:17], L.current_stream(dev)))))
    for a in sppf_calls:  # (x, n, h, w, c1, ldx, w_packed, bias, y, c_, ldy, dtype, opts, stream): SPPF's cv1 + its three pools, one launch
        npx, c1_, cc_ = a[1] * a[2] * a[3], a[4], a[9]
        calls.append(("sppf_front_kernel<%d>" % (c1_ // 32), 2.0 * npx * c1_ * cc_, npx * c1_ * 2 + npx * 4 * cc_ * 2 + c1_ * cc_ * 2,
                      (lambda a=a: orig_sppf(*a[:13], L.current_stream(dev)))))
    for (a, bx, cl) in dstream_calls:  # (x, n, h, w, cin, ldx, box, cls, nc, stride, y, a_total, a0, best_keys, dtype, opts, stream): one Detect level, one launch
        import ctypes as C_
        npx, cin_, nc_ = a[1] * a[2] * a[3], a[4], a[8]
        wts = 9 * cin_ * (64 + 80) + 9 * (64 * 64 + 80 * 80) + 64 * 64 + 80 * nc_   # head.py:94-100: cv2 = 3x3, 3x3, 1x1 (64 = 4 reg_max); cv3 = 3x3, 3x3, 1x1 (nc)
        keys_ = a[13] is not None and R.current_opts() is not None and R.current_opts().keys_only
        nbytes = npx * cin_ * 2 + npx * (4 if keys_ else 4 + nc_) * 4 + (npx * 8 if a[13] is not None else 0) + wts * 2   # input + decoded rows (+ NMS keys) + weights
        calls.append(("detect_stream_kernel(DsParams)", 2.0 * npx * wts, nbytes,
                      (lambda a=a, bx=bx, cl=cl: orig_dstream(*a[:6], C_.byref(bx), C_.byref(cl), *a[8:16], L.current_stream(dev)))))
    for a in c2f32up_calls:  # (x, n, h, w, c1, ldx, up, up_c, up_ld, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, upc_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[7], a[9], a[18]
        wts = c1_ * 64 + nb_ * 18 * 32 * 32 + (2 + nb_) * 32 * c2_
        o_ = R.current_opts()
        name = ("void c2f32_stream1_kernel<%d>(C2fsParams)" % (c1_ // 64) if (c1_ <= 192 and (o_ is None or o_.c2f_stream != 1)) else
                "void c2f32_fused_kernel<1, 16, true>(C2f32Params)")
        calls.append((name, 2.0 * npx * wts, npx * (c1_ - upc_ * 3 // 4 + c2_) * 2 + wts * 2,  # block input (the upsampled channels at quarter size) + output + weights
                      (lambda a=a: orig_c2f32up(*a[:23], L.current_stream(dev)))))
    for a in c2f64_calls:  # (x, n, h, w, c1, ldx, up, up_c, up_ld, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, upc_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[7], a[9], a[18]
        wts = c1_ * 128 + nb_ * 18 * 64 * 64 + (2 + nb_) * 64 * c2_
        flops = 2.0 * npx * wts
        nbytes = npx * (c1_ - upc_ * 3 // 4 + c2_) * 2 + wts * 2  # block input (the upsampled channels at quarter size) + output + weights
        calls.append(("void c2f64_fused_kernel<%d, 10, %d>(C2f64Params)" % (nb_, 10 if nb_ == 2 else 20), flops, nbytes,
                      (lambda a=a: orig_c2f64(*a[:23], L.current_stream(dev)))))
    for a in btail_calls:  # (x, n, h, w, c, ldx, w3, b3, wt, bt, kind, nc, stride, y, a_total, a0, best_keys, dtype, opts, stream)
        npx, c_, kind, nc_ = a[1] * a[2] * a[3], a[4], a[10], a[11]
        cout = 64 if kind == 1 else nc_
        flops = 2.0 * npx * (9 * c_ * c_ + c_ * cout)
        nbytes = npx * c_ * 2 + npx * (4 if kind == 1 else nc_) * 4 + (9 * c_ * c_ + c_ * cout) * 2
        mt = 1 if (npx + 255) // 256 < torch.cuda.get_device_properties(dev).multi_processor_count else 2
        calls.append(("void conv_big_kernel<3, 1, 8, 1, %d, %d, %d>(BigParams)" % (mt, 4 if kind == 1 else (5 if c_ == 80 else 6), kind), flops, nbytes,
                      (lambda a=a: orig_btail(*a[:19], L.current_stream(dev)))))
    torch.cuda.synchronize(dev)
    fam = {}
    with torch.no_grad():
        for (name, flops, nbytes, replay) in calls:
            def body():
                for _ in range(reps):
                    replay()

            body()
            g = R.HipGraph()
            g.capture(body, device=dev)
            g.replay(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay(dev)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
            d["bytes"] += nbytes
    if torch.is_tensor(x) and model._stem_fusable(x, model._concat_placement()):
        # rows 0-1 run as one kernel that bypasses hip_conv2d (csrc/stem.hip: stem_conv_fused_kernel)
        n_, _, h_, w_ = x.shape
        with torch.no_grad(), R.static_buffers(pool):
            def body_f():
                for _ in range(reps):
                    model._fused_stem(x)
            body_f()
            g = R.HipGraph()
            g.capture(body_f, device=dev)
            g.replay(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay(dev)
            e1.record()
            torch.cuda.synchronize(dev)
        ca_ = model.model[0].conv
        k0_, c0_, s0_ = int(ca_.kernel_size[0]), int(ca_.out_channels), int(ca_.stride[0])
        fl = 2.0 * n_ * ((h_ // s0_) * (w_ // s0_) * c0_ * 3 * k0_ * k0_ + (h_ // (2 * s0_)) * (w_ // (2 * s0_)) * 2 * c0_ * 9 * c0_)
        by = n_ * 3 * h_ * w_ * es + n_ * (h_ // (2 * s0_)) * (w_ // (2 * s0_)) * 2 * c0_ * es
        o_ = R.current_opts()
        nw_ = 4 if (o_ is not None and o_.stemf_waves == 4) else 8
        name_ = ("void stem_conv_fused32_kernel<%d>(StemFusedParams)" % s0_ if c0_ == 32 else
                 "void stem_conv_fused_kernel<%d, %d>(StemFusedParams)" % (4 if k0_ == 6 else nw_, k0_))   # (the names rocprofv3 / the PMC tables print)
        fam[name_] = dict(launches=1, ms=e0.elapsed_time(e1) / reps, flops=fl, bytes=float(by))
    conv_ms = sum(d["ms"] for d in fam.values())
    conv_flops = sum(d["flops"] for d in fam.values())
    conv_bytes = sum(d["bytes"] for d in fam.values())
    dom_name, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_BF16_TFLOPS if es == 2 else PEAK_F32_TFLOPS
    avg_s = dom["ms"] / dom["launches"] * 1e-3
    achieved_tf = dom["flops"] / dom["launches"] / avg_s / 1e12
    achieved_gbs = dom["bytes"] / dom["launches"] / avg_s / 1e9
    ai = dom["flops"] / dom["bytes"]
    bound = "mfma" if ai > peak * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    roofline = {
        "kernel": dom_name,
        "bound": bound,
        "achieved": round(achieved_tf if bound == "mfma" else achieved_gbs, 2),
        "peak": peak if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round((achieved_tf / peak) if bound == "mfma" else (achieved_gbs / PEAK_HBM_GBS), 4),
        "traffic": None,
        "launches_per_step": dom["launches"],
        "avg_launch_us": round(avg_s * 1e6, 2),
        "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
        "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
        "flop_per_byte": round(ai, 1),
        "achieved_tflops": round(achieved_tf, 2),
        "achieved_gbs": round(achieved_gbs, 1),
        "timing": f"HIP events around a hipGraph replay of {reps} back-to-back launches per layer, on the launch stream "
                  "(isolated kernel time; agrees with rocprofv3 AverageNs of `bench.py --serial`, while in the default "
                  "run the Detect branches overlap on side streams and rocprofv3 reports stretched durations)",
    }
    # HBM traffic per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see profiles/): only valid for
    # the configuration they were collected on
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_hbm_summary.json"), reverse=True):  # the latest round that measured this kernel
        try:
            pmc = json.loads(pf.read_text())
            if pmc.get("config") == f"{args.model} bs={args.batch} {args.dtype}" and dom_name in pmc["kernels"]:
                roofline["traffic"] = round(pmc["kernels"][dom_name]["hbm_bytes_per_launch"])
                roofline["traffic_source"] = f"profiles/{pf.name} (rocprofv3 --pmc, separate passes)"
                break
        except (OSError, KeyError, ValueError):
            pass
    # What bounds the quoted mode: with several steps in flight the small-map launches of other steps hide under the chip-filling
    # ones, so the step is (nearly) the SUM of the launches that fill the chip by themselves - listed here, each against the tighter
    # of its two rooflines, with the vector-issue time of its instruction count (PMC INSTS_VALU per launch over 1024 SIMDs at the
    # measured 2.5 cycles per wave-instruction and 2.1 GHz; transcendental instructions cost 8.2, so this is a lower bound)
    valu = {}
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_step_budget.txt"), reverse=True):
        try:
            for ln in pf.read_text().splitlines()[1:]:
                rest = ln[60:].split()  # (kernel name padded to 60 columns) calls INSTS_VALU INSTS_SALU ...
                if len(rest) > 2 and rest[0].isdigit():
                    valu.setdefault(ln[:60].strip(), float(rest[1]) / max(int(rest[0]), 1))
        except (OSError, ValueError, IndexError):
            pass
        break
    crit = []
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"] / kv[1]["launches"]):
        us = v["ms"] / v["launches"] * 1e3
        if us < 25.0:
            continue
        tf, gb = v["flops"] / (v["ms"] * 1e-3) / 1e12, v["bytes"] / (v["ms"] * 1e-3) / 1e9
        kn = k.replace("void ", "").split("(")[0]  # the PMC table strips "void " and the parameter list and cuts names at 60 columns
        vi = next((valu[n_] for n_ in valu if n_ and (kn == n_ or kn[:60].rstrip() == n_)), None)
        if vi is None:  # same kernel template, one instantiation in the table (its template list may be spelled with defaults)
            same = [n_ for n_ in valu if n_ and n_.split("<")[0] == kn.split("<")[0]]
            vi = valu[same[0]] if len(same) == 1 else None
        crit.append({"kernel": k, "launches": v["launches"], "avg_us": round(us, 1), "frac_mfma": round(tf / peak, 3),
                     "frac_hbm": round(gb / PEAK_HBM_GBS, 3), "frac_of_tighter_roofline": round(max(tf / peak, gb / PEAK_HBM_GBS), 3),
                     "valu_issue_us": None if vi is None else round(vi / 1024 * 2.5 / 2.1e3, 1)})
    kernels = {
        "critical_path": crit,
        "conv_ms_per_step": round(conv_ms, 4),
        "conv_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 1),
        "conv_algorithmic_gbs": round(conv_bytes / (conv_ms * 1e-3) / 1e9, 1),
        "conv_hbm_floor_ms": round(conv_bytes / 6.0e12 * 1e3, 4),
        "conv_algorithmic_bytes": conv_bytes,
        "conv_launches_per_step": int(sum(d["launches"] for d in fam.values())),
        "families": {k: dict(launches=v["launches"], avg_us=round(v["ms"] / v["launches"] * 1e3, 2),
                             tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                             gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)) for k, v in sorted(fam.items())},
    }
    return roofline, kernels
