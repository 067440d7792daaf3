#!/usr/bin/env python3
"""Per-layer conv micro-benchmark: every distinct conv problem of a model (hooked from one eager forward) timed in
isolation with HIP events over back-to-back launches. Prints time, TFLOP/s, algorithmic GB/s and the roofline bound.
usage: python tools/bench_conv.py [--model yolov8n] [--batch 32] [--dtype bf16] [--iters 20]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch  # noqa: E402

from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402
from ultralytics_pro_amd.nn.modules import block as pblock  # noqa: E402
from ultralytics_pro_amd.nn.modules import conv as pconv  # noqa: E402
from ultralytics_pro_amd.nn.modules import head as phead  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


def stamp_report(launch):
    from stamps import report
    if R.current_opts().conv_ws3 != 1:
        report(launch, "conv_ws3", ["halo 0 + weights", "tile 0 (whole)", "tile 1: issue next halo", "tile 1: MFMA loop", "tile 1: vmcnt wait", "tile 1: epilogue", "rest"])
    report(launch, "conv_big", ["halo+slab0 wait"] + [f"tap {i}" for i in range(9)] + ["epilogue", "store drain"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8n")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="cin,cout,k,H filter, e.g. 64,64,3,80")
    ap.add_argument("--big-mode", type=int, default=-1, help="upa_opts.conv_big: 0 size rule, 1 never, 2 every eligible shape")
    ap.add_argument("--opts", default="env", help="upa_opts fields as name=value,... or `env` (UPA_* variables, the default)")
    ap.add_argument("--stamps", action="store_true", help="with a -DUPA_STAMP library (UPA_HIP_LIB): phase times of conv_big workgroups")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    opts = L.Opts.from_env() if args.opts == "env" else L.Opts(**{k: int(v) for k, v in (kv.split("=") for kv in args.opts.split(","))})
    if args.big_mode >= 0:
        opts.conv_big = args.big_mode
    R.set_default_opts(opts)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    es = 2 if dtype == torch.bfloat16 else 4
    m = DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(m)
    m = m.to(dev).eval()
    m.set_compute_dtype(dtype)
    x = P.synthetic_images(args.batch).to(dev)
    if dtype == torch.bfloat16:
        x = x.to(torch.bfloat16)
    calls = []
    orig = pconv.hip_conv2d

    def rec(xx, pk, stride, pad, act, out=None, residual=None, out_dtype=None, key=None, up=None):
        if up is not None:  # per-layer table: time the conv over the materialised concat buffer
            up.materialize()
        y = orig(xx, pk, stride, pad, act, out=out, residual=residual, out_dtype=out_dtype, key=key)
        calls.append((xx, pk, stride, pad, act, y, residual, out_dtype))
        return y

    for mod in (pconv, pblock, phead):
        mod.hip_conv2d = rec
    with torch.no_grad():
        m(x)
    for mod in (pconv, pblock, phead):
        mod.hip_conv2d = orig
    torch.cuda.synchronize()
    agg = {}
    tot_ms = tot_fl = tot_by = 0.0
    only = tuple(int(v) for v in args.only.split(",")) if args.only else None
    for (xx, pk, stride, pad, act, y, residual, odt) in calls:
        n, cin, h, w = xx.shape
        if only and (cin, pk.cout, pk.k, h) != only:
            continue
        oh, ow = y.shape[2], y.shape[3]
        fl = 2.0 * n * oh * ow * pk.cout * cin * pk.k * pk.k
        by = n * h * w * cin * xx.element_size() + n * oh * ow * pk.cout * es * (2 if residual is not None else 1) + \
            pk.cout * cin * pk.k * pk.k * es
        for _ in range(3):
            orig(xx, pk, stride, pad, act, out=y, residual=residual, out_dtype=odt)
        torch.cuda.synchronize()

        def body():
            for _ in range(args.iters):
                orig(xx, pk, stride, pad, act, out=y, residual=residual, out_dtype=odt)

        g = R.HipGraph()  # back-to-back launches replayed from a hipGraph: no host launch gaps in the measurement
        g.capture(body, device=dev)
        g.replay(dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay(dev)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        if args.stamps:
            stamp_report(lambda: orig(xx, pk, stride, pad, act, out=y, residual=residual, out_dtype=odt))
        var = -1 if pk.stem else L.lib().upa_conv_variant(n, h, w, cin, pk.cout, pk.k, stride, pad, L.dtype_code(dtype), R.opts_ptr())
        key = (cin, pk.cout, pk.k, stride, h, w, residual is not None, var)
        d = agg.setdefault(key, [0, 0.0, fl, by])
        d[0] += 1
        d[1] += ms
        tot_ms += ms
        tot_fl += fl
        tot_by += by
    print(f"{'cin':>4} {'cout':>4} k s {'HxW':>9} res {'var':>7} calls {'us/call':>8} {'TFLOP/s':>8} {'GB/s':>7} {'t_hbm':>6} {'t_mfma':>6}")
    peak = 2500e12 if es == 2 else 157e12
    for key, (cnt, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        cin, cout, k, s, h, w, res, var = key
        us = ms / cnt * 1e3
        print(f"{cin:4d} {cout:4d} {k} {s} {h:4d}x{w:<4d} {int(res):3d} {(var & 0xfffffff) if var >= 0 else 0xffffff:7x} {cnt:5d} {us:8.1f} {fl / (us * 1e-6) / 1e12:8.1f} "
              f"{by / (us * 1e-6) / 1e9:7.0f} {by / 6.0e12 * 1e6:6.1f} {fl / peak * 1e6:6.1f}")
    print(f"TOTAL conv {tot_ms:.3f} ms/step  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s  {tot_by / tot_ms / 1e6:.0f} GB/s  "
          f"(hbm floor {tot_by / 6.0e12 * 1e3:.3f} ms, mfma floor {tot_fl / peak * 1e3:.3f} ms)")


if __name__ == "__main__":
    main()
