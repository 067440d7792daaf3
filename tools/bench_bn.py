#!/usr/bin/env python3
"""Per-shape timing of the BatchNorm + activation passes of the training step (upa_bn_stats / upa_bn_finalize /
upa_bn_act_fwd / upa_bn_act_bwd) against their HBM floors.  usage: python tools/bench_bn.py [--iters 20]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from ultralytics_pro_amd import _lib as L  # noqa: E402

SHAPES = [(32 * 320 * 320, 32), (32 * 160 * 160, 64), (32 * 160 * 160, 32), (32 * 80 * 80, 128), (32 * 80 * 80, 64),
          (32 * 40 * 40, 256), (32 * 40 * 40, 128), (32 * 20 * 20, 512), (32 * 20 * 20, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    st = L.current_stream(dev)
    print(f"{'npix':>9} {'c':>4} | {'stats':>7} {'final':>6} {'apply':>7} | {'bwd all':>8} | floors: stats apply bwd(reduce+apply) us @6 TB/s")
    for npix, c in SHAPES:
        z = torch.randn(npix, c, device=dev).to(torch.bfloat16)
        dy = torch.randn(npix, c, device=dev).to(torch.bfloat16)
        y = torch.empty_like(z)
        dz = torch.empty_like(z)
        ws = torch.zeros(lib.upa_channel_reduce_workspace_bytes(c) // 8, dtype=torch.float64, device=dev)
        mean, var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        fns = {
            "stats": lambda: L.check(lib.upa_bn_stats(z.data_ptr(), npix, c, c, ws.data_ptr(), 1, st)),
            "final": lambda: L.check(lib.upa_bn_finalize(ws.data_ptr(), npix, c, 0.03, mean.data_ptr(), var.data_ptr(), rm.data_ptr(), rv.data_ptr(), st)),
            "apply": lambda: L.check(lib.upa_bn_act_fwd(z.data_ptr(), npix, c, c, mean.data_ptr(), var.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                        1e-3, 1, y.data_ptr(), c, None, 0, 1, st)),
            "bwd": lambda: L.check(lib.upa_bn_act_bwd(z.data_ptr(), dy.data_ptr(), npix, c, c, c, mean.data_ptr(), var.data_ptr(), g.data_ptr(),
                                                      b.data_ptr(), 1e-3, 1, dz.data_ptr(), c, dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), 1, st)),
        }
        res = {}
        for k, f in fns.items():
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k] = e0.elapsed_time(e1) / a.iters * 1e3
        nb = npix * c * 2
        print(f"{npix:9d} {c:4d} | {res['stats']:7.1f} {res['final']:6.1f} {res['apply']:7.1f} | {res['bwd']:8.1f} | "
              f"{nb / 6e6:6.1f} {2 * nb / 6e6:6.1f} {5 * nb / 6e6:6.1f}")


if __name__ == "__main__":
    main()
