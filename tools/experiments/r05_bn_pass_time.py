"""The element-wise / reduction passes of a training-mode Conv (upa_bn_act_fwd; upa_bn_act_bwd = reduce + combine + apply) per output
shape of the yolov8s training step (bs 32), one call at a time: us per call and the rate over the algorithmic bytes (bf16: forward
2 + 2 B per value, backward reduce 4 B, apply 6 B).
    python3 tools/experiments/r05_bn_pass_time.py            # run under rocprofv3 --kernel-trace --stats for the per-kernel split"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
st = L.current_stream(dev)
# (cout, h) and how many layers of the model have that output shape
SHAPES = [(32, 320, 1), (64, 160, 1), (32, 160, 2), (64, 160, 2), (128, 80, 3), (64, 80, 6), (256, 40, 3), (128, 40, 8), (512, 20, 3), (256, 20, 5),
          (64, 40, 1), (64, 20, 1), (128, 20, 1)]
n = 32
tot = [0.0, 0.0]
print(f"{'cout':>4} {'HxW':>7} layers | {'fwd us':>8} {'TB/s':>5} | {'bwd us':>8} {'TB/s':>5}")
for c, h, cnt in SHAPES:
    z = R.alloc_nhwc(n, c, h, h, torch.bfloat16, dev); z.normal_()
    dy = R.alloc_nhwc(n, c, h, h, torch.bfloat16, dev); dy.normal_()
    y = R.alloc_nhwc(n, c, h, h, torch.bfloat16, dev)
    dz = R.alloc_nhwc(n, c, h, h, torch.bfloat16, dev)
    vz, vdy, vy, vdz = R.view_of(z), R.view_of(dy), R.view_of(y), R.view_of(dz)
    npix = n * h * h
    ws = torch.zeros(lib.upa_channel_reduce_workspace_bytes(c) // 8, dtype=torch.float64, device=dev)
    m, g, b, dg, db = (torch.zeros(c, device=dev) for _ in range(5))
    v = torch.ones(c, device=dev); g += 1

    def fwd():
        L.check(lib.upa_bn_act_fwd(vz.ptr, npix, c, vz.ld, m.data_ptr(), v.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-3, L.ACT_SILU, vy.ptr, vy.ld,
                                   None, 0, L.UPA_BF16, st))

    def bwd():
        L.check(lib.upa_bn_act_bwd(vz.ptr, vdy.ptr, npix, c, vz.ld, vdy.ld, m.data_ptr(), v.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-3, L.ACT_SILU,
                                   vdz.ptr, vdz.ld, dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), L.UPA_BF16, st))
    res = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    el = npix * c
    print(f"{c:4d} {h:3d}x{h:<3d} {cnt:6d} | {res[0]:8.1f} {el * 4 / res[0] / 1e6:5.2f} | {res[1]:8.1f} {el * 10 / res[1] / 1e6:5.2f}")
    tot[0] += res[0] * cnt
    tot[1] += res[1] * cnt
print(f"sum over the layers counted: forward {tot[0]:.0f} us, backward {tot[1]:.0f} us per step")
