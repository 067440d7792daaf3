"""Per-layer weight-gradient table of the yolov8s training step (isolated launches, kernel + partial-sum reduce)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine import runtime as R
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "yolov8s"
m = DetectionModel(name + ".yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
for _ in range(2):
    tr.step(x, lab)
torch.cuda.synchronize()
lib = L.lib(); st = L.current_stream(dev)
rows = {}
scratch = {}
for cv in tr.convs:
    if cv.x is None:
        continue
    vx = R.view_of(cv.x)
    oh, ow = (vx.h + 2 * cv.p - cv.k) // cv.s + 1, (vx.w + 2 * cv.p - cv.k) // cv.s + 1
    dz = scratch.setdefault((vx.n, cv.cout, oh, ow), torch.randn(vx.n, oh, ow, cv.cout, device=dev).to(torch.bfloat16))
    dw = torch.zeros(cv.cout, cv.cin, cv.k, cv.k, device=dev)
    ws = tr.ctx.wgrad_ws
    def call():
        L.check(lib.upa_conv2d_wgrad(vx.ptr, vx.n, vx.h, vx.w, cv.cin, vx.ld, dz.data_ptr(), cv.cout, cv.cout, dw.data_ptr(),
                                     cv.k, cv.s, cv.p, 1, vx.dtype, ws.data_ptr(), ws.numel(), st), "wgrad")
    call(); call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    key = (cv.cin, cv.cout, cv.k, cv.s, vx.h, vx.w)
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1; r[1] += us
print(" cin cout k s    HxW  calls  us/call  TFLOP/s  t_hbm t_mfma")
tot = 0.0; fh = 0.0; fm = 0.0
for (cin, cout, k, s, h, w), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    oh, ow = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * 32 * oh * ow * cin * cout * k * k
    by = 2.0 * 32 * (h * w * cin + oh * ow * cout) + 4.0 * cin * cout * k * k
    t_h, t_m = by / 8e12 * 1e6, fl / 2.5e15 * 1e6
    print(f"{cin:4d} {cout:4d} {k} {s} {h:3d}x{w:<3d} {n:5d} {us / n:8.1f} {fl / (us / n) / 1e6:8.1f} {t_h:6.1f} {t_m:6.1f}")
    tot += us; fh += n * t_h; fm += n * t_m
print(f"TOTAL wgrad {tot / 1e3:.3f} ms/step (hbm floor {fh / 1e3:.3f} ms, mfma floor {fm / 1e3:.3f} ms)")
