#!/usr/bin/env python3
"""yolov8n rows 2-3 - C2f(32, 32, n = 1) at 160 x 160 + Conv(32, 64, 3, 2), batch 32 - alone: the two launches (line-buffer block + conv_igemm)
against the ONE line-buffer launch (upa_c2f16_down_fused) at several rows-per-workgroup settings and both wave layouts of the stride-2 conv;
HIP events around 20 back-to-back calls."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from tests.hip_utils import DEV, bf16_round, bn_fix, to_dev_nhwc  # noqa: E402
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402
from ultralytics_pro_amd.nn import modules as pm  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

m = bn_fix(pm.C2f(32, 32, 1, True))
d = bn_fix(pm.Conv(32, 64, 3, 2))
P.apply_procedural_weights(m, family="default")
P.apply_procedural_weights(d, family="default")
m, d = m.to(DEV).eval(), d.to(DEV).eval()
x = to_dev_nhwc(bf16_round(P.uniform("c16t", (32, 32, 160, 160), -1.5, 1.5)), torch.bfloat16)


def t(fused, **o):
    def call():
        if fused:
            assert m.forward_down(x, d) is not None
        else:
            d(m(x))
    with torch.no_grad(), R.use_opts(L.Opts(**o)):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


print(f"two launches: {t(False):.1f} us (tile-form block: {t(False, c2f16_waves=4):.1f} us)")
for dw in (4,):
    for rows in (0, -1, 160, 80, 54, 40, 32, 20):
        print(f"one launch, {dw} stride-2 waves, input rows per workgroup {rows:3d}: {t(True, c2f_stream_rows=rows):.1f} us")


try:  # the -DUPA_STAMP build (make -C ultralytics_pro_amd/csrc stamp; UPA_HIP_LIB=ultralytics_pro_amd/libupa_hip_stamp.so): per-wave step stamps
    import ctypes as C
    import numpy as np
    rd = L.lib().upa_debug_stamps_c2f16s
except AttributeError:
    sys.exit(0)
rd.argtypes = [C.c_void_p, C.c_int]
t(True, c2f_stream_rows=80)
STEPS = 64
buf = np.zeros(4 * 12 * STEPS * 2, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4, 12, STEPS, 2).astype(np.int64)[0]
names = ["cv1u23+DMA", "t u01", "b u01", "cv2 u01", "cv1 u01", "t u2", "b u2", "cv2 u2", "down 0", "down 1", "down 2", "down 3"]
n = int((st[0, :, 0] > 0).sum())
print("workgroup 0: steps", n, "life", int(st[:, n - 1, 1].max() - st[:, 0, 0].min()), "cycles (s_memtime: 100 MHz)")
print("step   len | busy per wave: " + " ".join(f"{x:>10s}" for x in names))
for s_ in range(min(n, 24)):
    start = st[:, s_, 0].min()
    nxt = st[:, s_ + 1, 0].min() if s_ + 1 < n else st[:, s_, 1].max()
    print(f"{s_:3d} {int(nxt - start):6d} | " + " ".join(f"{int(b):10d}" for b in (st[:, s_, 1] - st[:, s_, 0])))
