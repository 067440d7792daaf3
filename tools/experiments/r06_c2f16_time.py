#!/usr/bin/env python3
"""model.2 of yolov8n - C2f(32, 32, n = 1) at 160 x 160, batch 32 - alone: the 16 x 16 tile form (c2f16_fused_kernel) against the line-buffer form
(csrc/c2f16_stream.hip) at several rows-per-workgroup settings; HIP events around 20 back-to-back calls."""
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
P.apply_procedural_weights(m, family="default")
m = m.to(DEV).eval()
x = to_dev_nhwc(bf16_round(P.uniform("c16t", (32, 32, 160, 160), -1.5, 1.5)), torch.bfloat16)


def t(**o):
    with torch.no_grad(), R.use_opts(L.Opts(**o)):
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m(x)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


print(f"tile form (4 waves) {t(c2f16_waves=4):.1f} us | (8 waves) {t(c2f16_waves=8):.1f} us")
for rows in (0, 80, 54, 40, 32, 20):
    print(f"line-buffer, rows {rows:3d}: {t(c2f_stream_rows=rows):.1f} us")


try:
    import ctypes as C
    import numpy as np
    rd = L.lib().upa_debug_stamps_c2f16s
except AttributeError:
    sys.exit(0)
rd.argtypes = [C.c_void_p, C.c_int]
t(c2f_stream_rows=80)
STEPS = 64
buf = np.zeros(4 * 8 * STEPS * 2, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4, 8, STEPS, 2).astype(np.int64)[0]
names = ["cv1u2+DMA", "t u01", "b u01", "cv2 u01", "cv1 u01", "t u2", "b u2", "cv2 u2"]
n = int((st[0, :, 0] > 0).sum())
print("workgroup 0: steps", n, "life", int(st[:, n - 1, 1].max() - st[:, 0, 0].min()), "cycles")
print("step   len | busy cycles per wave: " + " ".join(f"{x:>9s}" for x in names))
for s_ in range(min(n, 20)):
    start = st[:, s_, 0].min()
    nxt = st[:, s_ + 1, 0].min() if s_ + 1 < n else st[:, s_, 1].max()
    print(f"{s_:3d} {int(nxt - start):6d} | " + " ".join(f"{int(b):9d}" for b in (st[:, s_, 1] - st[:, s_, 0])))
