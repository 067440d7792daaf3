#!/usr/bin/env python3
"""Per-step timeline of the line-buffer C2f kernel (csrc/c2f_stream.hip, -DUPA_STAMP build): for each wave of workgroup 0..7 the
s_memtime at the start of every step and just before its barrier.  Prints, per step, each wave's busy cycles and the step length.
usage: UPA_HIP_LIB=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so python tools/experiments/r05_c2fs_stamps.py [--wg 0]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.hip_utils import DEV, bf16_round, bn_fix, to_dev_nhwc  # noqa: E402
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.nn import modules as pm  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--wg", type=int, default=0)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--n1", action="store_true", help="the n = 1 block C2f(192, 64) (yolov8n model.15) instead of C2f(64, 64, n = 2) (model.4)")
args = ap.parse_args()
m = bn_fix(pm.C2f(192, 64, 1, False) if args.n1 else pm.C2f(64, 64, 2, True))
P.apply_procedural_weights(m, family="default")
m = m.to(DEV).eval()
x = to_dev_nhwc(bf16_round(P.uniform("st", (args.batch, 192 if args.n1 else 64, 80, 80), -1.5, 1.5)), torch.bfloat16)
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
STEPS = 48
rd = L.lib().upa_debug_stamps_c2fs
rd.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(8 * 16 * STEPS * 2, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
st = buf.reshape(8, 16, STEPS, 2).astype(np.int64)[args.wg]
names = ["B01", "B23", "C01", "D01", "C2", "D2", "E2", "E01", "Y0", "Y1", "Y2", "Y3", "F0", "F1", "F2", "DMA"]
if args.n1:
    names = ["B01", "C01", "B2", "Y21", "Y00", "Y10", "C2", "F1", "Y01", "Y11", "Y20", "F2", "D0", "D1", "F0", "D2"]
nsteps = int((st[0, :, 0] > 0).sum())
t0 = st[:, 0, 0].min()
print("workgroup", args.wg, "steps", nsteps, "life", int(st[:, nsteps - 1, 1].max() - t0), "cycles")
print("step  len   | busy cycles per wave (start of step -> its barrier)")
print("            | " + " ".join(f"{n:>5s}" for n in names))
for s in range(nsteps):
    start = st[:, s, 0].min()
    nxt = st[:, s + 1, 0].min() if s + 1 < nsteps else st[:, s, 1].max()
    busy = st[:, s, 1] - st[:, s, 0]
    print(f"{s:3d} {int(nxt - start):6d} | " + " ".join(f"{int(b):5d}" for b in busy))


