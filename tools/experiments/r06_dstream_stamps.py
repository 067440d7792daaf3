#!/usr/bin/env python3
"""The line-buffer Detect level kernel (csrc/detect_stream.hip) alone: launch time by HIP events (yolov8n's 80 x 80 level, batch 32) and,
with the -DUPA_STAMP build, the per-step timeline of every wave of a class and a box workgroup (s_memtime at the start of each step and just
before its barrier).
usage: [UPA_HIP_LIB=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so] python tools/experiments/r06_dstream_stamps.py [--rows 0] [--wg 0] [--tile]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.hip_utils import DEV, bf16_round, bn_fix, to_dev_nhwc  # noqa: E402
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402
from ultralytics_pro_amd.nn import modules as pm  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--wg", type=int, default=0)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--keys-only", type=int, default=1)
args = ap.parse_args()

# the head of the real yolov8n model on the feature maps its own backbone / neck produce for the bench's images (random activations would send
# the class tails through their rare fallback path: logits above 4, no unique maximum)
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
model = DetectionModel("yolov8n.yaml")
P.apply_procedural_weights(model)
model = model.to(DEV).eval()
model.set_compute_dtype(torch.bfloat16)
det = model.model[-1]
det.keep_raw, det.nms_keys, det.concurrent, det.scores_out = False, True, False, not args.keys_only
grabbed = {}
h = det.register_forward_pre_hook(lambda m, inp: grabbed.__setitem__("xs", [t.clone() for t in inp[0]]))
with torch.no_grad():
    model(P.synthetic_images(args.batch).to(DEV).to(torch.bfloat16).contiguous())
h.remove()
xs = grabbed["xs"]


def time_head(**opts):
    with torch.no_grad(), R.use_opts(L.Opts(**opts)):
        for _ in range(3):
            det(xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            det(xs)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


print(f"whole head (3 levels, eager launches, batch {args.batch}), us per call: tile form {time_head(detect_stream=0):.1f} | "
      f"line-buffer level 0 {time_head(detect_stream=2, detect_stream_rows=args.rows):.1f}")
try:
    rd = L.lib().upa_debug_stamps_dstream
except AttributeError:
    sys.exit(0)
rd.argtypes = [C.c_void_p, C.c_int]
STEPS = 48
buf = np.zeros(8 * 8 * STEPS * 2, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
allst = buf.reshape(8, 8, STEPS, 2).astype(np.int64)
for kind, slot, names in (("class", args.wg, ["B32a", "B32b", "A32a", "A32b", "A16", "T01", "B16", "T23D"]),
                          ("box", 4 + args.wg, ["A32a", "A32b", "B32a", "B32b", "T0", "T1", "T2", "T3D"])):
    st = allst[slot]
    ph = st[:, 44:48, 0]
    for wv in range(8):
        if ph[wv, 0] > 0:
            print(f"  {kind} wave {wv} ({names[wv]}), step 10, first unit: 1x1 MFMAs issued {ph[wv,1]-ph[wv,0]}, keys decoded {ph[wv,2]-ph[wv,1] if ph[wv,2] > 0 else -1}, "
                  f"decode + store done {ph[wv,3]-(ph[wv,2] if ph[wv,2] > 0 else ph[wv,1])} (cycles)")
    st = st.copy(); st[:, 44:48, :] = 0
    nsteps = int((st[0, :, 0] > 0).sum())
    if nsteps == 0:
        print(kind, "workgroup: no stamps")
        continue
    t0 = st[:, 0, 0].min()
    print(f"{kind} workgroup {args.wg}: steps {nsteps}, life {int(st[:, nsteps - 1, 1].max() - t0)} ticks (100 MHz -> x 24 = ~cycles)")
    print("step  len   | busy ticks per wave (start of step -> its barrier)")
    print("            | " + " ".join(f"{n:>5s}" for n in names))
    for s in range(nsteps):
        start = st[:, s, 0].min()
        nxt = st[:, s + 1, 0].min() if s + 1 < nsteps else st[:, s, 1].max()
        busy = st[:, s, 1] - st[:, s, 0]
        print(f"{s:3d} {int(nxt - start):6d} | " + " ".join(f"{int(b):5d}" for b in busy))
