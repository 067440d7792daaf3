#!/usr/bin/env python3
"""Where does the line-buffer C2f kernel (csrc/c2f_stream.hip) differ from the tile form?  Prints error counts per row / column / channel.
usage: python tools/experiments/r05_c2fs_debug.py N H W [rows] [shortcut]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from tests.hip_utils import DEV, bf16_round, to_cpu_nchw, to_dev_nhwc  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402
from ultralytics_pro_amd.nn import modules as pm  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

N, H, W = (int(v) for v in sys.argv[1:4])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
m = pm.C2f(64, 64, 2, sc)
from tests.hip_utils import bn_fix
m = bn_fix(m)
P.apply_procedural_weights(m, family="default")
m = m.to(DEV).eval()
x = bf16_round(P.uniform("dbgx", (N, 64, H, W), -1.5, 1.5))
xd = to_dev_nhwc(x, torch.bfloat16)
with torch.no_grad():
    with R.use_opts(c2f_stream_rows=rows):
        y = to_cpu_nchw(m(xd)).float()
    with R.use_opts(c2f_stream=1):
        t = to_cpu_nchw(m(xd)).float()
d = (y - t).abs()
bad = d > 0.05
print("bad", int(bad.sum()), "of", bad.numel(), "max", float(d.max()))
print("per image", bad.sum((1, 2, 3)).tolist())
print("per row  ", bad.sum((0, 1, 3)).tolist())
print("per col  ", bad.sum((0, 1, 2)).tolist())
print("per chan ", bad.sum((0, 2, 3)).tolist())
