"""Where the fused stem kernel (yolov8n: Conv(3,16,3,2) -> Conv(16,32,3,2)) spends its cycles: library built with -DUPA_STEM_PROF
(r05_stem_phases.sh), shader cycles of wave 0 per phase, summed over the tiles of a workgroup, mean over the workgroups; bs 32, 640 x 640.
    UPA_HIP_LIB=/tmp/libupa_hip_stemprof.so python3 tools/experiments/r05_stem_phases.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

dev = torch.device("cuda:0")
m = DetectionModel("yolov8n.yaml")
P.apply_procedural_weights(m)
m = m.to(dev).eval()
m.set_compute_dtype(torch.bfloat16)
raw = C.CDLL(str(L.LIB_PATH))
x = P.synthetic_images(32).to(dev).to(torch.bfloat16)
NAMES = ["wait for the patch + barrier", "issue the next patch (LDS-DMA)", "stage 2 (stem tile)", "barrier", "stage 3 (second conv + stores)"]
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    raw.upa_debug_stem_prof(buf)
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    assert raw.upa_debug_stem_prof(buf) == 0
wgs, tiles = max(1, buf[6]), max(1, buf[5])
tot = sum(buf[i] for i in range(5))
print(f"{wgs // 5} workgroups per launch, {tiles / wgs:.2f} tiles per workgroup; cycles per workgroup {tot / wgs:.0f}, per tile {tot / tiles:.0f}")
for i, nm in enumerate(NAMES):
    print(f"    {nm:34s} {buf[i] / tiles:8.0f} per tile  ({100.0 * buf[i] / tot:4.1f} %)")
