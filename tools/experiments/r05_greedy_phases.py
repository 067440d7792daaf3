"""Where the greedy NMS kernel spends its cycles (library built with -DUPA_GREEDY_PROF, loaded through UPA_HIP_LIB: r05_greedy_phases.sh):
shader cycles of wave 0 per phase, mean over the images of a batch, for the headline single-label call (conf 0.25) and the validate call
(conf 0.001, multi_label).
    UPA_HIP_LIB=/tmp/libupa_hip_prof.so python3 tools/experiments/r05_greedy_phases.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402
from ultralytics_pro_amd.utils.nms import nms_raw  # noqa: E402

dev = torch.device("cuda:0")
m = DetectionModel("yolov8n.yaml")
P.apply_procedural_weights(m)
m = m.to(dev).eval()
m.set_compute_dtype(torch.bfloat16)
raw = C.CDLL(str(L.LIB_PATH))
NAMES = {0: "init", 1: "stage load", 2: "phase 1 (vs kept list)", 3: "barrier 1", 8: "phase 2: alive mask", 9: "phase 2: suppression columns", 10: "phase 2: resolve",
         11: "phase 2: output rows", 4: "phase 2: rest", 5: "barrier 2 (+ rows barrier)"}


def prof(label, reps):
    buf = (C.c_ulonglong * 12)()
    assert raw.upa_debug_greedy_prof(buf) == 0
    wgs = max(1, buf[7])
    tot = sum(buf[i] for i in NAMES)
    print(f"{label}: {reps} calls, {wgs} workgroups, {buf[6] / wgs:.1f} chunks of 64 per image; cycles per image: total {tot / wgs:.0f}")
    for i, nm in NAMES.items():
        print(f"    {nm:26s} {buf[i] / wgs:9.0f}  ({100.0 * buf[i] / max(1, tot):4.1f} %)   {buf[i] / max(1, buf[6]):7.0f} per chunk")


for bi in range(2):
    x = P.synthetic_images(32, first=32 * bi).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        y = m(x)
        y = y[0] if isinstance(y, (tuple, list)) else y
        torch.cuda.synchronize()
        raw.upa_debug_greedy_prof((C.c_ulonglong * 12)())
        for _ in range(5):
            nms_raw(y, 0.25, 0.7, max_det=300, key="hot")
        torch.cuda.synchronize()
        prof(f"batch {bi} single-label conf 0.25", 5)
        yf = y.float().contiguous()
        for _ in range(5):
            nms_raw(yf, 0.001, 0.7, multi_label=True, max_det=300, key="val")
        torch.cuda.synchronize()
        prof(f"batch {bi} multi-label conf 0.001 (three greedy launches per call, two of them return at once)", 5)
