"""Validate-path NMS alone (conf 0.001, multi_label): candidates per image and, under rocprofv3 --kernel-trace, the duration of each
NMS kernel dispatch in order (is the second (sort, greedy) pair - the full-path fallback - doing work?)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.utils.nms import nms_raw
dev = torch.device("cuda:0")
m = DetectionModel("yolov8n.yaml"); P.apply_procedural_weights(m); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
x = P.synthetic_images(32).to(dev).to(torch.bfloat16).contiguous()
with torch.no_grad():
    y = m(x)[0]
    torch.cuda.synchronize()
    n = (y[:, 4:] > 0.001).sum((1, 2))
    print("multi-label candidates per image: min %d mean %.0f max %d" % (int(n.min()), float(n.float().mean()), int(n.max())))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out, counts, keep = nms_raw(y, 0.001, 0.7, max_det=300, multi_label=True, key="v")
        e1.record(); torch.cuda.synchronize()
        print("nms_raw val settings: %.1f us per call; kept mean %.1f" % (e0.elapsed_time(e1) / 5 * 1e3, float(counts.float().mean())))
