"""Time of the NMS launches (sort + greedy, single-label keys path) on the head output of the headline batches, per batch.
    python3 tools/experiments/r05_nms_time.py [--batches 8]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402
from ultralytics_pro_amd.utils.nms import non_max_suppression  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m = m.to(dev).eval()
    m.set_compute_dtype(torch.bfloat16)
    tot = 0.0
    for b in range(a.batches):
        x = P.synthetic_images(32, first=32 * b).to(dev).to(torch.bfloat16)
        with torch.no_grad():
            y = m(x)
            y = (y[0] if isinstance(y, (tuple, list)) else y).float().contiguous()
            for _ in range(3):
                non_max_suppression(y, 0.25, 0.7, max_det=300)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                non_max_suppression(y, 0.25, 0.7, max_det=300)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print(f"batch {b}: {us:7.1f} us per NMS call (candidates + sort + greedy launches, host wrapper included)")
    print(f"mean {tot / a.batches:.1f} us")


if __name__ == "__main__":
    main()
