"""Candidates (best-class score > conf) and kept boxes per image of the headline batch: what the greedy NMS kernel's time depends on.
    python3 tools/experiments/r05_nms_counts.py [--first 0 --batches 8]"""
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
    for b in range(a.batches):
        x = P.synthetic_images(32, first=32 * b).to(dev).to(torch.bfloat16)
        with torch.no_grad():
            y = m(x)
            y = y[0] if isinstance(y, (tuple, list)) else y
            best = y[:, 4:].float().amax(1)
            cand = (best > 0.25).sum(1).cpu()
            det = non_max_suppression(y.float(), 0.25, 0.7, max_det=300)
        kept = torch.tensor([int(d.shape[0]) for d in det])
        print(f"batch {b}: candidates per image min {int(cand.min())} median {int(cand.median())} max {int(cand.max())} | kept min {int(kept.min())} "
              f"median {int(kept.median())} max {int(kept.max())} | top candidates {sorted(cand.tolist())[-4:]}")


if __name__ == "__main__":
    main()
