#!/usr/bin/env python3
"""Time the HIP training step (forward + loss + backward + optimizer) of a detection model on one GPU.
usage: python tools/bench_train.py [--model yolov8s] [--batch 32] [--imgsz 640] [--dtype bf16] [--steps 5]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from ultralytics_pro_amd.engine.trainer import DetectionTrainer  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8s")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture of the step")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = DetectionModel(a.model + ".yaml")
    P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32, device=dev)
    x = P.synthetic_images(a.batch, h=a.imgsz, w=a.imgsz).to(dev)
    lab = P.synthetic_labels(a.batch)
    if a.eager:
        for _ in range(2):
            items = tr.step(x, lab)
    else:
        tr.compile(x, lab)
        items = tr.step(x, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        items = tr.step(x, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.model} bs={a.batch} {a.imgsz} {a.dtype}: {dt * 1e3:.2f} ms/step  {a.batch / dt:.1f} img/s  loss items {items.tolist()}  "
          f"pool {tr.pool.nbytes() / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
