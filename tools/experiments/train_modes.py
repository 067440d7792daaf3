"""yolov8s bs 32 training step: eager launches vs hipGraph replay, with 0 / 1 / 2 weight-gradient side streams."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
def t(tr, n=8):
    for _ in range(2): tr.step(x, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(x, lab)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for ws in (1, 0, 2):
    m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=torch.bfloat16, device=dev, wgrad_streams=ws)
    te = t(tr)
    # host time of an eager step: enqueue only
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(x, lab); th = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    tr.compile(x, lab, warm_steps=0)
    tg = t(tr)
    print(f"wgrad_streams {ws}: eager {te:.2f} ms (host enqueue of one step {th:.2f} ms)  graph {tg:.2f} ms", flush=True)
    del tr, m
