"""Host time to enqueue one eager training step vs the step time (is the eager step host-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P

dev = torch.device("cuda:0")
m = DetectionModel("yolov8s.yaml")
P.apply_procedural_weights(m)
tr = DetectionTrainer(m, dtype=torch.bfloat16, device=dev)
x = P.synthetic_images(32, h=640, w=640).to(dev)
labels = P.synthetic_labels(32)
for _ in range(3):
    tr.step(x, labels)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    tr.step(x, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.2f} ms/step (host), wall {1e3 * (t2 - t0) / n:.2f} ms/step")
