"""Training step with the main chain on a high-priority stream (the weight-gradient side stream stays at default priority):
does the hardware queue priority keep the main chain's small kernels ahead of the ring kernels' workgroups?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
def timeit(n=40):
    for _ in range(4):
        tr.step(x, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("default stream ms:", round(timeit(), 3))
hi = torch.cuda.Stream(device=dev, priority=-1)
with torch.cuda.stream(hi):
    print("main chain on a priority -1 stream ms:", round(timeit(), 3))
print("default stream again ms:", round(timeit(), 3))
