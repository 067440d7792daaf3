"""How much of the eager training step is the weight-gradient side stream?  The same step with upa_conv2d_wgrad replaced by a no-op
(wrong gradients; timing only) = the main chain alone."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev)
lab = P.synthetic_labels(32)
def timeit(n=30):
    for _ in range(3):
        tr.step(x, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("full step ms:", round(timeit(), 3))
lib = L.lib()
real = lib.upa_conv2d_wgrad
lib.upa_conv2d_wgrad = lambda *a: 0
print("without weight gradients ms:", round(timeit(), 3))
lib.upa_conv2d_wgrad = real
