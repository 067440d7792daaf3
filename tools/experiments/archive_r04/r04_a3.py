"""Main chain alone (upa_conv2d_wgrad replaced by a no-op): kernel time vs wall time of the training step.  Run under
rocprofv3 --kernel-trace --stats to get the kernel sum; prints the wall time."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
L.lib().upa_conv2d_wgrad = lambda *a: 0
for _ in range(4):
    tr.step(x, lab)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    tr.step(x, lab)
e1.record(); torch.cuda.synchronize()
print("main chain wall ms/step", e0.elapsed_time(e1) / 20)
