"""Training step with one / two weight-gradient side streams (ring kernels on 128 workgroups per launch)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
def run(ns):
    m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m.to(dev)
    tr = DetectionTrainer(m, dtype=torch.bfloat16, wgrad_streams=ns)
    for _ in range(5):
        tr.step(x, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        tr.step(x, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e3
for ns in (1, 2, 1, 2, 0):
    print("weight-gradient streams", ns, ": %.3f ms" % run(ns))
