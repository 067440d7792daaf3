"""Host-side cost of one eager training step: time of each tr.step() call (no synchronisation) and where it goes."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev)
lab = P.synthetic_labels(32)
for _ in range(3):
    tr.step(x, lab)
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(40):
    tr.step(x, lab)
    ts.append(time.perf_counter())
torch.cuda.synchronize()
t1 = time.perf_counter()
print("host ms per step call:", [round((b - a) * 1e3, 2) for a, b in zip([t0] + ts[:-1], ts)], " total with sync per step:", round((t1 - t0) / 40 * 1e3, 2))
import ultralytics_pro_amd.engine.trainer as T
orig_fb, orig_opt, orig_up = tr.forward_backward, tr.optimizer_step, tr._upload_labels
acc = {"fb": 0.0, "opt": 0.0, "up": 0.0}
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    return g
tr.forward_backward = timed("fb", orig_fb); tr.optimizer_step = timed("opt", orig_opt); tr._upload_labels = timed("up", orig_up)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    tr.step(x, lab)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("40 steps: host issue", round((t1 - t0) * 1e3, 1), "ms, with sync", round((t2 - t0) * 1e3, 1), "ms; per step host sections (ms):", {k: round(v / 40 * 1e3, 2) for k, v in acc.items()})
