"""How far does the weight-gradient side stream trail the main stream at the end of backward?  Events: main stream reaches the
join / side stream done (timing only; monkeypatches DetectionTrainer._join_wgrad)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from ultralytics_pro_amd.engine import trainer as T
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
m = DetectionModel("yolov8s.yaml"); P.apply_procedural_weights(m); m = m.to(dev)
tr = T.DetectionTrainer(m, dtype=torch.bfloat16)
x = P.synthetic_images(32).to(dev); lab = P.synthetic_labels(32)
rec = []
orig = T.DetectionTrainer._join_wgrad
def join(self):
    ctx = self.ctx
    if ctx.wgrad_streams and ctx.wgrad_pending:
        e_main = torch.cuda.Event(enable_timing=True); e_main.record(torch.cuda.current_stream(self.device))
        e_side = torch.cuda.Event(enable_timing=True); e_side.record(ctx.wgrad_streams[0])
        rec.append((e_main, e_side))
    orig(self)
T.DetectionTrainer._join_wgrad = join
for _ in range(5):
    tr.step(x, lab)
torch.cuda.synchronize(); rec.clear()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    tr.step(x, lab)
e1.record(); torch.cuda.synchronize()
print("step ms", e0.elapsed_time(e1) / 20)
d = [a.elapsed_time(b) for a, b in rec]
print("side stream done after the main stream reached the join, ms: mean %.3f min %.3f max %.3f" % (sum(d) / len(d), min(d), max(d)))
