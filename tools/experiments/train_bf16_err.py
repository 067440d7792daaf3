"""bf16 training step vs the reference golden (tests/golden/train_yolov8n.npz): relative errors of the loss items and the
gradient norm per step - the numbers behind the statistical bounds of tests/test_hip_train.py."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
G = np.load(ROOT / "tests/golden/train_yolov8n.npz")
bs, imgsz, steps = int(G["bs"][0]), int(G["imgsz"][0]), int(G["steps"][0])
for dtype in (torch.float32, torch.bfloat16):
    m = DetectionModel("yolov8n.yaml"); P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=dtype, device=dev)
    for step in range(steps):
        x = P.synthetic_images(bs, h=imgsz, w=imgsz, seed=step).to(dev); lab = P.synthetic_labels(bs, seed=step)
        items = tr.forward_backward(x, lab); norm = tr.grad_norm(); torch.cuda.synchronize()
        ri, rn = G[f"loss_items_{step}"], float(G[f"grad_norm_{step}"][0])
        print(dtype, "step", step, "loss rel err", np.abs(items.cpu().numpy() - ri) / np.abs(ri), "grad norm rel err %.4f" % (abs(norm - rn) / rn))
        tr.optimizer_step()
