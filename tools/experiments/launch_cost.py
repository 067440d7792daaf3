"""CPU cost of enqueueing one step (hipGraphLaunch of ~100 kernel nodes) vs the GPU time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.utils.nms import nms_raw
from ultralytics_pro_amd.engine.pipeline import PipelinedRunner

dev = torch.device("cuda:0")
model = DetectionModel("yolov8n.yaml")
P.apply_procedural_weights(model)
model = model.to(dev).eval()
model.set_compute_dtype(torch.bfloat16)
x = P.synthetic_images(32).to(dev).to(torch.bfloat16).contiguous()
post = lambda o: nms_raw(o[0], 0.25, 0.7, max_det=300, key="bench")
with torch.no_grad():
    for nf in (1, 2, 4, 8):
        r = PipelinedRunner(model, x, post, micro_batches=1, in_flight=nf, linear=True)
        for _ in range(20):
            r.step()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            r.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"in_flight={nf}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step (CPU), total {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
