"""Experiment: throughput of independent lanes (one compiled graph per lane, round robin) for several lane batch sizes."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.utils.nms import nms_raw
from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
dev = torch.device('cuda:0')
m = DetectionModel('yolov8n.yaml'); P.apply_procedural_weights(m); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
import os
if os.environ.get('SERIAL_HEAD'): m.model[-1].concurrent = False
xs = P.synthetic_images(32).to(dev).to(torch.bfloat16).contiguous()
cfgs = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(32, 3, 2)]
for cfg in cfgs:
    lb, lanes, mb = cfg[:3]
    pr = cfg[3] if len(cfg) > 3 else 0
    sp = cfg[4] if len(cfg) > 4 else 0
    x = xs[:lb].contiguous()
    r = PipelinedRunner(m, x, post=lambda o: nms_raw(o[0], 0.25, 0.7, max_det=300, key=f"l{lb}{lanes}{mb}"), micro_batches=mb, in_flight=lanes,
                        priority=pr, sub_priority=sp)
    t = r.measure(steps=100 * (32 // lb), warmup=10)
    print(f"lane batch {lb} lanes {lanes} mb {mb} prio {pr} sub {sp}: {lb / t:9.1f} img/s")
