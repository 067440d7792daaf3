"""Fused stem / first-layer kernels: XCD-aware tile walk on vs off (upa_opts.no_xcd), isolated launch time (10 launches per graph)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.engine import runtime as R
dev = torch.device('cuda:0')
for cfg in ("yolov8n.yaml", "yolov5-BoT3.yaml", "yolov8s.yaml", "yolov3-tiny.yaml"):
    m = DetectionModel(cfg); P.apply_procedural_weights(m); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
    x = P.synthetic_images(32).to(dev).to(torch.bfloat16).contiguous()
    for no_xcd in (1, 0, 1, 0):
        pool = R.BufferPool()
        with torch.no_grad(), R.static_buffers(pool), R.use_opts(no_xcd=no_xcd):
            if m._stem_fusable(x, set()):
                fn, name = (lambda: m._fused_stem(x)), "fused stem"
            else:
                fn, name = (lambda: m.model[0](x)), "first layer"
            fn(); torch.cuda.synchronize()
            g = R.HipGraph(); g.capture(lambda: [fn() for _ in range(10)], device=dev); g.replay(dev); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(dev); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            print(f"{cfg:18s} {name:11s} no_xcd={no_xcd}  {min(ts):7.1f} us (min of 5)", flush=True)
