import sys, torch
sys.path.insert(0, '/root/repo')
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.engine import runtime as R
dev = torch.device('cuda:0')
m = DetectionModel('yolov8n.yaml'); P.apply_procedural_weights(m); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
x = P.synthetic_images(32).to(dev).to(torch.bfloat16).contiguous()
pool = R.BufferPool()
with torch.no_grad(), R.static_buffers(pool):
    for fn, name in ((lambda: m._fused_stem(x), 'fused'), (lambda: m.model[1](m.model[0](x)), 'two layers')):
        fn(); torch.cuda.synchronize()
        g = R.HipGraph(); g.capture(lambda: [fn() for _ in range(10)], device=dev); g.replay(dev); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(dev); e1.record(); torch.cuda.synchronize()
        print(name, e0.elapsed_time(e1) / 10 * 1e3, 'us')
