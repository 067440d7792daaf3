"""upa_linear (RT-DETR decoder shapes) timed alone: hipGraph of 20 back-to-back launches."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine import runtime as R
from ultralytics_pro_amd.nn.modules.conv import PackedConv
dev = torch.device("cuda:0")
for (m, k, n) in [(4800, 256, 256), (4800, 256, 512), (4800, 256, 1024), (4800, 1024, 256), (4800, 256, 80), (4800, 256, 288), (134400, 256, 256)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, 1, 1) * 0.05
    b = torch.randn(n)
    pk = PackedConv(w, b, 1, dev, torch.float32, False)
    y = torch.empty(m, n, device=dev)
    def launch():
        L.check(L.lib().upa_linear(x.data_ptr(), m, k, k, pk.w.data_ptr(), pk.bias.data_ptr(), y.data_ptr(), n, n, None, 0, 0, L.current_stream(dev)))
    launch(); torch.cuda.synchronize()
    ref = x @ w.reshape(n, k).t().to(dev) + b.to(dev)
    err = float((y - ref).abs().max())
    g = R.HipGraph(); g.capture(lambda: [launch() for _ in range(20)], device=dev); g.replay(dev); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(dev); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"M {m} K {k} N {n}: {us:7.1f} us  {2.0 * m * k * n / us / 1e6:7.1f} TFLOP/s  max|err| {err:.2e}")
