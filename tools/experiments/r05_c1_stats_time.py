"""Time of the pointwise convolution WITH the statistics epilogue (upa_conv2d_bn_stats, conv1x1_stream_kernel<.., 3>) per shape of the yolov8s
training step under forced (pixel tiles per wave, waves) - the NTW = 8 variants run close to the register limit.
    python3 tools/experiments/r05_c1_stats_time.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
st = L.current_stream(dev)
SHAPES = [(64, 64, 32 * 160 * 160), (96, 64, 32 * 160 * 160), (128, 128, 32 * 80 * 80), (384, 128, 32 * 80 * 80), (256, 256, 32 * 40 * 40),
          (384, 256, 32 * 40 * 40), (512, 512, 32 * 20 * 20), (128, 128, 32 * 40 * 40)]
print(f"{'cin':>4} {'cout':>4} {'px':>8} | " + " ".join(f"{k:>9}" for k in ("auto", "mt1 w8", "mt2 w8", "mt1 w4", "mt2 w4", "mt4 w4", "no stats")))
for cin, cout, px in SHAPES:
    x = torch.randn(px, cin, device=dev).to(torch.bfloat16)
    z = torch.empty(px, cout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    wp = torch.empty(lib.upa_conv_packed_weight_bytes(cout, cin, 1, L.UPA_BF16), dtype=torch.uint8, device=dev)
    L.check(lib.upa_pack_conv_weight_dev(w.data_ptr(), cout, cin, 1, L.UPA_BF16, 0, wp.data_ptr(), st))
    ws = torch.zeros(lib.upa_channel_reduce_workspace_bytes(cout) // 8, dtype=torch.float64, device=dev)
    m, v, rm, rv = (torch.zeros(cout, device=dev) for _ in range(4))
    res = []
    for kw in ({}, dict(c1_mt=1, c1_waves=8), dict(c1_mt=2, c1_waves=8), dict(c1_mt=1, c1_waves=4), dict(c1_mt=2, c1_waves=4),
               dict(c1_mt=4, c1_waves=4), dict(no_epi_stats=1)):
        with R.use_opts(L.Opts(**kw)):
            def f():
                L.check(lib.upa_conv2d_bn_stats(x.data_ptr(), 1, 1, px, cin, cin, wp.data_ptr(), z.data_ptr(), cout, cout, 1, 1, 0, 0.03,
                                                m.data_ptr(), v.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws.data_ptr(), L.UPA_BF16,
                                                R.opts_ptr(), st))
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{cin:4d} {cout:4d} {px:8d} | " + " ".join(f"{r:9.1f}" for r in res))
