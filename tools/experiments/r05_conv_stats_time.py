"""upa_conv2d_bn_stats per conv shape of the yolov8s training step (bs 32): statistics from the convolution's epilogue vs conv + reduction pass
(`upa_opts.no_epi_stats`), us per call (conv + statistics + combine).
    python3 tools/experiments/r05_conv_stats_time.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
st = L.current_stream(dev)
# cin, cout, k, s, h (= w)
SHAPES = [(64, 128, 3, 2, 160), (128, 256, 3, 2, 80), (256, 512, 3, 2, 40), (64, 64, 3, 1, 80), (128, 128, 3, 1, 40), (256, 256, 3, 1, 20),
          (128, 128, 3, 1, 80), (128, 64, 3, 1, 80), (256, 128, 3, 1, 40), (512, 128, 3, 1, 20), (256, 64, 3, 1, 40), (128, 128, 3, 2, 80),
          (256, 256, 3, 2, 40), (768, 512, 1, 1, 20), (1024, 512, 1, 1, 20), (32, 32, 3, 1, 160), (32, 64, 3, 2, 320)]
n = 32
print(f"{'cin':>4} {'cout':>4} k s {'HxW':>7} | {'epilogue':>9} {'pass':>9}")
for cin, cout, k, s, h in SHAPES:
    x = R.alloc_nhwc(n, cin, h, h, torch.bfloat16, dev)
    x.normal_()
    oh = (h + 2 * (k // 2) - k) // s + 1
    z = R.alloc_nhwc(n, cout, oh, oh, torch.bfloat16, dev)
    vx, vz = R.view_of(x), R.view_of(z)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wp = torch.empty(lib.upa_conv_packed_weight_bytes(cout, cin, k, L.UPA_BF16), dtype=torch.uint8, device=dev)
    L.check(lib.upa_pack_conv_weight_dev(w.data_ptr(), cout, cin, k, L.UPA_BF16, 0, wp.data_ptr(), st))
    ws = torch.zeros(lib.upa_channel_reduce_workspace_bytes(cout) // 8, dtype=torch.float64, device=dev)
    m, v, rm, rv = (torch.zeros(cout, device=dev) for _ in range(4))
    res = []
    for off in (0, 1):
        with R.use_opts(L.Opts(no_epi_stats=off)):
            def f():
                L.check(lib.upa_conv2d_bn_stats(vx.ptr, n, h, h, cin, vx.ld, wp.data_ptr(), vz.ptr, cout, vz.ld, k, s, k // 2, 0.03, m.data_ptr(),
                                                v.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws.data_ptr(), L.UPA_BF16, R.opts_ptr(), st))
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
    print(f"{cin:4d} {cout:4d} {k} {s} {h:3d}x{h:<3d} | {res[0]:9.1f} {res[1]:9.1f}")
