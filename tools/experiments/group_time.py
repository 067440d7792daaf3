"""upa_conv2d_bias_act_group: the two first convs of the 80 x 80 Detect level (64 -> 64 and 64 -> 80 on the same input) as one grid vs
two launches (hipGraph of 20 repetitions each)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine import runtime as R
from ultralytics_pro_amd.nn.modules.conv import PackedConv
dev = torch.device("cuda:0")
n, h, w = 32, 80, 80
x = R.alloc_nhwc(n, 64, h, w, torch.bfloat16, dev); x.normal_()
pks = [PackedConv(torch.randn(c, 64, 3, 3) * 0.05, torch.randn(c), 3, dev, torch.bfloat16, False) for c in (64, 80)]
outs = [R.alloc_nhwc(n, c, h, w, torch.bfloat16, dev) for c in (64, 80)]
probs = (L.ConvProblem * 2)()
for j, c in enumerate((64, 80)):
    vx, vy = R.view_of(x), R.view_of(outs[j])
    probs[j] = L.ConvProblem(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, pks[j].w.data_ptr(), pks[j].bias.data_ptr(), vy.ptr, c, vy.ld, None, 0)
for tag, opts in (("one grid", L.Opts(conv_ws3=1)), ("two launches", L.Opts(conv_ws3=1, no_group=1))):
    def launch():
        L.check(L.lib().upa_conv2d_bias_act_group(C.cast(probs, C.c_void_p), 2, 3, 1, 1, L.ACT_SILU, L.UPA_BF16, C.pointer(opts), L.current_stream(dev)))
    launch(); torch.cuda.synchronize()
    g = R.HipGraph(); g.capture(lambda: [launch() for _ in range(20)], device=dev); g.replay(dev); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(dev); e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 50:.1f} us")
