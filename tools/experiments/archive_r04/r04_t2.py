"""Phase times of the 3x3 ring weight-gradient kernel (needs `make -C ultralytics_pro_amd/csrc stamp` and
UPA_HIP_LIB=ultralytics_pro_amd/libupa_hip_stamp.so): usage r04_t2.py cin cout stride H"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine import runtime as R
from stamps import report
cin, cout, s, H = [int(v) for v in sys.argv[1:5]]
N, k, p = 32, 3, 1
dev = torch.device("cuda:0")
OH = (H + 2 * p - k) // s + 1
x = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
dz = torch.randn(N, OH, OH, cout, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
dw = torch.zeros(cout, cin, k, k, device=dev)
vx, vz = R.view_of(x), R.view_of(dz)
st = L.current_stream(dev)
ws = torch.empty(L.lib().upa_conv2d_wgrad_workspace_bytes(cin, cout, k), dtype=torch.uint8, device=dev)
def call():
    L.check(L.lib().upa_conv2d_wgrad(vx.ptr, N, H, H, cin, vx.ld, vz.ptr, cout, vz.ld, dw.data_ptr(), k, s, p, 1, vx.dtype,
                                     ws.data_ptr(), ws.numel(), st))
print(f"wgrad {cin}->{cout} k3 s{s} {H}x{H}")
report(call, "train", ["prologue", "DMA waits", "barriers", "DMA issue", "reads + MFMA", "flush"])
