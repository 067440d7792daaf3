#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
run() {
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "tools")
import bench_wgrad as B
import torch
from ultralytics_pro_amd import _lib as L
from ultralytics_pro_amd.engine import runtime as R
# the stem: 3 channels stored as 8 (ld 8)
dev = torch.device("cuda:0")
N, H, cin, cout = 32, 640, 3, 32
x8 = torch.zeros(N, H, H, 8, device=dev, dtype=torch.bfloat16); x8[..., :3] = torch.randn(N, H, H, 3, device=dev).to(torch.bfloat16)
x = x8[..., :3].permute(0, 3, 1, 2)
dz = torch.randn(N, 320, 320, cout, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
dw = torch.zeros(cout, cin, 3, 3, device=dev)
vx, vz = R.view_of(x), R.view_of(dz)
st = L.current_stream(dev)
ws = torch.empty(L.lib().upa_conv2d_wgrad_workspace_bytes(cin, cout, 3), dtype=torch.uint8, device=dev)
def call(acc=1):
    L.check(L.lib().upa_conv2d_wgrad(vx.ptr, N, H, H, cin, vx.ld, vz.ptr, cout, vz.ld, dw.data_ptr(), 3, 2, 1, acc, vx.dtype, ws.data_ptr(), ws.numel(), st))
call(0); torch.cuda.synchronize()
ref = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, 3, 3), dz.float(), stride=2, padding=1)
print("stem wgrad max rel err", float((dw - ref).abs().max() / ref.abs().max()))
for _ in range(3): call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
print("stem wgrad us", e0.elapsed_time(e1) / 10 * 1e3)
PY
}
echo "== register-staged"; UPA_WGRAD_RING=0 run
echo "== ring 512"; run
echo "== ring 768"; UPA_WGRAD_STEM_WGS=768 run
echo "== ring 1024"; UPA_WGRAD_STEM_WGS=1024 run
