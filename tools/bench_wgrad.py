#!/usr/bin/env python3
"""Micro-benchmark of upa_conv2d_wgrad: python tools/bench_wgrad.py cin cout k stride H N [dtype]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402


def run(cin, cout, k, s, H, N, dtype=torch.bfloat16, iters=10):
    dev = torch.device("cuda:0")
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, cin, device=dev).to(dtype).permute(0, 3, 1, 2)
    dz = torch.randn(N, OH, OH, cout, device=dev).to(dtype).permute(0, 3, 1, 2)
    dw = torch.zeros(cout, cin, k, k, device=dev)
    vx, vz = R.view_of(x), R.view_of(dz)
    st = L.current_stream(dev)
    ws = torch.empty(L.lib().upa_conv2d_wgrad_workspace_bytes(cin, cout, k), dtype=torch.uint8, device=dev)

    def call():
        L.check(L.lib().upa_conv2d_wgrad(vx.ptr, N, H, H, cin, vx.ld, vz.ptr, cout, vz.ld, dw.data_ptr(), k, s, p, 1, vx.dtype,
                                         ws.data_ptr(), ws.numel(), st))

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * OH * OH * cout * cin * k * k
    print(f"wgrad {cin}->{cout} k{k} s{s} {H}x{H} N={N} {str(dtype)[6:]}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        run(*[int(v) for v in a[:6]], dtype=torch.float32 if len(a) > 6 and a[6] == "f32" else torch.bfloat16)
    else:
        for cfg in [(64, 64, 3, 1, 160, 32), (64, 64, 3, 1, 160, 8), (64, 64, 3, 1, 160, 2), (128, 128, 3, 1, 80, 32), (256, 256, 3, 1, 40, 32),
                    (512, 512, 3, 1, 20, 32), (128, 128, 1, 1, 80, 32), (32, 64, 3, 2, 320, 32), (3, 32, 3, 2, 640, 32),
                    (64, 64, 3, 1, 80, 32), (128, 64, 3, 1, 80, 32)]:
            if cfg[0] == 3:
                continue
            run(*cfg)
