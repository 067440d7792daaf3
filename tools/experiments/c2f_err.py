"""Error of the fused C2f kernel and of the four-launch path against the oracle in plain f32 (no intermediate rounding)."""
import sys
sys.path.insert(0, ".")
import torch
from oracle import modules as om
from ultralytics_pro_amd.utils import procedural as P
from tests.hip_utils import DEV, bf16_round, bn_fix, to_cpu_nchw, to_dev_nhwc
from ultralytics_pro_amd.nn.modules import block as pm

for shape in [(2, 160, 160), (2, 37, 50)]:
    o = bn_fix(om.C2f(32, 32, 1, True)); p = bn_fix(pm.C2f(32, 32, 1, True))
    P.apply_procedural_weights(o); P.apply_procedural_weights(p)
    p = p.to(DEV)
    x = bf16_round(P.uniform(f"c2f{shape}", (shape[0], 32, shape[1], shape[2]), -1.5, 1.5))
    with torch.no_grad():
        ref = o(x)
        xd = to_dev_nhwc(x, torch.bfloat16)
        p.fuse_block = True
        y = to_cpu_nchw(p(xd))
        p.fuse_block = False
        y2 = to_cpu_nchw(p(xd))
    for nm, t in (("fused", y), ("4 launches", y2)):
        d = (t - ref).abs()
        print(shape, nm, "mean", d.mean().item(), "p99", d.flatten().quantile(0.99).item() if d.numel() < 16e6 else None, "max", d.max().item(),
              "border mean", torch.cat([d[:, :, :2].flatten(), d[:, :, -2:].flatten(), d[:, :, :, :2].flatten(), d[:, :, :, -2:].flatten()]).mean().item())
    print(shape, "fused vs 4 launches: differing fraction", ((y - y2).abs() > 0).float().mean().item())
