"""Helpers shared by the -m gpu parity tests (HIP path through the C ABI vs the oracle)."""

import numpy as np
import torch

from ultralytics_pro_amd.engine import runtime as R
from ultralytics_pro_amd.utils import procedural as P

DEV = torch.device("cuda:0")


def unit_input(name, shape, lo=-1.0, hi=1.0):
    return P.uniform(f"unit:{name}", shape, lo, hi)


def bn_fix(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return m.eval()


def to_dev_nhwc(x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """CPU NCHW f32 -> device NHWC view of dtype (via the HIP transpose kernel)."""
    return R.to_nhwc(x.to(DEV).contiguous(), dtype)


def to_cpu_nchw(y: torch.Tensor) -> torch.Tensor:
    out = R.to_nchw_f32(y)
    torch.cuda.synchronize()
    return out.cpu()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
