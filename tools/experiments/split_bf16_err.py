#!/usr/bin/env python3
"""Would a split-bf16 parity mode meet the 1e-3 tolerance?  (round-4 review, item 4.)  CPU study on the ORACLE (build container or any host):
every Conv2d of yolov8n is replaced by the sum of bf16 x bf16 products with f32 accumulation that a matrix-core kernel would compute when
activations and BN-folded weights are split into T bf16 terms each (x = x1 + x2 (+ x3), every term exactly representable in bf16):
   T = 1: x1 w1                                        (the perf mode's products, but with f32 storage between layers)
   T = 2: x1 w1 + x1 w2 + x2 w1                        (3 MFMAs per product; drops x2 w2 ~ 2^-16 relative)
   T = 3: all cross terms down to 2^-24 (6 MFMAs)      (x1 w1 + x1 w2 + x2 w1 + x1 w3 + x2 w2 + x3 w1)
and the decoded head output is compared with the plain f32 oracle on the procedural images.  Prints max |box| (px) and |score| deviations.
usage: python tools/experiments/split_bf16_err.py [--images 2] [--model yolov8n]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import tasks as ot  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


def split(t, terms):
    out, r = [], t
    for _ in range(terms):
        h = r.to(torch.bfloat16).float()
        out.append(h)
        r = r - h
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--model", default="yolov8n")
    args = ap.parse_args()
    torch.set_num_threads(8)
    m = ot.DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(m)
    m.fuse()
    x = P.synthetic_images(args.images)
    with torch.no_grad():
        y_ref = m(x)[0]
    convs = [c for c in m.modules() if isinstance(c, torch.nn.Conv2d)]
    for terms in (1, 2, 3):
        pairs = [(i, j) for i in range(terms) for j in range(terms) if i + j < terms]  # cross terms down to 2^(-8 (i + j + 1))
        saved = []
        for c in convs:
            saved.append(c.forward)
            ws = split(c.weight.data, terms)

            def fwd(inp, c=c, ws=ws):
                xs = split(inp, terms)
                acc = None
                for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):  # small terms first
                    t = F.conv2d(xs[i], ws[j], None, c.stride, c.padding, c.dilation, c.groups)
                    acc = t if acc is None else acc + t
                return acc + (c.bias.view(1, -1, 1, 1) if c.bias is not None else 0)
            c.forward = fwd
        with torch.no_grad():
            y = m(x)[0]
        for c, f in zip(convs, saved):
            c.forward = f
        d = (y - y_ref).abs()
        print(f"{args.model} {args.images} images, {terms}-term split ({len(pairs)} bf16 MFMA products per f32 product): "
              f"max |box d| {d[:, :4].max():.3e} px (p99.9 {d[:, :4].flatten().quantile(0.999):.3e}), max |score d| {d[:, 4:].max():.3e}")


if __name__ == "__main__":
    main()
