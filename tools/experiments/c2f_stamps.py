#!/usr/bin/env python3
"""Phase times of the whole-block C2f kernel (csrc/c2f64.hip) on one layer of a model: needs UPA_HIP_LIB=<-DUPA_STAMP library>.
usage: UPA_HIP_LIB=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so python tools/experiments/c2f_stamps.py [--layer 6] [--batch 32]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from stamps import report  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8n")
    ap.add_argument("--layer", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-stamps", action="store_true", help="timing only (any library)")
    ap.add_argument("--opts", default="", help="upa_opts fields as name=value,... for every call of this process (A/B)")
    args = ap.parse_args()
    if args.opts:
        from ultralytics_pro_amd import _lib as L
        from ultralytics_pro_amd.engine import runtime as R
        R.set_default_opts(L.Opts(**{k: int(v) for k, v in (kv.split("=") for kv in args.opts.split(","))}))
    dev = torch.device("cuda:0")
    m = DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(m)
    m = m.to(dev).eval()
    m.set_compute_dtype(torch.bfloat16)
    x = P.synthetic_images(args.batch).to(dev).to(torch.bfloat16)
    grabbed = {}
    mod = m.model[args.layer]
    h = mod.register_forward_pre_hook(lambda _m, inp: grabbed.setdefault("x", inp[0]))
    with torch.no_grad():
        m(x)
    h.remove()
    xin = grabbed["x"]
    print(f"layer {args.layer}: {type(mod).__name__} input {tuple(xin.shape)}")

    def launch():
        with torch.no_grad():
            mod(xin)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launch()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    print(f"  eager {e0.elapsed_time(e1) * 100:.1f} us per call (host-launch bound if small)")
    from ultralytics_pro_amd.engine import runtime as R
    gr = R.HipGraph()
    gr.capture(lambda: [launch() for _ in range(20)], device=dev)
    gr.replay(dev)
    torch.cuda.synchronize()
    e0.record()
    gr.replay(dev)
    e1.record()
    torch.cuda.synchronize()
    print(f"  graph replay {e0.elapsed_time(e1) * 50:.1f} us per call")
    if args.no_stamps:
        return
    if xin.shape[1] == 64:
        report(launch, "c2f", ["x halo + weights", "cv1", "m1.cv1 3x3", "m1.cv2 3x3", "m2.cv1 3x3", "m2.cv2 3x3", "cv2 + stores", "store drain"])
        return
    report(launch, "c2f64", ["x chunk 0 wait", "cv1 (chunks)", "m1.cv1 3x3", "m1.cv2 3x3", "m2.cv1 3x3", "m2.cv2 3x3", "cv2 + stores", "store drain"])


if __name__ == "__main__":
    main()
