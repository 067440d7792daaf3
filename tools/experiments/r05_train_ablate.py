"""Which launches of the yolov8s training step are on its critical path?  Timing-only ablations (results are WRONG while a call is
skipped): the eager step is timed with one family of C-ABI calls turned into a no-op at a time - the drop in ms/step is what removing
those launches from the main stream could buy at most.

    python3 tools/experiments/r05_train_ablate.py [--steps 12]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd import _lib as L  # noqa: E402
from ultralytics_pro_amd.engine.trainer import DetectionTrainer  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


UNSUPPORTED = {"upa_conv2d_dgrad_s2"}  # calls answered with "outside the fused form" instead of being skipped (an A/B, results stay right)


class Proxy:
    def __init__(self, real, skip):
        self._real, self._skip = real, set(skip)

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if name in self._skip:
            return lambda *a: (-2 if name in UNSUPPORTED else 0)  # -2 = UPA_EUNSUPPORTED: the caller takes its fallback path
        return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--model", default="yolov8s")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = DetectionModel(a.model + ".yaml")
    P.apply_procedural_weights(model)
    tr = DetectionTrainer(model, dtype=torch.bfloat16, device=dev)
    x = P.synthetic_images(a.batch, h=640, w=640).to(dev)
    lab = P.synthetic_labels(a.batch)
    for _ in range(3):
        tr.step(x, lab)
    real = L.lib()
    orig = L.lib

    def run(skip):
        L.lib = (lambda: Proxy(real, skip)) if skip else orig
        try:
            tr.step(x, lab)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                tr.step(x, lab)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / a.steps * 1e3, t_issue / a.steps * 1e3
        finally:
            L.lib = orig

    # (round 5, later: the forward conv + statistics became one C call, upa_conv2d_bn_stats - the first ablations of
    # profiles/r05_train_ablation.txt were taken on the three-call form)
    cases = [("baseline", ()), ("no bn_act_fwd", ("upa_bn_act_fwd",)),
             ("no bn_act_bwd (reduce + combine + apply)", ("upa_bn_act_bwd",)), ("no wgrad", ("upa_conv2d_wgrad",)),
             ("no forward/backward convs", ("upa_conv2d_bias_act",)), ("baseline again", ())]
    for name, skip in cases:
        ms, iss = run(skip)
        print(f"{name:45s} {ms:7.3f} ms/step   (host issue {iss:6.3f} ms/step)", flush=True)
    # A/B: the Detect head's small levels on the main stream vs beside the 80 x 80 level on a second stream
    for rep in range(3):
        for fk in (False, True):
            tr.detect.fork_levels = fk
            ms, iss = run(())
            print(f"Detect levels {'two streams' if fk else 'one stream '}                        {ms:7.3f} ms/step   (host issue {iss:6.3f})", flush=True)
    # A/B: the stride-2 data gradients as phase conv + interleave pass (the fused call answered UPA_EUNSUPPORTED) vs one launch
    for rep in range(3):
        for skip in (("upa_conv2d_dgrad_s2",), ()):
            ms, iss = run(skip)
            print(f"stride-2 data gradient {'conv + interleave pass' if skip else 'one launch            '}   {ms:7.3f} ms/step   (host issue {iss:6.3f})", flush=True)
    # A/B: the step issued on a HIGH-priority stream (the weight-gradient side stream keeps normal priority)
    hi = torch.cuda.Stream(device=dev, priority=-1)
    for rep in range(3):
        ms, iss = run(())
        print(f"main stream: default priority                 {ms:7.3f} ms/step   (host issue {iss:6.3f})", flush=True)
        hi.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(hi):
            ms, iss = run(())
        torch.cuda.current_stream(dev).wait_stream(hi)
        print(f"main stream: high priority                    {ms:7.3f} ms/step   (host issue {iss:6.3f})", flush=True)
    # A/B: batch statistics from the convolution's epilogue (default) vs a reduction pass over z (upa_opts.no_epi_stats = 1)
    from ultralytics_pro_amd.engine import runtime as R
    for rep in range(3):
        for off in (1, 0):
            with R.use_opts(L.Opts(no_epi_stats=off)):
                ms, iss = run(())
            print(f"batch statistics {'by a pass over z          ' if off else 'from the conv epilogue    '}   {ms:7.3f} ms/step   (host issue {iss:6.3f})", flush=True)


if __name__ == "__main__":
    main()
