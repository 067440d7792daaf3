"""CPU proxy for tuning a procedural weight family whose bf16 detections agree with its f32 detections (round-2 review item 2b).

The oracle model is run twice on the same procedural weights / images: plain f32, and a bf16 EMULATION with the product's
rounding points (input, BN-folded weights and every Conv output rounded to bf16; the Detect 1x1 outputs stay f32 as the fused
decode consumes accumulators).  Prints detection-set agreement (tests/hip_utils.detection_agreement) per recipe.
Test infrastructure only (imports oracle/)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import modules as om  # noqa: E402
from oracle import nms as onms  # noqa: E402
from oracle import tasks as ot  # noqa: E402
from tests.hip_utils import bf16_round, bf16_weight_oracle, detection_agreement  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402


def bf16_emulated(model):
    m = bf16_weight_oracle(model)
    for mod in m.modules():
        if isinstance(mod, om.Conv):
            mod.register_forward_hook(lambda _m, _i, out: bf16_round(out))
    return m


def run(name, family, B=2, conf=0.25, verbose=True):
    m = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m, family=family)
    m.eval()
    x = P.synthetic_images(B)
    with torch.no_grad():
        y = m(x)[0]
        yb = bf16_emulated(m)(bf16_round(x))[0]
    ref = [o.numpy() for o in onms.non_max_suppression(y, conf, 0.7, max_det=300)]
    mine = [o.numpy() for o in onms.non_max_suppression(yb, conf, 0.7, max_det=300)]
    a9, a5 = detection_agreement(mine, ref, 0.9), detection_agreement(mine, ref, 0.5)
    d = (y - yb).abs()
    sc = y[:, 4:].max(1).values
    above = float((sc > conf).float().mean())
    near = float(((sc - conf).abs() < 0.01).float().mean())
    if verbose:
        print(f"{name:12s} {family:22s} dets {a9['n_mine']:4d}/{a9['n_ref']:4d} above {above * 100:5.2f}% near(+-.01) {near * 100:5.3f}% "
              f"| @.9 R {a9['recall']:.3f} P {a9['precision']:.3f} | @.5 R {a5['recall']:.3f} P {a5['precision']:.3f} | matched box p99 "
              f"{a9['box_p99']:.3f} max {a9['box_max']:.3f} | head box p99 {float(np.quantile(d[:, :4].numpy(), 0.99)):.3f} max {float(d[:, :4].max()):.3f} "
              f"score max {float(d[:, 4:].max()):.4f}")
    return a9, a5


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["yolov8n"]
    fams = sys.argv[2].split(",") if len(sys.argv) > 2 else [None]
    for n in names:
        for f in fams:
            run(n, f or P.model_family(ot.DetectionModel(n + ".yaml")))
