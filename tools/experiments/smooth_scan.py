"""Scan conv gains for a 'smooth' procedural family: bf16-emulation noise and detection agreement vs the f32 oracle (CPU)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from oracle import nms as onms, tasks as ot
from tests.hip_utils import bf16_round, detection_agreement
from tools.experiments.smooth_family import bf16_emulated
from ultralytics_pro_amd.utils import procedural as P

name = sys.argv[1]
gains = [float(v) for v in sys.argv[2].split(",")]
target_above = float(sys.argv[3]) if len(sys.argv) > 3 else 0.006
cls_gain = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
box_gain = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
res_gain = float(sys.argv[6]) if len(sys.argv) > 6 else 0.3
x = P.synthetic_images(2)
for g in gains:
    P.SMOOTH_RECIPE["scan"] = (g, cls_gain, 0.0, box_gain)
    P.SMOOTH_DFL_SLOPE = res_gain  # (7th argument reused: DFL bias slope)
    m = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m, family="smooth:scan")
    m.eval()
    with torch.no_grad():
        y = m(x)[0]
    # calibrate the class bias: shift so that `target_above` of the anchors have max score > 0.25
    logit = torch.logit(y[:, 4:].clamp(1e-7, 1 - 1e-7))
    mx = logit.max(1).values.flatten()
    thr = float(torch.quantile(mx, 1 - target_above))
    shift = float(np.log(0.25 / 0.75)) - thr
    P.SMOOTH_RECIPE["scan"] = (g, cls_gain, round(shift, 2), box_gain)
    m = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m, family="smooth:scan")
    m.eval()
    with torch.no_grad():
        y = m(x)[0]
        yb = bf16_emulated(m)(bf16_round(x))[0]
    ref = [o.numpy() for o in onms.non_max_suppression(y, 0.25, 0.7, max_det=300)]
    mine = [o.numpy() for o in onms.non_max_suppression(yb, 0.25, 0.7, max_det=300)]
    a9, a5 = detection_agreement(mine, ref, 0.9), detection_agreement(mine, ref, 0.5)
    # threshold-band-excluded agreement: rows whose score lies within +-band of conf are not counted on either side
    band = 0.005
    def drop(rows_a, rows_b):
        return [a[np.abs(a[:, 4] - 0.25) > band] for a in rows_a], rows_b
    refx = [r[np.abs(r[:, 4] - 0.25) > band] for r in ref]
    minex = [r[np.abs(r[:, 4] - 0.25) > band] for r in mine]
    rec = detection_agreement(mine, refx, 0.9)["recall"]      # reference rows outside the band must be found
    prec = detection_agreement(minex, ref, 0.9)["precision"]  # my rows outside the band must exist in the reference
    print(f"   band-excluded (+-{band}): recall {rec:.4f} precision {prec:.4f}; rows in band ref {sum(len(r) for r in ref) - sum(len(r) for r in refx)} mine {sum(len(r) for r in mine) - sum(len(r) for r in minex)}")
    d = (y - yb).abs()
    lg = torch.logit(y[:, 4:].clamp(1e-7, 1 - 1e-7))
    box = y[:, :4]
    print(f"{name} gain {g:4.1f} bias {shift:6.2f} logit std {float(lg.std()):.2f} box wh mean {float(box[:, 2:].mean()):.1f} std {float(box[:, 2:].std()):.1f} | dets {a9['n_mine']}/{a9['n_ref']} "
          f"@.9 R {a9['recall']:.3f} P {a9['precision']:.3f} @.5 R {a5['recall']:.3f} P {a5['precision']:.3f} | matched box p99 {a9['box_p99']:.3f} max {a9['box_max']:.3f} "
          f"| head box p99 {float(np.quantile(d[:, :4].numpy(), 0.99)):.3f} max {float(d[:, :4].max()):.2f} score max {float(d[:, 4:].max()):.4f}", flush=True)
