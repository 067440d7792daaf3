"""Detection validation metrics with the reference's function names (ultralytics/utils/metrics.py,
engine/validator.py:267-308, models/yolo/detect/val.py:274-288).

IoU and the greedy prediction-to-label matching run on the GPU for a whole batch at once (`upa_match_predictions` on the
fixed-shape NMS outputs: no per-image host round trip); what remains on the host is what the reference's `DetMetrics.process`
does on the host too - the per-class cumulative sums and the 101-point AP integration over a few thousand rows, once per
validation run (`ap_per_class` + `mean_results` reproduce P, R, mAP50, mAP50-95 bit for bit).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L


def box_iou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """(N,4), (M,4) xyxy on the GPU -> (N,M) IoU (utils/metrics.py:54-74)."""
    L.require_gpu(box1, "box_iou")
    b1, b2 = box1.float().contiguous(), box2.float().contiguous()
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    if out.numel() == 0:
        return out
    L.check(L.lib().upa_box_iou(b1.data_ptr(), b1.shape[0], b2.data_ptr(), b2.shape[0], float(eps), out.data_ptr(),
                                L.current_stream(b1.device)), "box_iou")
    return out


IOUV = np.linspace(0.5, 0.95, 10).astype(np.float32)  # torch.linspace(0.5, 0.95, 10), detect/val.py:59


def match_predictions_batched(det: torch.Tensor, counts: torch.Tensor, gt: torch.Tensor, ngt: torch.Tensor, iouv=IOUV,
                              out: torch.Tensor | None = None) -> torch.Tensor:
    """True-positive matrices of a whole batch on the GPU (`upa_match_predictions`): det (B, max_det, 6) + counts (B,) as
    `nms_raw` returns them, gt (B, max_gt, 5) rows [cls, x1, y1, x2, y2] + ngt (B,) -> (B, max_det, 10) uint8, no host sync
    (engine/validator.py:267-308 through models/yolo/detect/val.py:274-288)."""
    L.require_gpu(det, "match_predictions")
    b, max_det, _ = det.shape
    thr = np.ascontiguousarray(np.asarray(iouv, dtype=np.float32))
    tp = out if out is not None else torch.empty((b, max_det, thr.shape[0]), dtype=torch.uint8, device=det.device)
    L.check(L.lib().upa_match_predictions(det.data_ptr(), counts.data_ptr(), b, max_det, gt.data_ptr(), ngt.data_ptr(),
                                          int(gt.shape[1]), thr.ctypes.data, int(thr.shape[0]), tp.data_ptr(),
                                          L.current_stream(det.device)), "match_predictions")
    return tp


def process_batch(pred_boxes: torch.Tensor, pred_cls: torch.Tensor, gt_boxes: torch.Tensor, gt_cls: torch.Tensor) -> np.ndarray:
    """True-positive matrix of one image (detect/val.py:274-288) - the single-image form of `match_predictions_batched`."""
    n, m = int(pred_cls.shape[0]), int(gt_cls.shape[0])
    if m == 0 or n == 0:
        return np.zeros((n, len(IOUV)), dtype=bool)
    dev = pred_boxes.device
    det = torch.zeros((1, n, 6), dtype=torch.float32, device=dev)
    det[0, :, :4] = pred_boxes.float()
    det[0, :, 5] = pred_cls.float()
    gt = torch.cat([gt_cls.float().view(1, m, 1), gt_boxes.float().view(1, m, 4)], 2).contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    ng = torch.tensor([m], dtype=torch.int32, device=dev)
    return match_predictions_batched(det, cnt, gt, ng)[0].cpu().numpy().astype(bool)


# ---- host-side AP arithmetic.  These three functions are a numerical RECIPE, not a design: validation mAP has to come out equal to
# the reference's to the last bit (tests/test_validator.py compares at 1e-7 against goldens produced by the imported reference), and
# that fixes the operations and their order - the box filter's edge padding, `np.interp` on the NEGATED confidences (descending x),
# the precision envelope by a reversed running maximum and the 101-point trapezoid.  They restate utils/metrics.py:612-617, 708-737
# and 740-835 (plotting, names and the per-class dict output left out); everything on the GPU side of validation (`match_predictions`,
# the statistics gather) is this repository's own.
def smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    """Box filter of fraction f (utils/metrics.py:612-617)."""
    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    return np.convolve(np.concatenate((pad * y[0], y, pad * y[-1]), 0), np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    """101-point interpolated AP (utils/metrics.py:708-737)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return np.trapezoid(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps: float = 1e-16):
    """Per-class precision / recall / F1 at the max-F1 confidence and AP at the 10 IoU thresholds
    (utils/metrics.py:740-835 without plotting). Returns (p, r, f1, ap, unique_classes)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    x = np.linspace(0, 1, 1000)
    ap = np.zeros((classes.shape[0], tp.shape[1]))
    p_curve, r_curve = np.zeros((classes.shape[0], 1000)), np.zeros((classes.shape[0], 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if sel.sum() == 0 or n_labels[ci] == 0:
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(0), tp[sel].cumsum(0)
        recall = tpc / (n_labels[ci] + eps)
        precision = tpc / (tpc + fpc)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()
    return p_curve[:, best], r_curve[:, best], f1_curve[:, best], ap, classes.astype(int)


def mean_results(p, r, ap):
    """(mean precision, mean recall, mAP50, mAP50-95) as `Metric.mean_results` reports them."""
    if not len(ap):
        return 0.0, 0.0, 0.0, 0.0
    return float(p.mean()), float(r.mean()), float(ap[:, 0].mean()), float(ap.mean())
