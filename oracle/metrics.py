"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference's detection-validation arithmetic: `match_predictions`
(engine/validator.py:267-308, the non-scipy branch), `DetectionValidator._process_batch` (models/yolo/detect/val.py:274-288),
`smooth` (utils/metrics.py:612-617), `compute_ap` (:708-737, 101-point interpolation), `ap_per_class` (:740-835) and the
`Metric` means (mp, mr, map50, map).  Pinned by tests/golden/map_yolov8n.npz (outputs of the imported reference).
"""

from __future__ import annotations

import numpy as np
import torch

from .nms import box_iou

IOUV = torch.linspace(0.5, 0.95, 10)  # models/yolo/detect/val.py:59


def match_predictions(pred_classes, true_classes, iou, iouv=IOUV):
    """(N,) pred classes, (M,) true classes, (M,N) IoU -> (N,10) bool (validator.py:267-308)."""
    correct = np.zeros((pred_classes.shape[0], iouv.shape[0])).astype(bool)
    correct_class = true_classes[:, None] == pred_classes
    iou = (iou * correct_class).cpu().numpy()
    for i, threshold in enumerate(iouv.cpu().tolist()):
        matches = np.array(np.nonzero(iou >= threshold)).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return torch.tensor(correct, dtype=torch.bool)


def process_batch(pred_boxes, pred_cls, gt_boxes, gt_cls):
    """val.py:274-288."""
    if gt_cls.shape[0] == 0 or pred_cls.shape[0] == 0:
        return np.zeros((pred_cls.shape[0], IOUV.shape[0]), dtype=bool)
    return match_predictions(pred_cls, gt_cls, box_iou(gt_boxes, pred_boxes)).numpy()


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return np.trapezoid(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """utils/metrics.py:740-835 (no plotting). Returns (p, r, f1, ap, unique_classes)."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    x = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l, n_p = nt[ci], i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    i = smooth(f1_curve.mean(0), 0.1).argmax()
    return p_curve[:, i], r_curve[:, i], f1_curve[:, i], ap, unique_classes.astype(int)


def mean_results(p, r, ap):
    """Metric.mean_results: (mp, mr, map50, map) (utils/metrics.py Metric)."""
    return float(p.mean()) if len(p) else 0.0, float(r.mean()) if len(r) else 0.0, \
        float(ap[:, 0].mean()) if len(ap) else 0.0, float(ap.mean()) if len(ap) else 0.0
