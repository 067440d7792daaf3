"""Detection validation metrics with the reference's function names (ultralytics/utils/metrics.py,
engine/validator.py:267-308, models/yolo/detect/val.py:274-288).

The pairwise IoU matrix is the only device work (`upa_box_iou`); the greedy IoU matching and the AP integration are
small host-side numpy code in the reference as well (it calls `.cpu().numpy()` first), and are written here against
numpy the same way.  `DetMetrics.process()`'s numbers (P, R, mAP50, mAP50-95) are reproduced by `ap_per_class` +
`mean_results`.
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


def match_predictions(pred_classes: torch.Tensor, true_classes: torch.Tensor, iou: torch.Tensor, iouv=IOUV) -> np.ndarray:
    """Greedy one-to-one matching of predictions to labels at 10 IoU thresholds -> (N, 10) bool
    (engine/validator.py:267-308, default non-scipy branch)."""
    pc = pred_classes.detach().cpu().numpy()
    tc = true_classes.detach().cpu().numpy()
    m = iou.detach().cpu().numpy() * (tc[:, None] == pc[None, :])
    correct = np.zeros((pc.shape[0], len(iouv)), dtype=bool)
    for k, thr in enumerate([float(t) for t in iouv]):
        lab, det = np.nonzero(m >= thr)
        if lab.size:
            pairs = np.stack([lab, det], 1)
            if pairs.shape[0] > 1:
                pairs = pairs[m[pairs[:, 0], pairs[:, 1]].argsort()[::-1]]
                pairs = pairs[np.unique(pairs[:, 1], return_index=True)[1]]
                pairs = pairs[np.unique(pairs[:, 0], return_index=True)[1]]
            correct[pairs[:, 1].astype(int), k] = True
    return correct


def process_batch(pred_boxes: torch.Tensor, pred_cls: torch.Tensor, gt_boxes: torch.Tensor, gt_cls: torch.Tensor) -> np.ndarray:
    """True-positive matrix of one image (detect/val.py:274-288)."""
    if gt_cls.shape[0] == 0 or pred_cls.shape[0] == 0:
        return np.zeros((pred_cls.shape[0], len(IOUV)), dtype=bool)
    return match_predictions(pred_cls, gt_cls, box_iou(gt_boxes, pred_boxes))


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
