"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference post-processing: `xywh2xyxy` (utils/ops.py:268-284), `box_iou`
(utils/metrics.py:54-74), `TorchNMS.nms` (utils/nms.py:239-296), `non_max_suppression` (utils/nms.py:13-166; the
non-rotated, non-end2end, no-apriori-labels branches) and `RTDETRPredictor.postprocess` semantics
(models/rtdetr/predict.py:35-74).  The reference's wall-clock abort (nms.py:81,162-164) is deliberately NOT restated.
Pinned by tests/golden/nms_cases.npz (outputs of the imported reference on the same inputs).
"""

from __future__ import annotations

import torch


def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:268-284: xy = c -/+ wh/2."""
    y = torch.empty_like(x)
    xy, wh = x[..., :2], x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def box_iou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """(N,4),(M,4) xyxy -> (N,M) IoU with eps in the denominator (utils/metrics.py:54-74)."""
    (a1, a2), (b1, b2) = box1.float().unsqueeze(1).chunk(2, 2), box2.float().unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def greedy_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy hard-NMS; survivor iff IoU <= thr with every kept higher-scored box; IoU has no eps
    (utils/nms.py:239-296).  Stable descending sort: ties keep ascending index order."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = scores.argsort(dim=0, descending=True, stable=True)
    keep = []
    while order.numel() > 0:
        i = order[0]
        keep.append(int(i))
        if order.numel() == 1:
            break
        rest = order[1:]
        w = (torch.minimum(x2[i], x2[rest]) - torch.maximum(x1[i], x1[rest])).clamp_(min=0)
        h = (torch.minimum(y2[i], y2[rest]) - torch.maximum(y1[i], y1[rest])).clamp_(min=0)
        inter = w * h
        if inter.sum() == 0:
            order = rest
            continue
        iou = inter / (areas[i] + areas[rest] - inter)
        order = rest[iou <= iou_threshold]
    return torch.tensor(keep, dtype=torch.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, nc=0, max_nms=30000, max_wh=7680, return_idxs=False):
    """(B, 4+nc, A) xywh+scores -> list of (n,6) [x1,y1,x2,y2,conf,cls] (utils/nms.py:13-166)."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    prediction = prediction.clone()
    if classes is not None:
        classes = torch.tensor(classes)
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    mi = 4 + nc
    xc = prediction[:, 4:mi].amax(1) > conf_thres
    xinds = torch.arange(prediction.shape[-1]).expand(bs, -1)[..., None]
    multi_label &= nc > 1
    prediction = prediction.transpose(-1, -2)
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])
    output = [torch.zeros((0, 6))] * bs
    keepi = [torch.zeros((0,), dtype=torch.int64)] * bs
    for xi, (x, xk) in enumerate(zip(prediction, xinds)):
        filt = xc[xi]
        x, xk = x[filt], xk[filt]
        if not x.shape[0]:
            continue
        box, cls = x[:, :4], x[:, 4:mi]
        if multi_label:
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float()), 1)
            xk = xk[i]
        else:
            conf, j = cls.max(1, keepdim=True)
            filt = conf.view(-1) > conf_thres
            x = torch.cat((box, conf, j.float()), 1)[filt]
            xk = xk[filt]
        if classes is not None:
            filt = (x[:, 5:6] == classes).any(1)
            x, xk = x[filt], xk[filt]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            filt = x[:, 4].argsort(descending=True, stable=True)[:max_nms]
            x, xk = x[filt], xk[filt]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        i = greedy_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        output[xi] = x[i]
        keepi[xi] = xk[i].view(-1)
    return (output, keepi) if return_idxs else output


def rtdetr_postprocess(preds: torch.Tensor, conf: float = 0.25, max_det: int = 300, imgsz=(640, 640),
                       classes=None):
    """(B,300,4+nc) normalised cxcywh + scores -> list of (n,6) xyxy(px),score,cls sorted by score
    (models/rtdetr/predict.py:35-74; inputs are already letterboxed squares so ow=oh=imgsz)."""
    nd = preds.shape[-1]
    bboxes, scores = preds.split((4, nd - 4), dim=-1)
    out = []
    for bbox, score in zip(bboxes, scores):
        bbox = xywh2xyxy(bbox)
        max_score, cls = score.max(-1, keepdim=True)
        idx = max_score.squeeze(-1) > conf
        if classes is not None:
            idx = (cls == torch.tensor(classes)).any(1) & idx
        pred = torch.cat([bbox, max_score, cls], dim=-1)[idx]
        pred = pred[pred[:, 4].argsort(descending=True, stable=True)][:max_det]
        pred = pred.clone()
        pred[..., [0, 2]] *= imgsz[1]
        pred[..., [1, 3]] *= imgsz[0]
        out.append(pred)
    return out
