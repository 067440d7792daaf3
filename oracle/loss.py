"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the YOLOv8 detection training loss.

Follows the reference: ultralytics/utils/loss.py:21-46 (SlideLoss), :308-360 (DFLoss, BboxLoss), :415-528
(v8DetectionLoss), ultralytics/utils/tal.py:12-316 (TaskAlignedAssigner), :352-390 (make_anchors, dist2bbox, bbox2dist),
ultralytics/utils/metrics.py:77-150 (bbox_iou, CIoU branch). Pinned against the imported reference by
oracle/gen_golden.py (section "train"): loss items, gradients and the updated parameters agree bit for bit on CPU.
Plain torch ops with autograd - this is the checker the HIP training kernels are compared with.
"""

import math

import torch
import torch.nn.functional as F

from .modules import dist2bbox, make_anchors

GAINS = dict(box=7.5, cls=0.5, dfl=1.5)  # cfg/default.yaml hyperparameters read through model.args (loss.py:513-515)


def ciou_xyxy(b1, b2, eps=1e-7):
    """metrics.py:77-150 with xywh=False, CIoU=True; `alpha` is a constant for the gradient (:139-140)."""
    x11, y11, x12, y12 = b1.chunk(4, -1)
    x21, y21, x22, y22 = b2.chunk(4, -1)
    w1, h1 = x12 - x11, y12 - y11 + eps
    w2, h2 = x22 - x21, y22 - y21 + eps
    inter = (x12.minimum(x22) - x11.maximum(x21)).clamp_(0) * (y12.minimum(y22) - y11.maximum(y21)).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = x12.maximum(x22) - x11.minimum(x21)
    ch = y12.maximum(y22) - y11.minimum(y21)
    c2 = cw.pow(2) + ch.pow(2) + eps
    rho2 = ((x21 + x22 - x11 - x12).pow(2) + (y21 + y22 - y11 - y12).pow(2)) / 4
    v = (4 / math.pi ** 2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def bbox2dist(anchor_points, bbox, reg_max):
    """tal.py:379-382."""
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp_(0, reg_max - 0.01)


def slide_bce(pred, true, auto_iou=0.5):
    """loss.py:21-46: BCE-with-logits reweighted by the target score (SlideLoss, reduction 'none')."""
    loss = F.binary_cross_entropy_with_logits(pred, true, reduction="none")
    auto_iou = max(auto_iou, 0.2)
    b1 = true <= auto_iou - 0.1
    b2 = (true > (auto_iou - 0.1)) & (true < auto_iou)
    b3 = true >= auto_iou
    weight = 1.0 * b1 + math.exp(1.0 - auto_iou) * b2 + torch.exp(-(true - 1.0)) * b3
    return loss * weight


def dfl_loss(pred_dist, target, reg_max=16):
    """loss.py:308-326."""
    target = target.clamp_(0, reg_max - 1 - 0.01)
    tl = target.long()
    tr = tl + 1
    wl = tr - target
    wr = 1 - wl
    return (F.cross_entropy(pred_dist, tl.view(-1), reduction="none").view(tl.shape) * wl
            + F.cross_entropy(pred_dist, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)


class TaskAlignedAssigner:
    """tal.py:12-316 (topk=10, alpha=0.5, beta=6.0 as constructed at loss.py:441)."""

    def __init__(self, topk=10, num_classes=80, alpha=0.5, beta=6.0, eps=1e-9):
        self.topk, self.nc, self.alpha, self.beta, self.eps = topk, num_classes, alpha, beta, eps

    @torch.no_grad()
    def __call__(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        bs, n_max = pd_scores.shape[0], gt_bboxes.shape[1]
        if n_max == 0:  # tal.py:69-76
            return (torch.full_like(pd_scores[..., 0], self.nc), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                    torch.zeros_like(pd_scores[..., 0]), torch.zeros_like(pd_scores[..., 0]))
        na = pd_bboxes.shape[-2]
        # anchors whose centre lies inside the gt box (tal.py:271-291)
        lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
        deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), dim=2).view(bs, n_max, na, -1)
        mask_in_gts = deltas.amin(3).gt_(1e-9)
        # alignment metric (tal.py:146-178)
        m = (mask_in_gts * mask_gt).bool()
        overlaps = torch.zeros([bs, n_max, na], dtype=pd_bboxes.dtype)
        bbox_scores = torch.zeros([bs, n_max, na], dtype=pd_scores.dtype)
        ind0 = torch.arange(bs).view(-1, 1).expand(-1, n_max)
        ind1 = gt_labels.squeeze(-1).long()
        bbox_scores[m] = pd_scores[ind0, :, ind1][m]
        pd_b = pd_bboxes.unsqueeze(1).expand(-1, n_max, -1, -1)[m]
        gt_b = gt_bboxes.unsqueeze(2).expand(-1, -1, na, -1)[m]
        overlaps[m] = ciou_xyxy(gt_b, pd_b).squeeze(-1).clamp_(0)
        align = bbox_scores.pow(self.alpha) * overlaps.pow(self.beta)
        # top-k anchors per gt (tal.py:193-222)
        topk_mask = mask_gt.expand(-1, -1, self.topk).bool()
        _, topk_idxs = torch.topk(align, self.topk, dim=-1, largest=True)
        topk_idxs.masked_fill_(~topk_mask, 0)
        count = torch.zeros(align.shape, dtype=torch.int8)
        ones = torch.ones_like(topk_idxs[:, :, :1], dtype=torch.int8)
        for k in range(self.topk):
            count.scatter_add_(-1, topk_idxs[:, :, k:k + 1], ones)
        count.masked_fill_(count > 1, 0)
        mask_pos = count.to(align.dtype) * mask_in_gts * mask_gt
        # an anchor claimed by several gts goes to the one with the highest overlap (tal.py:293-316)
        fg = mask_pos.sum(-2)
        if fg.max() > 1:
            multi = (fg.unsqueeze(1) > 1).expand(-1, n_max, -1)
            is_max = torch.zeros(mask_pos.shape, dtype=mask_pos.dtype)
            is_max.scatter_(1, overlaps.argmax(1).unsqueeze(1), 1)
            mask_pos = torch.where(multi, is_max, mask_pos).float()
            fg = mask_pos.sum(-2)
        target_gt_idx = mask_pos.argmax(-2)
        # targets (tal.py:224-268)
        flat_idx = target_gt_idx + torch.arange(bs)[..., None] * n_max
        target_labels = gt_labels.long().flatten()[flat_idx].clamp_(0)
        target_bboxes = gt_bboxes.view(-1, 4)[flat_idx]
        target_scores = torch.zeros((bs, na, self.nc), dtype=torch.int64)
        target_scores.scatter_(2, target_labels.unsqueeze(-1), 1)
        target_scores = torch.where(fg[:, :, None].repeat(1, 1, self.nc) > 0, target_scores, 0)
        # normalise by the best alignment / overlap of each gt (tal.py:118-124)
        align = align * mask_pos
        pos_align = align.amax(dim=-1, keepdim=True)
        pos_overlaps = (overlaps * mask_pos).amax(dim=-1, keepdim=True)
        norm = (align * pos_overlaps / (pos_align + self.eps)).amax(-2).unsqueeze(-1)
        return target_labels, target_bboxes, target_scores * norm, fg.bool(), target_gt_idx


def preprocess_targets(batch_idx, cls, bboxes, batch_size, scale):
    """loss.py:445-461: (n,) image index + (n,) class + (n,4) normalised xywh -> (B, max_n, 5) [cls, xyxy pixels]."""
    targets = torch.cat((batch_idx.view(-1, 1), cls.view(-1, 1), bboxes), 1)
    if targets.shape[0] == 0:
        return torch.zeros(batch_size, 0, 5)
    i = targets[:, 0]
    _, counts = i.unique(return_counts=True)
    out = torch.zeros(batch_size, int(counts.max()), 5)
    for j in range(batch_size):
        sel = i == j
        if n := int(sel.sum()):
            out[j, :n] = targets[sel, 1:]
    xywh = out[..., 1:5].mul_(scale)
    xy, wh = xywh[..., :2], xywh[..., 2:] / 2
    out[..., 1:5] = torch.cat((xy - wh, xy + wh), -1)
    return out


def v8_detection_loss(feats, batch, strides, nc=80, reg_max=16, gains=GAINS, tal_topk=10):
    """loss.py:471-528. feats: list of (B, 4*reg_max+nc, H, W) head maps (train-mode Detect output).
    Returns (loss (3,) * batch_size [box, cls, dfl], detached loss items (3,))."""
    no = nc + reg_max * 4
    b = feats[0].shape[0]
    pred_distri, pred_scores = torch.cat([xi.view(b, no, -1) for xi in feats], 2).split((reg_max * 4, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()
    dtype = pred_scores.dtype
    imgsz = torch.tensor(feats[0].shape[2:], dtype=dtype) * strides[0]
    anchor_points, stride_tensor = make_anchors(feats, strides, 0.5)
    targets = preprocess_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], b, imgsz[[1, 0, 1, 0]])
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = gt_bboxes.sum(2, keepdim=True).gt_(0.0)
    proj = torch.arange(reg_max, dtype=torch.float)
    pred_ltrb = pred_distri.view(b, -1, 4, reg_max).softmax(3).matmul(proj.type(dtype))  # loss.py:463-469
    pred_bboxes = dist2bbox(pred_ltrb, anchor_points, xywh=False)
    assigner = TaskAlignedAssigner(topk=tal_topk, num_classes=nc, alpha=0.5, beta=6.0)
    _, target_bboxes, target_scores, fg_mask, _ = assigner(
        pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor).type(gt_bboxes.dtype),
        anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt)
    tss = max(target_scores.sum(), 1)
    loss = torch.zeros(3)
    loss[1] = slide_bce(pred_scores, target_scores.to(dtype)).sum() / tss
    if fg_mask.sum():
        tb = target_bboxes / stride_tensor
        weight = target_scores.sum(-1)[fg_mask].unsqueeze(-1)
        iou = ciou_xyxy(pred_bboxes[fg_mask], tb[fg_mask])
        loss[0] = ((1.0 - iou) * weight).sum() / tss
        t_ltrb = bbox2dist(anchor_points, tb, reg_max - 1)
        ld = dfl_loss(pred_distri[fg_mask].view(-1, reg_max), t_ltrb[fg_mask], reg_max) * weight
        loss[2] = ld.sum() / tss
    loss[0] *= gains["box"]
    loss[1] *= gains["cls"]
    loss[2] *= gains["dfl"]
    return loss * b, loss.detach()
