"""Validation loop of the detect path on the HIP kernels, one process per GPU (reference: engine/validator.py:195-260 and
models/yolo/detect/val.py:168-288).

Per batch shard: model forward -> `non_max_suppression` with the validator's defaults (conf 0.001, iou 0.7, multi_label,
max_det 300; validator.py / val.py:108-123) -> true-positive matrices against the batch's labels (`upa_match_predictions`) -
all device side, fixed shapes, no host sync.  End of run: the per-image statistics of every rank are gathered with two
`all_gather_into_tensor` calls (RCCL; the reference pickles Python lists through `dist.gather_object`, val.py:225-240) and
every rank computes the class metrics (`ap_per_class`) on the host exactly as `DetMetrics.process` does.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from ..parallel import dp
from ..utils import metrics as M
from ..utils.nms import nms_raw


class DetectionValidator:
    def __init__(self, model=None, conf: float = 0.001, iou: float = 0.7, max_det: int = 300, max_gt: int = 64):
        self.model, self.conf, self.iou, self.max_det, self.max_gt = model, conf, iou, max_det, max_gt
        self.reset()

    def reset(self):
        self._det, self._cnt, self._tp, self._gt, self._ngt = [], [], [], [], []
        self._loss, self._loss_batches = None, 0

    # ---- validation loss (training-time validate only) --------------------------------------------------------------------
    def add_loss(self, loss_items: torch.Tensor):
        """Accumulate one batch's (box, cls, dfl) loss items (validator.py:222: `self.loss += model.loss(batch, preds)[1]`)."""
        li = loss_items.detach().float()
        self._loss = li.clone() if self._loss is None else self._loss + li
        self._loss_batches += 1

    def reduce_loss(self, dst: int = 0):
        """The accumulated validation loss averaged over the ranks on rank `dst` (validator.py:241-249: `dist.reduce(loss, dst=0,
        op=AVG)`, then divided by the number of batches); every other rank gets None, as the reference returns there."""
        import torch.distributed as dist
        if self._loss is None:
            return None
        loss = dp.reduce_mean_(self._loss.clone(), dst)
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != dst:
            return None
        return loss / max(self._loss_batches, 1)

    # ---- per batch ------------------------------------------------------------------------------------------------------
    def pack_labels(self, labels: dict, batch_size: int, imgsz_hw, device):
        """{"batch_idx", "cls", "bboxes" (normalised xywh)} -> padded (B, max_gt, 5) [cls, x1, y1, x2, y2] pixels + counts:
        the label preparation of DetectionValidator._prepare_batch (val.py:141-166: xywh2xyxy * imgsz)."""
        h, w = imgsz_hw
        bi = labels["batch_idx"].view(-1).long().cpu()
        cls = labels["cls"].view(-1).float().cpu()
        bb = labels["bboxes"].view(-1, 4).float().cpu() * torch.tensor([w, h, w, h], dtype=torch.float32)
        per = [(bi == j).nonzero().view(-1) for j in range(batch_size)]
        cap = max(self.max_gt, max([int(ix.numel()) for ix in per] + [1]))
        gt = torch.zeros(batch_size, cap, 5)
        ngt = torch.zeros(batch_size, dtype=torch.int32)
        for j, ix in enumerate(per):
            k = int(ix.numel())
            if k:
                xy, wh = bb[ix, :2], bb[ix, 2:] / 2
                gt[j, :k, 0] = cls[ix]
                gt[j, :k, 1:3] = xy - wh
                gt[j, :k, 3:5] = xy + wh
            ngt[j] = k
        return gt.to(device), ngt.to(device)

    def update(self, preds, gt: torch.Tensor, ngt: torch.Tensor):
        """preds: the model's eval output (y or (y, raw)); gt / ngt: padded labels of the same images (device tensors)."""
        out, counts, _ = nms_raw(preds, self.conf, self.iou, multi_label=True, max_det=self.max_det)
        self.update_detections(out, counts, gt, ngt)

    def update_detections(self, out: torch.Tensor, counts: torch.Tensor, gt: torch.Tensor, ngt: torch.Tensor):
        """Already post-processed detections (B, max_det, 6) + counts: match them and keep the batch's statistics."""
        tp = M.match_predictions_batched(out, counts, gt, ngt)
        self._det.append(out.clone())
        self._cnt.append(counts.clone())
        self._tp.append(tp)
        self._gt.append(gt[:, :, 0].clone())
        self._ngt.append(ngt.clone())

    def add_batch_stats(self, out: torch.Tensor, counts: torch.Tensor, tp: torch.Tensor, gt: torch.Tensor, ngt: torch.Tensor):
        """Statistics of a batch whose matching already ran (e.g. inside a captured step: `bench.py --workload val`)."""
        self._det.append(out.clone())
        self._cnt.append(counts.clone())
        self._tp.append(tp.clone())
        self._gt.append(gt[:, :, 0].clone())
        self._ngt.append(ngt.clone())

    # ---- end of run -----------------------------------------------------------------------------------------------------
    def local_stats(self):
        """This rank's fixed-shape statistics: (conf|cls|tp rows (I, max_det, 12) f32, counts (I,), gt classes (I, G) f32,
        n_gt (I,)) with I = images seen by this rank."""
        det = torch.cat(self._det, 0)
        tp = torch.cat(self._tp, 0).float()
        rows = torch.cat([det[:, :, 4:6], tp], 2).contiguous()
        g = max(t.shape[1] for t in self._gt)
        gcls = torch.cat([torch.nn.functional.pad(t, (0, g - t.shape[1])) for t in self._gt], 0).contiguous()
        return rows, torch.cat(self._cnt, 0).contiguous(), gcls, torch.cat(self._ngt, 0).contiguous()

    def gather_stats(self):
        """All ranks' statistics on every rank, ordered by rank (= by global image index for contiguous shards)."""
        rows, cnt, gcls, ngt = self.local_stats()
        # shards may differ in image count (uneven split) and in the padded gt width (`pack_labels` grows it to the rank's
        # largest image, COCO has images with > max_gt boxes): the ranks agree on both before anything is gathered
        rows, cnt = dp.gather_ragged(rows, cnt)
        gcls, ngt = dp.gather_ragged(gcls, ngt)
        return rows, cnt, gcls, ngt

    def get_stats(self):
        """{"tp", "conf", "pred_cls", "target_cls"} numpy arrays over all images of all ranks (val.py:212-240) and the class
        metrics `DetMetrics.process` derives from them."""
        rows, cnt, gcls, ngt = (t.cpu() for t in self.gather_stats())
        cnt, ngt = cnt.tolist(), ngt.tolist()
        sel = [rows[i, :cnt[i]] for i in range(len(cnt))]
        allr = torch.cat(sel, 0).numpy() if sel else np.zeros((0, 12), np.float32)
        tcls = np.concatenate([gcls[i, :ngt[i]].numpy() for i in range(len(ngt))]) if ngt else np.zeros(0, np.float32)
        stats = dict(tp=allr[:, 2:].astype(bool), conf=allr[:, 0], pred_cls=allr[:, 1], target_cls=tcls)
        if len(stats["tp"]) and len(tcls):
            p, r, f1, ap, uc = M.ap_per_class(stats["tp"], stats["conf"], stats["pred_cls"], tcls)
            mp, mr, map50, map5095 = M.mean_results(p, r, ap)
        else:
            p = r = f1 = np.zeros(0)
            ap, uc = np.zeros((0, 10)), np.zeros(0, int)
            mp = mr = map50 = map5095 = 0.0
        stats.update(p=p, r=r, f1=f1, ap=ap, classes=uc, mean=(mp, mr, map50, map5095))
        return stats
