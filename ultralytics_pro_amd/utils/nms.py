"""non_max_suppression with the reference signature (ultralytics/utils/nms.py:13-29), executed by `upa_nms_batched`.

One call = memset + 3 kernels for the whole batch, no host synchronisation inside; the list-of-tensors return value
of the reference is produced from the fixed-shape device outputs only when the caller asks for it.
"""

from __future__ import annotations

import torch

from .. import _lib as L
from ..engine import runtime as R


def nms_raw(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
            max_det=300, nc=0, max_nms=30000, max_wh=7680, key=None):
    """Device-side NMS. Returns (out (B,max_det,6) f32, counts (B,) int32, keep_idx (B,max_det) int32) without syncing."""
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    L.require_gpu(prediction, "non_max_suppression")
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        raise L.UpaError("non_max_suppression expects the contiguous float32 (B, 4+nc, A) Detect output")
    b, ch, a = prediction.shape
    nc = nc or (ch - 4)
    if ch != 4 + nc:
        raise L.UpaError("extra mask channels (segmentation) are outside the hot-path scope")
    dev = prediction.device
    lib = L.lib()
    ws_bytes = lib.upa_nms_workspace_bytes(b, nc, a, int(bool(multi_label)), max_nms)
    ws = R.alloc_plain((ws_bytes,), torch.uint8, dev, key=(key, "nms_ws"))
    out = R.alloc_plain((b, max_det, 6), torch.float32, dev, key=(key, "nms_out"))
    counts = R.alloc_plain((b,), torch.int32, dev, key=(key, "nms_counts"))
    keep = R.alloc_plain((b, max_det), torch.int32, dev, key=(key, "nms_keep"))
    cmask = None
    if classes is not None:
        m = torch.zeros(nc, dtype=torch.uint8)
        m[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        cmask = m.to(dev)
    # Detect's fused class tails may have written the NMS key of every anchor's best class next to the scores
    # (head.Detect.nms_keys): single-label NMS compacts those (B, A) keys instead of re-reading the (B, nc, A) scores
    hot = getattr(prediction, "_upa_hot", None)
    if getattr(prediction, "_upa_keys_only", False) and (hot is None or (multi_label and nc > 1)):
        raise L.UpaError("this Detect output was produced in keys-only mode (Detect.scores_out = False): its class rows were not written, "
                         "only single-label NMS (which reads the best-class keys) can consume it - set Detect.scores_out = True for "
                         "multi-label NMS / validation")
    if hot is not None and tuple(hot.shape) == (b, a) and hot.device == dev and not (multi_label and nc > 1):
        L.check(lib.upa_nms_batched_hot(prediction.data_ptr(), b, nc, a, float(conf_thres), float(iou_thres),
                                        int(bool(multi_label)), int(bool(agnostic)), None if cmask is None else cmask.data_ptr(),
                                        int(max_det), int(max_nms), float(max_wh), out.data_ptr(), counts.data_ptr(),
                                        keep.data_ptr(), ws.data_ptr(), ws_bytes, hot.data_ptr(), L.current_stream(dev)),
                "nms_batched_hot")
        return out, counts, keep
    L.check(lib.upa_nms_batched_opts(prediction.data_ptr(), b, nc, a, float(conf_thres), float(iou_thres), int(bool(multi_label)),
                                     int(bool(agnostic)), None if cmask is None else cmask.data_ptr(), int(max_det),
                                     int(max_nms), float(max_wh), out.data_ptr(), counts.data_ptr(), keep.data_ptr(),
                                     ws.data_ptr(), ws_bytes, R.opts_ptr(), L.current_stream(dev)), "nms_batched")
    return out, counts, keep


def non_max_suppression(prediction, conf_thres: float = 0.25, iou_thres: float = 0.45, classes=None,
                        agnostic: bool = False, multi_label: bool = False, labels=(), max_det: int = 300, nc: int = 0,
                        max_time_img: float = 0.05, max_nms: int = 30000, max_wh: int = 7680, rotated: bool = False,
                        end2end: bool = False, return_idxs: bool = False):
    """Reference-compatible wrapper: list of (n, 6) tensors [x1, y1, x2, y2, conf, cls] per image (nms.py:13-166).
    `max_time_img` is accepted and ignored: the reference's wall-clock abort (nms.py:81,162-164) is not emulated."""
    if labels or rotated:
        raise L.UpaError("apriori labels / rotated boxes are outside the hot-path scope")
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if prediction.shape[-1] == 6 or end2end:
        raise L.UpaError("end2end (NMS-free) heads are outside the hot-path scope")
    out, counts, keep = nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, nc, max_nms,
                                max_wh)
    n = counts.tolist()  # the only device->host sync, outside the kernels
    res = [out[i, : n[i]] for i in range(len(n))]
    if return_idxs:
        return res, [keep[i, : n[i]].long() for i in range(len(n))]
    return res


def rtdetr_postprocess_raw(preds: torch.Tensor, conf: float = 0.25, max_det: int = 300, imgsz=(640, 640), classes=None,
                           orig_shapes=None, key=None):
    """Device-side RTDETRPredictor.postprocess (models/rtdetr/predict.py:35-74): (out (B, max_det, 6) f32, counts (B,)
    int32) without syncing.  `imgsz` = (h, w) every box is scaled to, or `orig_shapes` = per-image (h, w) list."""
    if isinstance(preds, (list, tuple)):
        preds = preds[0]
    L.require_gpu(preds, "rtdetr_postprocess")
    if preds.dtype != torch.float32 or not preds.is_contiguous() or preds.dim() != 3:
        raise L.UpaError("rtdetr_postprocess expects the contiguous float32 (B, queries, 4+nc) RTDETRDecoder output")
    b, q, nd = preds.shape
    nc, dev = nd - 4, preds.device
    out = R.alloc_plain((b, max_det, 6), torch.float32, dev, key=(key, "rtdetr_out"))
    counts = R.alloc_plain((b,), torch.int32, dev, key=(key, "rtdetr_counts"))
    cmask = wh = None
    if classes is not None:
        m = torch.zeros(nc, dtype=torch.uint8)
        m[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        cmask = m.to(dev)
    if orig_shapes is not None:
        wh = torch.tensor([[float(s[1]), float(s[0])] for s in orig_shapes], dtype=torch.float32).to(dev)
    L.check(L.lib().upa_rtdetr_postprocess(preds.data_ptr(), b, q, nc, float(conf), None if cmask is None else cmask.data_ptr(),
                                           int(max_det), None if wh is None else wh.data_ptr(), float(imgsz[1]), float(imgsz[0]),
                                           out.data_ptr(), counts.data_ptr(), L.current_stream(dev)), "rtdetr_postprocess")
    return out, counts


def rtdetr_postprocess(preds, conf: float = 0.25, max_det: int = 300, imgsz=(640, 640), classes=None, orig_shapes=None):
    """List of (n, 6) tensors [x1, y1, x2, y2, score, cls] per image, sorted by score (models/rtdetr/predict.py:35-74)."""
    out, counts = rtdetr_postprocess_raw(preds, conf, max_det, imgsz, classes, orig_shapes)
    n = counts.tolist()
    return [out[i, : n[i]] for i in range(len(n))]
