"""Detection-set agreement measures (host side, numpy): how two per-image lists of `[x1, y1, x2, y2, score, cls]` rows compare.

Used by `bench.py` to print the "box/cls match vs CPU ref" half of BASELINE.json's metric next to the throughput, and by the
`-m gpu` parity tests (`tests/hip_utils.py` re-exports these).  Nothing here touches the GPU or the oracle: the callers bring
both sides' rows.  The reference's own comparison of two runs is `torch.allclose(a.boxes.data, b.boxes.data, atol=0.5)`
(ultralytics/utils/checks.py:780); a row-by-row compare breaks as soon as one row crosses the confidence threshold, so the
measure here is one-to-one matching by IoU among rows of the same class."""

from __future__ import annotations

import numpy as np


def box_iou_np(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """IoU matrix of xyxy boxes (n,4) x (m,4)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = np.clip(rb - lt, 0, None).prod(2)
    return inter / (area_a[:, None] + area_b[None, :] - inter + 1e-12)


def match_detections(mine: np.ndarray, ref: np.ndarray, iou_thr: float = 0.9):
    """One-to-one matching of two detection sets of ONE image (rows [x1,y1,x2,y2,score,cls]): pairs are taken greedily by
    decreasing IoU among pairs of the SAME class with IoU >= iou_thr.  Returns (pairs [(i_mine, j_ref)], iou of pairs)."""
    if mine.shape[0] == 0 or ref.shape[0] == 0:
        return [], np.zeros(0)
    iou = box_iou_np(mine[:, :4], ref[:, :4])
    iou = np.where(mine[:, 5][:, None] == ref[:, 5][None, :], iou, 0.0)
    order = np.dstack(np.unravel_index(np.argsort(-iou, axis=None), iou.shape))[0]
    used_i, used_j, pairs, vals = set(), set(), [], []
    for i, j in order:
        if iou[i, j] < iou_thr:
            break
        if i in used_i or j in used_j:
            continue
        used_i.add(i)
        used_j.add(j)
        pairs.append((int(i), int(j)))
        vals.append(iou[i, j])
    return pairs, np.asarray(vals)


def detection_agreement(mine_list, ref_list, iou_thr: float = 0.9):
    """Set agreement of per-image detection lists: recall (matched / reference rows), precision (matched / my rows) and the
    |box| / |score| deviations of the matched pairs, pooled over the images."""
    nm = nr = nmatch = 0
    dbox, dscore = [], []
    for a, b in zip(mine_list, ref_list):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        pairs, _ = match_detections(a, b, iou_thr)
        nm, nr, nmatch = nm + a.shape[0], nr + b.shape[0], nmatch + len(pairs)
        for i, j in pairs:
            dbox.append(np.abs(a[i, :4] - b[j, :4]))
            dscore.append(abs(a[i, 4] - b[j, 4]))
    dbox = np.concatenate(dbox) if dbox else np.zeros(1)
    dscore = np.asarray(dscore) if dscore else np.zeros(1)
    return dict(recall=nmatch / max(nr, 1), precision=nmatch / max(nm, 1), n_ref=nr, n_mine=nm,
                box_p50=float(np.median(dbox)), box_p99=float(np.quantile(dbox, 0.99)), box_max=float(dbox.max()),
                score_p99=float(np.quantile(dscore, 0.99)), score_max=float(dscore.max()))


def split_rows(rows: np.ndarray, counts) -> list:
    out, o = [], 0
    for n in counts:
        out.append(rows[o:o + int(n)])
        o += int(n)
    return out


def rows_identical(mine_list, ref_list, tol: float = 1e-3):
    """Row-by-row comparison (the f32 parity statement): same number of rows per image, same class in every position, boxes and
    scores within `tol`.  Returns (equal, max|box d|, max|score d|) - the maxima over the images whose counts and classes agree."""
    equal, dbox, dscore = True, 0.0, 0.0
    for a, b in zip(mine_list, ref_list):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if a.shape != b.shape or not np.array_equal(a[:, 5], b[:, 5]):
            equal = False
            continue
        if a.shape[0]:
            dbox = max(dbox, float(np.abs(a[:, :4] - b[:, :4]).max()))
            dscore = max(dscore, float(np.abs(a[:, 4] - b[:, 4]).max()))
    return bool(equal and dbox <= tol and dscore <= tol), dbox, dscore


def rows_equivalent(mine_list, ref_list, tol: float = 1e-3, conf_thres: float = 0.25, iou_thres: float = 0.7, iou_tol: float = 1e-3):
    """The f32 parity statement for whole batches: both sides ran the same arithmetic to within `tol`, so every row of one side must
    have a partner on the other (same class, IoU >= 0.99, box and score within `tol`) EXCEPT rows whose presence is decided by a
    threshold within that tolerance - NMS output is a step function of its inputs:
      (a) the row's score is within `tol` of `conf_thres` (nms.py:76: kept on one side, filtered on the other);
      (b) the row's IoU with a same-class row (of either side) is within `iou_tol` of `iou_thres` (nms.py:283: `iou <= thr` keeps);
      (c) the row overlaps (IoU > iou_thres, same class) a row that is itself unmatched and excused: it was suppressed by / survives
          because of a row of kind (a) - (d);
      (d) the row overlaps (IoU > iou_thres, same class) a row whose score is within `tol` of its own: greedy NMS visits candidates
          in score order (nms.py:264), so which of two overlapping near-ties survives is decided below the tolerance.
    Returns a dict: equivalent (bool), matched, unmatched, explained, max |box| / |score| deviation of the matched pairs."""
    matched = unmatched = explained = 0
    dbox = dscore = 0.0
    for a, b in zip(mine_list, ref_list):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        pairs, _ = match_detections(a, b, 0.99)
        matched += len(pairs)
        for i, j in pairs:
            dbox = max(dbox, float(np.abs(a[i, :4] - b[j, :4]).max()))
            dscore = max(dscore, abs(float(a[i, 4] - b[j, 4])))
        ui = sorted(set(range(a.shape[0])) - {i for i, _ in pairs})
        uj = sorted(set(range(b.shape[0])) - {j for _, j in pairs})
        if not ui and not uj:
            continue
        rows = np.concatenate([a, b], 0)                       # every row of either side
        um = np.zeros(rows.shape[0], dtype=bool)
        um[ui] = True
        um[[a.shape[0] + j for j in uj]] = True
        iou = box_iou_np(rows[:, :4], rows[:, :4])
        iou = np.where(rows[:, 5][:, None] == rows[:, 5][None, :], iou, 0.0)
        np.fill_diagonal(iou, 0.0)
        ok = np.zeros(rows.shape[0], dtype=bool)
        ok |= um & (np.abs(rows[:, 4] - conf_thres) <= tol)                                 # (a)
        ok |= um & (np.abs(iou - iou_thres) <= iou_tol).any(1)                              # (b)
        ok |= um & ((iou > iou_thres) & (np.abs(rows[:, 4][:, None] - rows[:, 4][None, :]) <= tol)).any(1)  # (d)
        for _ in range(4):                                                                  # (c), a few rounds of propagation
            ok |= um & ((iou > iou_thres) & (um & ok)[None, :]).any(1)
        unmatched += int(um.sum())
        explained += int((um & ok).sum())
    return dict(equivalent=bool(unmatched == explained and dbox <= tol and dscore <= tol), matched=matched, unmatched=unmatched,
                explained_by_threshold_ties=explained, max_box_abs_px=dbox, max_score_abs=dscore)
