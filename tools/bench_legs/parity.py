"""bench.py's `parity` object: the GPU output against the oracle's on the same batch, in the same run.  The oracle's tensors come from the file
the CPU-baseline child wrote (`--parity-out`); nothing of oracle/ is imported here."""
from __future__ import annotations

import os

import torch


def gpu_parity(args, dev, ppath, model, x0, results, pb):
    """GPU output vs the oracle's on the SAME batch (rank 0's first resident batch = procedural images 0 .. pb - 1), in the same run:
      f32  - the parity mode (`--dtype f32`: exact-f32 MFMA): one extra forward + NMS of an f32 copy of the model after the timed
             region; max |box| / |score| over every anchor of the head output, and the detections row by row (north_star: 1e-3);
      bf16 - the mode the throughput is quoted in: the detections the TIMED region itself produced for that batch (the static result
             of compiled copy 0) as a set against the oracle's (one-to-one same-class matches at IoU >= 0.9 / 0.5), plus the head
             output of one eager forward.
    The oracle's tensors come from the CPU-baseline child (`--parity-out`); nothing of oracle/ is imported here."""
    import numpy as np

    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import parity as PA
    from ultralytics_pro_amd.utils import procedural as P
    from ultralytics_pro_amd.utils.nms import non_max_suppression

    if not os.path.exists(ppath):
        return {"error": "the CPU leg produced no oracle output (cut at its wall-clock limit?)"}
    ref = torch.load(ppath)
    y_ref = ref["y"]
    ref_rows = PA.split_rows(ref["rows"].numpy(), ref["n"])
    out = {"images": int(y_ref.shape[0]), "oracle": f"oracle (CPU f32, fused eval, {ref['threads']} threads) on procedural images 0..{pb - 1}",
           "reference_detections": int(sum(ref["n"]))}
    if "rtdetr" in args.model:
        return _gpu_parity_rtdetr(args, dev, y_ref, model, x0, out, pb, ref)
    with torch.no_grad():
        mf = DetectionModel(args.model + ".yaml")
        P.apply_procedural_weights(mf)
        mf = mf.to(dev).eval()
        mf.set_compute_dtype(torch.float32)
        x32 = P.synthetic_images(pb, first=0).to(dev)
        yf = mf(x32)[0]
        det = [d.cpu().numpy() for d in non_max_suppression(yf, 0.25, 0.7, max_det=300)]
        d = (yf.cpu() - y_ref).abs()
        eq, _, _ = PA.rows_identical(det, ref_rows, 1e-3)
        rq = PA.rows_equivalent(det, ref_rows, 1e-3, 0.25, 0.7)
        out["f32"] = {"max_box_abs_px": float(d[:, :4].max()), "max_score_abs": float(d[:, 4:].max()),
                      "detections": int(sum(len(r) for r in det)), "rows_equal": eq,
                      # rows whose presence a threshold decides within the tolerance (score within 1e-3 of conf, IoU within 1e-3 of
                      # iou_thres, or overlapping such a row) are counted and excused; every other row must have its partner
                      "rows": rq, "tolerance": 1e-3,
                      "within_tolerance": bool(d[:, :4].max() <= 1e-3 and d[:, 4:].max() <= 1e-3 and rq["equivalent"])}
        if args.model == "yolov8s":
            # the reference's OWN f32 output on this model moves by 2.2e-3 px between 8 and 1 CPU threads and sits 1.8e-3 .. 2.9e-3 px from
            # its float64 run (tools/ref_noise_floor.py; tests/test_oracle_golden.py): 1e-3 px is below its reproducibility there
            out["f32"]["box_tolerance_note"] = ("yolov8s: the reference's own f32 noise floor is 2.2e-3 - 2.9e-3 px (8 vs 1 threads, vs float64); "
                                                "the tests gate its boxes at 3e-3 px, scores at 1e-3")
            out["f32"]["within_reference_noise_floor"] = bool(d[:, :4].max() <= 3e-3 and d[:, 4:].max() <= 1e-3 and rq["equivalent"])
        del mf, yf
        if args.dtype == "bf16":
            mine = []
            for (o_, c_, _) in results:
                oc, cc = o_.cpu().numpy(), c_.cpu().tolist()
                mine += [oc[i, :int(cc[i])] for i in range(len(cc))]
            a9, a5 = PA.detection_agreement(mine, ref_rows, 0.9), PA.detection_agreement(mine, ref_rows, 0.5)
            det_ = model.model[-1]
            so_ = getattr(det_, "scores_out", True)
            if hasattr(det_, "scores_out"):
                det_.scores_out = True  # the head comparison reads the class rows
            try:
                yb = model(x0)[0].float().cpu()
            finally:
                if hasattr(det_, "scores_out"):
                    det_.scores_out = so_
            db = (yb - y_ref).abs()
            out["bf16"] = {"detections": a9["n_mine"], "recall_iou90": round(a9["recall"], 4), "precision_iou90": round(a9["precision"], 4),
                           "recall_iou50": round(a5["recall"], 4), "precision_iou50": round(a5["precision"], 4),
                           "matched_box_p99_px": round(a9["box_p99"], 4), "matched_box_max_px": round(a9["box_max"], 4),
                           "matched_score_max": round(a9["score_max"], 5),
                           "head_box_p99_px": float(np.quantile(db[:, :4].numpy().ravel()[::7], 0.99)), "head_box_max_px": float(db[:, :4].max()),
                           "head_score_max": float(db[:, 4:].max()),
                           "source": "detections: the timed region's own result for this batch (compiled copy 0); head: one eager forward"}
    return out


def _gpu_parity_rtdetr(args, dev, y_ref, model, x0, out, pb, ref=None):
    """Config 5: the (B, 300, 4 + nc) decoder output against the oracle's as SETS of rows per image (the 300 queries are the top-300
    tokens by encoder score, head.py:2175: two implementations may order near-ties differently, and the bf16 mode may pick other
    tokens near the cut).  f32: the largest distance of an oracle row to its own partner (1e-3 = north_star's tolerance); bf16: the
    fraction of oracle rows reproduced within the reference's AMP tolerance (0.5 px of 640, utils/checks.py:780) and 0.01 per score."""
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import procedural as P

    def sets(y):
        worst, frac = 0.0, []
        for i in range(y.shape[0]):
            dist = (y[i][:, None, :] - y_ref[i][None, :, :]).abs().amax(2)
            worst = max(worst, float(dist.min(0).values.max()))
            db = (y[i][:, None, :4] - y_ref[i][None, :, :4]).abs().amax(2)
            j = db.argmin(0)
            sc = (y[i][j, 4:] - y_ref[i][:, 4:]).abs().amax(1)
            frac.append(float(((db.min(0).values <= 0.5 / 640) & (sc <= 0.01)).float().mean()))
        return worst, frac
    with torch.no_grad():
        mf = DetectionModel(args.model + ".yaml")
        P.apply_procedural_weights(mf)
        mf = mf.to(dev).eval()
        mf.set_compute_dtype(torch.float32)
        yf = mf(P.synthetic_images(pb, first=0).to(dev))[0].float().cpu()
        worst, _ = sets(yf)
        out["f32"] = {"worst_row_to_partner": worst, "tolerance": 1e-3, "within_tolerance": bool(worst <= 1e-3),
                      "compared": "decoder output rows as sets per image (normalised boxes, class scores)"}
        del mf, yf
        if args.dtype == "bf16":
            yb = model(x0)[0].float().cpu()
            _, frac = sets(yb)
            out["bf16"] = {"oracle_rows_reproduced_mean": round(sum(frac) / len(frac), 4), "oracle_rows_reproduced_min": round(min(frac), 4),
                           "within": "0.5 px of 640 on the box and 0.01 on every class score",
                           "note": "with its OWN query selection: random-weight encoder scores are nearly flat, so the bf16 mode selects other "
                                   "top-300 tokens near the cut and this fraction measures the selection's sensitivity, not the arithmetic; the "
                                   "arithmetic is in `encoder_head_every_token` and `decoder_on_oracle_queries` (same two comparisons, gated on the "
                                   "smooth weight family, in tests/test_hip_e2e.py)"}
            if ref is not None and "topk" in ref:
                out["bf16"].update(_rtdetr_bf16_pins(model, x0, y_ref, ref))
    return out


def _rtdetr_bf16_pins(model, x0, y_ref, ref):
    """The bf16 mode of config 5 against the oracle WITHOUT the chaos of the top-300 selection (round-5 review item 1): (a) in front of it,
    class probabilities and encoder boxes of every one of the B x 8400 tokens; (b) behind it, the decoder run on the ORACLE'S query indices
    (`RTDETRDecoder.query_override`), row by row.  Fractions inside the reference's fp16-AMP tolerance (0.5 px of 640, 0.01) and inside that
    tolerance with the box part scaled by bf16's three missing significand bits (4 px)."""
    from ultralytics_pro_amd.nn.modules import rtdetr as RT
    head = model.model[-1]
    res = {}
    with torch.no_grad():
        head.taps = {}
        try:
            model(x0)
            t = head.taps
            st, bs = t["static"], t["bs"]
            prob = head.level_major_to_image(t["enc_scores"], st, bs).float().sigmoid().cpu()
            saved = RT._LINEAR_BF16[0]
            RT._LINEAR_BF16[0] = bool(head.linear_bf16)
            try:
                delta = head.enc_bbox_head(t["features"], key="all_tokens")
            finally:
                RT._LINEAR_BF16[0] = saved
            box = (head.level_major_to_image(delta, st, bs).float().cpu() + st["anchors"].cpu().view(1, -1, 4)).sigmoid()
        finally:
            head.taps = None
        valid = ref["enc_valid"]
        dp = (prob - ref["enc_prob"]).abs().amax(2)[:, valid]
        db = ((box - ref["enc_box"]).abs().amax(2) * 640)[:, valid]
        res["encoder_head_every_token"] = {
            "tokens": int(dp.numel()), "inside_0.5px_0.01": round(float(((db <= 0.5) & (dp <= 0.01)).float().mean()), 4),
            "inside_4px_0.01": round(float(((db <= 4.0) & (dp <= 0.01)).float().mean()), 4),
            "class_prob_max_abs": round(float(dp.max()), 5), "box_p99_px": round(float(db.flatten()[::3].quantile(0.99)), 3),
            "box_max_px": round(float(db.max()), 3)}
        head.query_override = ref["topk"]
        try:
            yq = model(x0)[0].float().cpu()
        finally:
            head.query_override = None
        rb = (yq[..., :4] - y_ref[..., :4]).abs().amax(2) * 640
        rs = (yq[..., 4:] - y_ref[..., 4:]).abs().amax(2)
        res["decoder_on_oracle_queries"] = {
            "rows": int(rb.numel()), "inside_0.5px_0.01": round(float(((rb <= 0.5) & (rs <= 0.01)).float().mean()), 4),
            "inside_4px_0.01": round(float(((rb <= 4.0) & (rs <= 0.01)).float().mean()), 4),
            "box_p50_px": round(float(rb.flatten().quantile(0.5)), 3), "box_p99_px": round(float(rb.flatten().quantile(0.99)), 3),
            "score_max_abs": round(float(rs.max()), 5)}
        res["weights"] = ("the bench's default (chaotic) procedural family: its decoder amplifies the backbone's bf16 noise to tens of pixels even "
                          "on fixed queries; on the smooth family (tests) the same comparisons give 1.000 / 0.998 inside (4 px, 0.01)")
    return res
