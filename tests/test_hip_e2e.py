"""-m gpu: whole hot path (model forward -> Detect decode -> NMS) on the HIP path vs the oracle and the committed
reference goldens (B=2 synthetic 640x640).  Tolerance from BASELINE.json north_star: 1e-3 on boxes/scores (f32 mode)."""

import contextlib

import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import tasks as ot
from ultralytics_pro_amd.utils import procedural as P

pytestmark = pytest.mark.gpu
TOL = 1e-3
# yolov8s: the reference's OWN f32 output moves by 2.2e-3 px between 8 and 1 CPU threads and sits 1.8e-3 .. 2.9e-3 px from
# its float64 run (tools/ref_noise_floor.py; box coordinates up to 640 px, DFL expectation x stride 32), so 1e-3 px is
# below the reference's reproducibility there; the gate for that config is the measured floor.  Scores stay at 1e-3.
BOX_TOL = {"yolov8s": 3e-3}


def _build(name, dtype, family=None):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m, family=family)
    m = m.to(DEV).eval()
    m.set_compute_dtype(dtype)
    return m


@pytest.mark.parametrize("name", ["yolov8n", "yolov8s", "yolov3-tiny", "yolov5-BoT3"])
def test_e2e_f32_matches_reference_golden(name, golden_dir):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    g = np.load(golden_dir / f"e2e_{name}.npz")
    m = _build(name, torch.float32)
    x = P.synthetic_images(2).to(DEV)
    with torch.no_grad():
        y = m(x)[0]
    torch.cuda.synchronize()
    yc = y.cpu()
    sel = g["anchor_sel"]
    d = np.abs(yc[:, :, sel].numpy() - g["y_sel"])
    print(f"{name} f32: max|box d|={d[:, :4].max():.3e} max|score d|={d[:, 4:].max():.3e}")
    box_tol = BOX_TOL.get(name, TOL)
    assert d[:, :4].max() <= box_tol and d[:, 4:].max() <= TOL
    out = non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    rows = torch.cat(out, 0).cpu().numpy()
    assert np.abs(rows[:, :4] - g["predict_rows"][:, :4]).max() <= box_tol
    assert np.abs(rows[:, 4] - g["predict_rows"][:, 4]).max() <= TOL
    assert np.array_equal(rows[:, 5], g["predict_rows"][:, 5])


def test_e2e_graph_replay_equals_eager():
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.float32)
    x = P.synthetic_images(2).to(DEV)
    with torch.no_grad():
        y_eager = m(x)[0].clone()
        run = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="e2e"))
        out, counts, keep = run()
        torch.cuda.synchronize()
        out1, c1 = out.clone(), counts.clone()
        out, counts, keep = run()
        torch.cuda.synchronize()
    assert torch.equal(out, out1) and torch.equal(counts, c1)
    ref = onms.non_max_suppression(y_eager.cpu(), 0.25, 0.7)
    assert counts.tolist() == [r.shape[0] for r in ref]
    for i, r in enumerate(ref):
        assert torch.equal(out[i, : r.shape[0]].cpu(), r)


# bf16 perf mode vs the REFERENCE goldens (not vs the HIP f32 path).  8 mantissa bits cannot hold 1e-3 px on 640-px boxes:
# the gate is detection-set agreement with the reference's f32 detections - one-to-one same-class matches at IoU >= 0.9 and
# at IoU >= 0.5, both ways - plus box / score deviation of the matched rows.  Bounds = measured on MI355X (round 2, the
# values are printed by the test and listed in DESIGN.md section 2) with headroom; the models carry RANDOM procedural weights
# whose scores crowd the 0.25 threshold (2 % of anchors above it by construction), so a 2e-3 score shift moves detections
# across the threshold and flips NMS decisions - at IoU >= 0.5 every config agrees to >= 95 %, at IoU >= 0.9 to 80-92 %.
# (The reference's own AMP check compares boxes with atol 0.5 on a trained model, utils/checks.py:780.)
BF16_BOUNDS = {  # name: (recall@.9, precision@.9, recall@.5, precision@.5, matched box p99 px, box max px, score p99)
    # yolov8n, three builds with the SAME rounding points (only f32 summation order differs: 0.02 % of model.2's outputs):
    # .906 .892 | .992 .977, then .945 .945 | .984 .984 (Detect branch tail), then .883 .890 | .961 .969 (fused model.2) -
    # 128 detections, so one threshold flip = 0.8 %; the bound sits below that spread
    "yolov8n": (0.82, 0.82, 0.93, 0.93, 0.7, 1.5, 0.005),      # matched box p99 0.33-0.38, max 0.52-0.56 px, score p99 .0019-.0025
    "yolov8s": (0.72, 0.76, 0.92, 0.95, 9.0, 15.0, 0.04),      # measured .800 .839 | .954 .984 | 6.97 9.83 0.0286
    # 51 reference rows: one flipped row is 2 %.  Rounds 2-3 measured .922 .940 | .980 .980 | 2.12 3.22 0.0098; round 4's 16 -> 32 layer on
    # the two-taps-per-k-step kernel (another f32 summation order, same rounding points: its unit tests are bit-exact / bf16-close)
    # .902 .939 | .941 .980 | 2.36 4.45 0.0100.  A report on the chaotic family; the gate is the smooth-family test below.
    "yolov3-tiny": (0.85, 0.85, 0.90, 0.93, 3.5, 6.0, 0.015),
    # rounds 2-3 measured .857 .866 | .979 .989 | 1.56 14.4 0.0243; round 4 with rows 0-1 on the fused stem kernel (99.958 % of the stem
    # output bit-identical to the rounding-point emulation, test_e2e_bf16_layer_by_layer...[yolov5-BoT3]) .725 .745 | .952 .978 | 1.68 2.69
    # 0.0307: on this chaotic family a different f32 summation order in the FIRST layer reshuffles which boxes agree to IoU 0.9 (the
    # matched-box median is 0.2 px on boxes a few pixels wide); the IoU 0.5 agreement and the deviations are what they were.
    "yolov5-BoT3": (0.65, 0.65, 0.93, 0.95, 2.5, 20.0, 0.04),
}


@pytest.mark.parametrize("name", list(BF16_BOUNDS))
def test_e2e_bf16_matches_reference_golden(name, golden_dir):
    """HIP bf16 pipeline (the mode the headline number is quoted in) vs tests/golden/e2e_<name>.npz = the imported
    reference's f32 CPU output on the same procedural weights and images."""
    from tests.hip_utils import DEV, detection_agreement, split_rows
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    g = np.load(golden_dir / f"e2e_{name}.npz")
    m = _build(name, torch.bfloat16)
    x = P.synthetic_images(2).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        y = m(x)[0]
    torch.cuda.synchronize()
    d = np.abs(y.cpu()[:, :, g["anchor_sel"]].numpy() - g["y_sel"])
    dbox, dsc = d[:, :4].ravel(), d[:, 4:].ravel()
    out = [o.cpu().numpy() for o in non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)]
    ref = split_rows(g["predict_rows"], g["predict_n"])
    a = detection_agreement(out, ref, 0.9)
    a5 = detection_agreement(out, ref, 0.5)
    print(f"{name} bf16 vs reference golden at IoU>=0.5: recall {a5['recall']:.3f} precision {a5['precision']:.3f}")
    print(f"{name} bf16 vs reference golden: head box |d| p50={np.median(dbox):.3f} p99={np.quantile(dbox, 0.99):.3f} "
          f"max={dbox.max():.3f} px, score max|d|={dsc.max():.4f}; detections {a['n_mine']} vs {a['n_ref']}: recall "
          f"{a['recall']:.3f} precision {a['precision']:.3f}, matched box p50={a['box_p50']:.3f} p99={a['box_p99']:.3f} "
          f"max={a['box_max']:.3f} px, score p99={a['score_p99']:.4f} max={a['score_max']:.4f}")
    # The IoU >= 0.9 agreement on this CHAOTIC family moves with every kernel that changes an f32 summation order (one flipped threshold
    # decision of ~50-130 rows is 1-2 %: see the table above), so its floor sits well under the measured spread - but it IS a floor:
    # a kernel that loses a tenth of the reference's boxes at IoU 0.9 fails here.  The tight gates are the per-seed test below, the
    # smooth-family tests and the rounding-point emulation, layer by layer.
    r9, p9, r5, p5, bp99, bmax, sp99 = BF16_BOUNDS[name]
    assert a5["recall"] >= r5 and a5["precision"] >= p5
    assert a["recall"] >= r9 and a["precision"] >= p9   # hard floor on the IoU >= 0.9 agreement (round 6: a gate again, see the table)
    assert a["box_p99"] <= bp99 and a["box_max"] <= bmax and a["score_p99"] <= sp99
    assert np.quantile(dbox, 0.99) <= max(bmax, 4.0) and dsc.max() <= 0.05


SEED_CONF = {"yolov8s": 0.01}  # default 0.1 (at the 0.25 of the other tests the random yolov8s head fires on one image seed in five)
SEED_BOUNDS = {  # name: (least IoU >= 0.5 recall / precision on ANY seed, least median over the seeds of the IoU >= 0.9 recall / precision)
    # measured on MI355X (round 5, six seeds x 6 images): yolov8n .968 | .906; yolov8s .903 | .862; yolov3-tiny .960 | .942; yolov5-BoT3 .925 | .854
    "yolov8n": (0.95, 0.87), "yolov8s": (0.87, 0.80), "yolov3-tiny": (0.93, 0.89), "yolov5-BoT3": (0.89, 0.80),
}


@pytest.mark.parametrize("name", list(SEED_BOUNDS))
def test_e2e_bf16_vs_f32_over_image_seeds(name):
    """The chaotic-family agreement as a DISTRIBUTION instead of one sample: six procedural image seeds, 6 images each, the bf16 mode
    against the f32 ORACLE on the same images (round 6; rounds 4-5 compared with the product's own f32 mode).  (The weight seed stays 0: the family's class bias was placed for that draw - with another weight seed the random head
    fires on 0 or on > 300 anchors per image, measured - so the samples vary the input.)  Gates: the IoU >= 0.5 agreement on EVERY seed
    and the median IoU >= 0.9 agreement over the seeds - bounds set from the measured spread (printed), not from one draw."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    mb = _build(name, torch.bfloat16)
    # round 6: the f32 side is the ORACLE itself (it was the product's own f32 mode, which the f32 tests pin to the oracle within 1e-3 -
    # transitively the same statement, but not an oracle test)
    o = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(o)
    o.fuse()
    conf = SEED_CONF.get(name, 0.1)
    rows = []
    for seed in range(6):
        x = P.synthetic_images(6, seed=seed)
        outs = {}
        with torch.no_grad():
            y = mb(x.to(DEV).to(torch.bfloat16).contiguous())[0]
            outs[torch.bfloat16] = [r.cpu().numpy() for r in non_max_suppression(y, conf_thres=conf, iou_thres=0.7, max_det=300)]
            outs[torch.float32] = [r.numpy() for r in onms.non_max_suppression(o(x)[0], conf, 0.7, max_det=300)]
        a9 = detection_agreement(outs[torch.bfloat16], outs[torch.float32], 0.9)
        a5 = detection_agreement(outs[torch.bfloat16], outs[torch.float32], 0.5)
        rows.append((a9["recall"], a9["precision"], a5["recall"], a5["precision"], a9["n_ref"]))
        print(f"  {name} seed {seed}: {a9['n_mine']} vs {a9['n_ref']} rows; IoU>=0.9 recall {a9['recall']:.3f} precision {a9['precision']:.3f}; "
              f"IoU>=0.5 {a5['recall']:.3f} / {a5['precision']:.3f}; matched box p99 {a9['box_p99']:.2f} px, score p99 {a9['score_p99']:.4f}")
    r = np.array([[v[0], v[1], v[2], v[3]] for v in rows if v[4] >= 20])  # (a seed whose images fire on < 20 anchors says nothing)
    assert len(r) >= 4, "too few image seeds with detections"
    print(f"{name} over {len(r)} seeds: IoU>=0.9 median {np.median(r[:, 0]):.3f} / {np.median(r[:, 1]):.3f} (min {r[:, 0].min():.3f} / {r[:, 1].min():.3f}); "
          f"IoU>=0.5 min {r[:, 2].min():.3f} / {r[:, 3].min():.3f}")
    lo5, med9 = SEED_BOUNDS[name]
    assert min(r[:, 2].min(), r[:, 3].min()) >= lo5
    assert min(np.median(r[:, 0]), np.median(r[:, 1])) >= med9


# ---- the "smooth" weight family: bf16 pinned at the reference's own AMP tolerance -----------------------------------------
# utils/procedural.py SMOOTH_RECIPE: bf16-exact conv weights (what a half-precision checkpoint holds), BatchNorm scale exactly
# 1, small boxes that do not compete in NMS.  On it the bf16 pipeline must reproduce the f32 REFERENCE detections
# (tests/golden/e2e_<cfg>_smooth.npz, generated from the imported reference): every matched box within 0.5 px - the reference's
# AMP self-check, utils/checks.py:780 `torch.allclose(a.boxes.data, b.boxes.data, atol=0.5)` - and the detection sets equal
# except for rows whose score lies within +-SMOOTH_BAND of conf_thres: a detection's presence is a step function of its score,
# so a row the reference scores 0.2508 can legitimately come out at 0.2493 (bf16 activations move scores by up to 3e-3).
# Bounds: (raw recall, raw precision) at IoU >= 0.9.  Measured on MI355X (round 3): yolov8s, yolov3-tiny, yolov5-BoT3 1.000 / 1.000
# (171, 42, 166 rows; matched boxes <= 0.02 px, scores <= 0.0015); yolov8n 0.985 / 0.939 (214 vs 204 rows) with EVERY mismatch a
# threshold-band row (outside the band 1.0 / 1.0, matched boxes <= 0.025 px): the narrow network's class map is nearly flat in
# space at any conv gain below the onset of chaos (6.5 -> 6.7: score noise x 10, tools/experiments/smooth_scan.py), so 40 % of
# its 204 reference rows score within +-0.005 of 0.25 - its raw bound is what that crowding allows, the band-excluded and the
# 0.5 px gates below are the same for all four.
SMOOTH_BAND = 0.005
SMOOTH_BOUNDS = {"yolov8n": (0.93, 0.90), "yolov8s": (0.97, 0.97), "yolov3-tiny": (0.97, 0.97), "yolov5-BoT3": (0.97, 0.97)}


def _dispatch(which):
    """The option sets a model runs under: "session" = the test session's (tests/conftest.py: every eligible shape on the pipelined /
    pair kernels), "serial" = the library defaults (`upa_opts` all zero: what `model(x)` and `bench.py --serial` run), "throughput" =
    the library defaults + what `engine/pipeline.py: PipelinedRunner.throughput_opts` adds for copies in flight - the dispatch every
    quoted images/s figure runs under."""
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    if which == "session":
        return contextlib.nullcontext()
    if which == "serial":
        return R.use_opts(L.Opts())
    assert which == "throughput"
    return R.use_opts(L.Opts(c2f=4, conv_ws3=1, c2f_stream_rows=-1, detect_stream=2, conv_big=2))


@pytest.mark.parametrize("dispatch", ["session", "serial", "throughput"])
@pytest.mark.parametrize("name", list(SMOOTH_BOUNDS))
def test_e2e_bf16_smooth_family_matches_reference_golden(name, dispatch, golden_dir):
    """bf16 on the smooth family against the REFERENCE's f32 detections, under each of the three dispatches (round-5 review: only
    yolov8n was compared under the throughput dispatch; the kernels a config runs on in the bench must be the kernels that are pinned)."""
    from tests.hip_utils import DEV, detection_agreement, split_rows
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    g = np.load(golden_dir / f"e2e_{name}_smooth.npz")
    m = _build(name, torch.bfloat16, family="smooth:" + name)
    x = P.synthetic_images(2).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad(), _dispatch(dispatch):
        y = m(x)[0]
    torch.cuda.synchronize()
    d = np.abs(y.cpu()[:, :, g["anchor_sel"]].numpy() - g["y_sel"])
    out = [o.cpu().numpy() for o in non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)]
    ref = split_rows(g["predict_rows"], g["predict_n"])
    a = detection_agreement(out, ref, 0.9)
    ref_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in ref]
    out_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in out]
    rec_x = detection_agreement(out, ref_x, 0.9)["recall"]       # every reference row outside the band is found ...
    prec_x = detection_agreement(out_x, ref, 0.9)["precision"]   # ... and every row of mine outside it exists in the reference
    print(f"{name} smooth bf16 [{dispatch}] vs reference golden: head box max|d| {d[:, :4].max():.3f} px score max|d| {d[:, 4:].max():.4f}; detections "
          f"{a['n_mine']} vs {a['n_ref']}: recall {a['recall']:.3f} precision {a['precision']:.3f} (outside the +-{SMOOTH_BAND} band: "
          f"{rec_x:.4f} / {prec_x:.4f}; band rows ref {sum(map(len, ref)) - sum(map(len, ref_x))} mine {sum(map(len, out)) - sum(map(len, out_x))}), "
          f"matched box p99 {a['box_p99']:.3f} max {a['box_max']:.3f} px, score max {a['score_max']:.4f}")
    r9, p9 = SMOOTH_BOUNDS[name]
    assert a["recall"] >= r9 and a["precision"] >= p9
    assert rec_x >= 0.995 and prec_x >= 0.995
    assert a["box_max"] <= 0.5 and a["score_max"] <= SMOOTH_BAND       # the reference's AMP tolerance on every matched row
    assert d[:, :4].max() <= 0.5 and d[:, 4:].max() <= SMOOTH_BAND     # ... and on the sampled head outputs


@pytest.mark.parametrize("name", ["yolov8n", "yolov5-BoT3"])
def test_e2e_f32_smooth_family_matches_reference_golden(name, golden_dir):
    """f32 parity mode on the smooth family: the 1e-3 north_star tolerance, rows identical."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    g = np.load(golden_dir / f"e2e_{name}_smooth.npz")
    m = _build(name, torch.float32, family="smooth:" + name)
    with torch.no_grad():
        y = m(P.synthetic_images(2).to(DEV))[0]
    torch.cuda.synchronize()
    d = np.abs(y.cpu()[:, :, g["anchor_sel"]].numpy() - g["y_sel"])
    assert d[:, :4].max() <= TOL and d[:, 4:].max() <= TOL
    out = non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    rows = torch.cat(out, 0).cpu().numpy()
    assert np.abs(rows[:, :5] - g["predict_rows"][:, :5]).max() <= TOL and np.array_equal(rows[:, 5], g["predict_rows"][:, 5])


@pytest.mark.parametrize("shape", [(1, 352, 416), (3, 320, 320), (2, 224, 640)], ids=["1x352x416", "3x320x320", "2x224x640"])
def test_e2e_bf16_fused_paths_match_unfused_at_other_sizes(shape):
    """yolov8n bf16 at input sizes other than 640 x 640 (ragged tiles in every fused kernel, rectangular maps, batch 1 / 3):
    the default path - fused stem, whole-block C2f kernels, Bottleneck + cv2, virtual Upsample + Concat, Detect branch tails,
    NMS keys - against the same model with every one of those switches off (one launch per conv, materialised upsample,
    separate decode, full-scan NMS).  Same bf16 rounding points, so the decoded outputs agree to bf16 resolution and the
    detection sets almost entirely."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.nn.modules import block as pblock
    from ultralytics_pro_amd.nn.modules import head as phead
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    n, h, w = shape
    m = _build("yolov8n", torch.bfloat16)
    det = m.model[-1]
    x = P.synthetic_images(n, h=h, w=w).to(DEV).to(torch.bfloat16).contiguous()
    saved = (pblock.C2f.fuse_block, pblock.C2f.fuse_pair_cv2, pblock.Bottleneck.fuse_pair, phead.Detect.fuse_branch,
             phead.Detect.fuse_decode, type(m).virtual_upsample)
    try:
        with torch.no_grad():
            det.keep_raw, det.nms_keys = False, True
            y1 = m(x)[0]
            d1 = [o.cpu().numpy() for o in non_max_suppression(y1, conf_thres=0.25, iou_thres=0.7, max_det=300)]
            y1 = y1.float().cpu()
            pblock.C2f.fuse_block = pblock.C2f.fuse_pair_cv2 = pblock.Bottleneck.fuse_pair = False
            phead.Detect.fuse_branch = phead.Detect.fuse_decode = False
            type(m).virtual_upsample = False
            det.keep_raw, det.nms_keys = True, False
            y0 = m(x)[0]
            d0 = [o.cpu().numpy() for o in non_max_suppression(y0, conf_thres=0.25, iou_thres=0.7, max_det=300)]
            y0 = y0.float().cpu()
    finally:
        (pblock.C2f.fuse_block, pblock.C2f.fuse_pair_cv2, pblock.Bottleneck.fuse_pair, phead.Detect.fuse_branch,
         phead.Detect.fuse_decode, type(m).virtual_upsample) = saved
    assert y1.shape == y0.shape == (n, 84, (h // 8) * (w // 8) + (h // 16) * (w // 16) + (h // 32) * (w // 32))
    d = (y1 - y0).abs()
    # the separate decode reads bf16-rounded logits, the fused one f32 accumulators: boxes differ by bf16 resolution of a DFL
    # expectation x stride (<= a few px at stride 32), scores by one bf16 ulp of the logit
    print(f"{shape}: box |d| p99={d[:, :4].flatten().quantile(0.99).item():.3f} max={d[:, :4].max().item():.3f}, "
          f"score max={d[:, 4:].max().item():.4f}")
    assert d[:, :4].flatten().quantile(0.99).item() <= 2.0 and d[:, :4].max().item() <= 8.0 and d[:, 4:].max().item() <= 0.03
    a = detection_agreement(d1, d0, 0.5)
    assert a["recall"] >= 0.9 and a["precision"] >= 0.9, a


@pytest.mark.parametrize("name,conf,multi_label", [("yolov8n", 0.25, False), ("yolov8n", 0.4, False), ("yolov8n", 0.4, True),
                                                   ("yolov8s", 0.25, False)])
def test_e2e_bf16_nms_prefilter_is_exact(name, conf, multi_label):
    """NMS prefilter (`Detect.nms_keys`, `upa_nms_batched_hot`): the fused class tails write the NMS key of every anchor's best
    class next to the scores, and single-label non_max_suppression (nms.py:13-166) compacts those keys instead of re-reading
    the scores - the detections must be BIT-identical to the full scan of the same output, eagerly and in a replayed
    hipGraph, at any conf_thres; multi_label and outputs without keys (a clone) take the full scan."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build(name, torch.bfloat16)
    det = m.model[-1]
    det.keep_raw = False
    x = P.synthetic_images(3).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        det.nms_keys = False
        y0 = m(x)[0]
        assert not hasattr(y0, "_upa_hot")
        full = [t.clone() for t in nms_raw(y0, conf, 0.7, multi_label=multi_label, key="full")]
        det.nms_keys = True
        y1 = m(x)[0]
        assert torch.equal(y0, y1)
        assert getattr(y1, "_upa_hot", None) is not None, "every class launch of the fused decode writes the keys"
        hot = [t.clone() for t in nms_raw(y1, conf, 0.7, multi_label=multi_label, key="hot")]
        for a, b in zip(full, hot):
            assert torch.equal(a, b)
        if not multi_label:  # class filter (nms.py:100-101) on the key path: the mask is applied to the key's class
            fc = [t.clone() for t in nms_raw(y1.clone(), conf, 0.7, classes=[0, 3, 17, 42, 79], key="fc")]
            hc = [t.clone() for t in nms_raw(y1, conf, 0.7, classes=[0, 3, 17, 42, 79], key="hc")]
            for a, b in zip(fc, hc):
                assert torch.equal(a, b)
            fa = [t.clone() for t in nms_raw(y1.clone(), conf, 0.7, agnostic=True, max_det=50, key="fa")]
            ha = [t.clone() for t in nms_raw(y1, conf, 0.7, agnostic=True, max_det=50, key="ha")]
            for a, b in zip(fa, ha):
                assert torch.equal(a, b)
        low = [t.clone() for t in nms_raw(y1, 0.05, 0.7, key="low")]
        ref_low = [t.clone() for t in nms_raw(y1.clone(), 0.05, 0.7, key="low2")]  # a clone carries no keys: full scan
        for a, b in zip(low, ref_low):
            assert torch.equal(a, b)
        run = m.compile(x, post=lambda o: nms_raw(o[0], conf, 0.7, multi_label=multi_label, key="graph"))
        for _ in range(3):
            out = run()
            torch.cuda.synchronize()
            for a, b in zip(full, out):
                assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["yolov8n", "yolov8s", "yolov3-tiny", "yolov5-BoT3"])
def test_e2e_f32_val_mode_matches_reference_golden(name, golden_dir):
    """The validate path of the same pipeline (conf 0.001, multi_label=True, max_det 300: validator defaults,
    engine/validator.py + utils/nms.py:115-116): f32 HIP vs the reference rows stored in the e2e fixtures.  With 300 rows
    per image cut by max_det, rows whose scores differ by less than the 1e-3 tolerance may swap places or fall on the
    other side of the cut, so rows are matched one-to-one (same class, IoU >= 0.99) instead of position by position."""
    from tests.hip_utils import DEV, detection_agreement, split_rows
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    g = np.load(golden_dir / f"e2e_{name}.npz")
    m = _build(name, torch.float32)
    x = P.synthetic_images(2).to(DEV)
    with torch.no_grad():
        y = m(x)[0]
    out = [o.cpu().numpy() for o in non_max_suppression(y, conf_thres=0.001, iou_thres=0.7, max_det=300, multi_label=True)]
    ref = split_rows(g["val_rows"], g["val_n"])
    assert [o.shape[0] for o in out] == [r.shape[0] for r in ref]
    exact = sum(int(np.array_equal(o[:, 5], r[:, 5]) and np.abs(o[:, :5] - r[:, :5]).max() <= TOL) for o, r in zip(out, ref))
    a = detection_agreement(out, ref, 0.99)
    print(f"{name} val mode f32: {exact}/{len(ref)} images identical row by row; recall {a['recall']:.4f} precision "
          f"{a['precision']:.4f} box max {a['box_max']:.2e} score max {a['score_max']:.2e}")
    assert a["recall"] >= 0.99 and a["precision"] >= 0.99
    assert a["box_max"] <= 2 * BOX_TOL.get(name, TOL) and a["score_max"] <= TOL


def test_e2e_rtdetr_bf16_backbone_matches_reference_golden(golden_dir):
    """Config 5 in its perf mode (bf16 darknet53 backbone, f32 RTDETRDecoder) vs the reference golden: the top-300 query
    selection (head.py:2175) is order-sensitive, so queries are compared as sets - one-to-one by box IoU >= 0.9 and equal
    arg-max class - and the post-processed detections (conf 0.25) by the same agreement measure as the Detect configs."""
    from tests.hip_utils import DEV, detection_agreement, split_rows
    from ultralytics_pro_amd.utils.nms import rtdetr_postprocess
    g = np.load(golden_dir / "e2e_yolov3-rtdetr.npz")
    m = _build("yolov3-rtdetr", torch.bfloat16)
    x = P.synthetic_images(2).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        y = m(x)[0]
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and tuple(y.shape) == tuple(g["y"].shape)

    def rows(t):  # (300, 84) cxcywh-normalised + scores -> xyxy pixels, max score, class
        t = np.asarray(t, dtype=np.float64)
        cx, cy, w, h = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        return np.stack([(cx - w / 2) * 640, (cy - h / 2) * 640, (cx + w / 2) * 640, (cy + h / 2) * 640,
                         t[:, 4:].max(1), t[:, 4:].argmax(1)], 1)

    yc = y.cpu().numpy()
    q = detection_agreement([rows(yc[b]) for b in range(2)], [rows(g["y"][b]) for b in range(2)], 0.9)
    post = [o.cpu().numpy() for o in rtdetr_postprocess(y, 0.25)]
    a = detection_agreement(post, split_rows(g["post_rows"], g["post_n"]), 0.9)
    print(f"rtdetr bf16 backbone vs reference golden: query-set overlap {q['recall']:.3f} (box p99 {q['box_p99']:.3f} px, score "
          f"p99 {q['score_p99']:.4f}); detections {a['n_mine']} vs {a['n_ref']}: recall {a['recall']:.3f} precision "
          f"{a['precision']:.3f} box p99 {a['box_p99']:.3f} max {a['box_max']:.3f} px score p99 {a['score_p99']:.4f}")
    # Measured on MI355X (round 2): query-set overlap 0.30, detections 103 vs 102 with recall 0.245 / precision 0.243 at
    # IoU >= 0.9, matched rows within 2.1 px (p99) / 0.02 in score.  The random-weight encoder scores of the 8400 tokens are
    # nearly flat, so WHICH 300 tokens are selected (head.py:2175) is decided by differences far below bf16 resolution: the
    # bf16 backbone picks a mostly different - equally valid - query set; the rows it shares with the reference agree.  The
    # order-exact comparison lives in the f32 test above; this one pins the bf16 behaviour as measured.
    # Round 6: the query-set overlap is a REPORT here (it was gated at 0.2, a bound 80 % wrong rows pass).  What pins this mode against
    # the oracle are `test_e2e_bf16_rtdetr_encoder_head_every_token_vs_oracle` (everything in front of the selection, token by token)
    # and `test_e2e_bf16_rtdetr_decoder_with_oracle_queries_vs_oracle` (everything behind it, row by row) - both at >= 0.99.
    assert a["n_mine"] >= 0.8 * a["n_ref"] and a["n_mine"] <= 1.25 * a["n_ref"]
    assert a["box_p99"] <= 4.0 and a["score_p99"] <= 0.04


def test_e2e_rtdetr_f32_matches_reference_golden(golden_dir):
    """Config 5: yolov3-rtdetr (darknet53 backbone + RTDETRDecoder), B=2: queries are compared row by row - the top-300
    selection order must reproduce the reference's (head.py:2175)."""
    from tests.hip_utils import DEV
    g = np.load(golden_dir / "e2e_yolov3-rtdetr.npz")
    m = _build("yolov3-rtdetr", torch.float32)
    x = P.synthetic_images(2).to(DEV)
    with torch.no_grad():
        y = m(x)[0]
    torch.cuda.synchronize()
    d = np.abs(y.cpu().numpy() - g["y"])
    print(f"rtdetr f32: max|box d|={d[..., :4].max():.3e} (normalised) max|score d|={d[..., 4:].max():.3e}")
    assert d.max() <= TOL
    outs = onms.rtdetr_postprocess(y.cpu(), 0.25)
    assert [o.shape[0] for o in outs] == list(g["post_n"])
    # the product's RTDETRPredictor.postprocess (upa_rtdetr_postprocess) on the same decoder output: bit-exact
    from ultralytics_pro_amd.utils.nms import rtdetr_postprocess
    mine = rtdetr_postprocess(y, 0.25)
    for a, b in zip(mine, outs):
        assert torch.equal(a.cpu(), b)


def test_rtdetr_postprocess_vs_oracle_random():
    """upa_rtdetr_postprocess vs the oracle's restatement of models/rtdetr/predict.py:35-74 on random decoder outputs:
    confidence filter, class filter, ties in the score (stable order), max_det truncation, per-image original shapes,
    an image with no detection; boxes / scores / classes bit-exact."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import rtdetr_postprocess
    B, Q, nc = 5, 300, 80
    box = P.uniform("rtp_box", (B, Q, 4), 0.05, 0.9)
    sc = P.uniform("rtp_sc", (B, Q, nc), 0.0, 0.6)
    sc[1, 10] = sc[1, 3]             # a tie in every class of two queries
    sc[2] *= 0.3                     # nothing above conf in image 2
    sc[3, :, 7] = 0.9 + 0.0001 * torch.arange(Q)  # every query valid: max_det truncation
    preds = torch.cat([box, sc], -1).contiguous()
    for conf, max_det, classes, shapes in ((0.25, 300, None, None), (0.4, 100, None, None), (0.25, 300, [7, 11, 42], None),
                                           (0.3, 50, None, [(480, 640), (640, 640), (100, 200), (720, 1280), (333, 500)])):
        ref = []
        for i in range(B):
            hw = (640, 640) if shapes is None else shapes[i]
            ref += onms.rtdetr_postprocess(preds[i:i + 1], conf, max_det, imgsz=hw, classes=classes)
        mine = rtdetr_postprocess(preds.to(DEV), conf, max_det, classes=classes, orig_shapes=shapes)
        assert [m.shape[0] for m in mine] == [r.shape[0] for r in ref]
        for a, b in zip(mine, ref):
            assert torch.equal(a.cpu(), b)
    assert ref[2].shape[0] == 0


@pytest.mark.parametrize("shape", [(1, 384, 640), (3, 320, 256), (2, 352, 608)], ids=["b1_384x640", "b3_320x256", "b2_352x608"])
def test_e2e_rect_and_odd_batches_vs_oracle(shape):
    """Non-square inputs (the reference predicts with rect=True, engine/model.py:527), batch 1 and odd batches, feature
    maps that are not multiples of the conv tiles (44x76, 40x32, ...): HIP f32 vs the oracle, boxes/scores/classes."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    b, h, w = shape
    x = P.synthetic_images(b, h=h, w=w)
    o = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(o)
    o.fuse()
    with torch.no_grad():
        y_ref = o(x)[0]
    m = _build("yolov8n", torch.float32)
    with torch.no_grad():
        y = m(x.to(DEV))[0]
    torch.cuda.synchronize()
    assert y.shape == y_ref.shape
    d = (y.cpu() - y_ref).abs()
    assert d[:, :4].max().item() <= TOL and d[:, 4:].max().item() <= TOL
    out = non_max_suppression(y, 0.25, 0.7)
    ref = onms.non_max_suppression(y_ref, 0.25, 0.7)
    assert [a.shape[0] for a in out] == [r.shape[0] for r in ref]
    for a, r in zip(out, ref):
        if r.shape[0]:
            assert (a.cpu()[:, :5] - r[:, :5]).abs().max().item() <= TOL
            assert torch.equal(a.cpu()[:, 5], r[:, 5])


@pytest.mark.parametrize("nc", [1, 3, 20])
def test_e2e_arbitrary_class_count_vs_oracle(nc):
    """the reference accepts any nc; the class branch of Detect is padded to the 16-byte store width inside the
    product (zero filters, never read back) - nc = 1, 3, 20 in f32 vs the oracle (1e-3) and in bf16 (runs, same shapes,
    detection-set agreement with the f32 oracle)."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x = P.synthetic_images(2, h=320, w=320)
    o = ot.DetectionModel("yolov8n.yaml", nc=nc)
    P.apply_procedural_weights(o)
    o.fuse()
    with torch.no_grad():
        y_ref = o(x)[0]
    assert y_ref.shape[1] == 4 + nc
    ref = onms.non_max_suppression(y_ref, 0.25, 0.7)
    for dt in (torch.float32, torch.bfloat16):
        m = DetectionModel("yolov8n.yaml", nc=nc)
        P.apply_procedural_weights(m)
        m = m.to(DEV).eval()
        m.set_compute_dtype(dt)
        xd = x.to(DEV) if dt == torch.float32 else x.to(DEV).to(torch.bfloat16).contiguous()
        with torch.no_grad():
            y, raw = m(xd)
        torch.cuda.synchronize()
        assert y.shape == y_ref.shape and all(r.shape[1] == 64 + nc for r in raw)
        out = non_max_suppression(y, 0.25, 0.7)
        if dt == torch.float32:
            d = (y.cpu() - y_ref).abs()
            assert d[:, :4].max().item() <= TOL and d[:, 4:].max().item() <= TOL
            assert [a.shape[0] for a in out] == [r.shape[0] for r in ref]
        else:
            a = detection_agreement([t.cpu().numpy() for t in out], [r.numpy() for r in ref], 0.9)
            print(f"nc={nc} bf16: recall {a['recall']:.3f} precision {a['precision']:.3f} ({a['n_mine']} vs {a['n_ref']} rows)")
            assert a["n_ref"] < 10 or (a["recall"] >= 0.8 and a["precision"] >= 0.8)  # a handful of rows at 320 px


def test_e2e_micro_batched_graph_equals_single_graph():
    """compile(micro_batches=2): two sub-batches on parallel hipGraph branches give bit-identical detections."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.float32)
    x = P.synthetic_images(4).to(DEV)
    with torch.no_grad():
        run1 = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="one"))
        out1, cnt1, _ = run1()
        torch.cuda.synchronize()
        out1, cnt1 = out1.clone(), cnt1.clone()
        run2 = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="two"), micro_batches=2)
        parts = run2()
        parts = run2()
        torch.cuda.synchronize()
    out2 = torch.cat([p_[0] for p_ in parts], 0)
    cnt2 = torch.cat([p_[1] for p_ in parts], 0)
    assert torch.equal(cnt1, cnt2) and torch.equal(out1, out2)


def test_e2e_pipelined_runner_copies_equal_single_graph():
    """engine.pipeline.PipelinedRunner: three compiled copies of the step in flight on separate streams (each split into
    two concurrent sub-batches) - every copy reproduces the single-graph detections bit for bit, repeatedly.  The runner picks its
    kernels for throughput (`PipelinedRunner.throughput_opts`: the 40 x 40 C2f blocks as separate launches, conv_big instead of the
    persistent 3x3, whole-height strips in the line-buffer C2f kernel, the line-buffer form of the 80 x 80 Detect level), so the single graph it is compared with is compiled under the same
    options."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.bfloat16)
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        with R.use_opts(c2f=4, conv_ws3=1, c2f_stream_rows=-1, detect_stream=2, conv_big=2):
            run1 = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="ref"))
        out1, cnt1, _ = run1()
        torch.cuda.synchronize()
        out1, cnt1 = out1.clone(), cnt1.clone()
        runner = PipelinedRunner(m, x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="pipe"), micro_batches=2, in_flight=3)
        lin = PipelinedRunner(m, x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="lin"), micro_batches=1, in_flight=4, linear=True)
        for _ in range(7):  # not a multiple of in_flight: the copies end at different points of the rotation
            runner.step()
            lin.step()
        torch.cuda.synchronize()
    assert runner.i == 7 and len(runner.results()) == 3 and len(lin.results()) == 4
    assert runner.throughput_opts == lin.throughput_opts == {"c2f": 4, "conv_ws3": 1, "c2f_stream_rows": -1, "detect_stream": 2, "conv_big": 2}
    assert m.model[-1].concurrent  # the linear runner restored the head's concurrency flag
    for parts in runner.results():
        out = torch.cat([p_[0] for p_ in parts], 0)
        cnt = torch.cat([p_[1] for p_ in parts], 0)
        assert torch.equal(cnt, cnt1) and torch.equal(out, out1)
    for (out, cnt, _) in lin.results():  # linear copies: one graph, whole batch
        assert torch.equal(cnt, cnt1) and torch.equal(out, out1)


def _iou_matrix(b):
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(2)
    return inter / (area[:, None] + area[None, :] - inter)


def test_e2e_full_size_properties():
    """BASELINE.json's headline configuration itself (yolov8n, 32 x 3 x 640 x 640, bf16, conf 0.25, iou 0.7) through
    size-independent properties: every copy of the pipelined runner (the bench default: 4 linear graphs in flight)
    reproduces the single-graph detections bit for bit; per image the detections are sorted by score (nms.py:137-141),
    above the confidence threshold, at most max_det, finite, inside the letterboxed image up to the box size; NMS is
    idempotent - no two kept boxes of one class overlap by more than iou_thres (nms.py:143-156), so running the
    reference's NMS on its own output would keep all of them; and an image's detections do not depend on the batch it
    is in (f32 parity mode: the same four images as a batch of 4, boxes / scores to 1e-3, equal classes)."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.bfloat16)
    x32 = P.synthetic_images(32).to(DEV)
    x = x32.to(torch.bfloat16).contiguous()
    with torch.no_grad():
        with R.use_opts(c2f=4, conv_ws3=1, c2f_stream_rows=-1, detect_stream=2, conv_big=2):  # the runner's throughput dispatch (PipelinedRunner.throughput_opts)
            run1 = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="ref"))
        out1, cnt1, _ = run1()
        torch.cuda.synchronize()
        out1, cnt1 = out1.clone(), cnt1.clone()
        lin = PipelinedRunner(m, x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="lin"), micro_batches=1, in_flight=4, linear=True)
        for _ in range(9):
            lin.step()
        torch.cuda.synchronize()
    for (out, cnt, _) in lin.results():
        assert torch.equal(cnt, cnt1) and torch.equal(out, out1)
    out, cnt = out1.cpu(), cnt1.cpu().tolist()
    assert sum(cnt) > 32, "the synthetic batch is expected to produce detections"
    for i, n in enumerate(cnt):
        assert 0 <= n <= 300
        d = out[i, :n]
        if n == 0:
            continue
        assert torch.isfinite(d).all()
        assert (d[:, 4] > 0.25).all() and (d[:, 4] <= 1.0).all()
        assert (d[1:, 4] <= d[:-1, 4]).all(), "scores must be descending"
        assert ((d[:, 5] >= 0) & (d[:, 5] < 80) & (d[:, 5] == d[:, 5].round())).all()
        assert (d[:, 2] >= d[:, 0]).all() and (d[:, 3] >= d[:, 1]).all()
        # idempotence: the reference's greedy NMS (on class-offset boxes, offsets of cls * 7680 as nms.py:143-156 - at
        # those magnitudes f32 coordinates carry 1/16 px, so the IoU the NMS sees differs from the plain-box IoU by up to
        # ~1e-3) keeps every one of the kept boxes
        off = d[:, 5:6] * 7680.0
        kept = onms.greedy_nms(d[:, :4] + off, d[:, 4], 0.7)
        assert kept.numel() == n, "the reference NMS applied to the kept detections drops some of them"
        iou = _iou_matrix(d[:, :4])
        same = d[:, 5][:, None] == d[:, 5][None, :]
        iou = torch.where(same, iou, torch.zeros_like(iou)).triu(1)
        assert iou.max().item() <= 0.7 + 5e-3, "two kept boxes of one class overlap by more than iou_thres"
    # batch-composition independence, f32 parity mode
    mf = _build("yolov8n", torch.float32)
    with torch.no_grad():
        o32, c32, _ = nms_raw(mf(x32)[0], 0.25, 0.7, key="f32_32")
        o4, c4, _ = nms_raw(mf(x32[8:12].contiguous())[0], 0.25, 0.7, key="f32_4")
        torch.cuda.synchronize()
    assert torch.equal(c32[8:12].cpu(), c4.cpu())
    for j in range(4):
        n = int(c4[j])
        a, b = o32[8 + j, :n].cpu(), o4[j, :n].cpu()
        assert (a[:, :5] - b[:, :5]).abs().max().item() <= TOL if n else True
        assert torch.equal(a[:, 5], b[:, 5])


@pytest.mark.parametrize("name", ["yolov8n", "yolov8s", "yolov3-tiny", "yolov5-BoT3"])
def test_e2e_grouped_detect_levels_equal_level_by_level(name):
    """`Detect.group_levels` (the head's levels through `upa_conv2d_bias_act_group` / `upa_detect_branch_tail_group` when its branches
    run on one stream, as in the linear graphs of the throughput runner): the decoded output is bit-identical to the level-by-level
    walk, for every Detect config - including heads whose class branch is outside the branch-tail form (yolov8s: c3 = 128 goes
    level by level inside the grouped walk) and two-level heads (yolov3-tiny).  Round 5: the grouped walk also STACKS the two first convs
    of the first level into one 64 + 80 = 144-channel convolution (`Detect.stack_first`) - still bit-identical per output channel."""
    from tests.hip_utils import DEV
    m = _build(name, torch.bfloat16)
    det = m.model[-1]
    det.keep_raw = False
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    saved = det.concurrent
    res = {}
    try:
        with torch.no_grad():
            # (a) library defaults, the first level's first convs as two problems: grouping alone must not change a bit
            # (b) the persistent 3x3 kernel off (as in the throughput runner; its accumulation order differs from conv_big's in the last
            #     bit, and the stacked 144-channel first conv of the 80 x 80 level runs on conv_big): grouping + stacking vs level by level
            for tag, kw in (("a", {"no_stack_first": 1}), ("b", {"conv_ws3": 1})):
                with R.use_opts(L.Opts(**kw)):
                    det.concurrent = False  # one stream: the grouped walk
                    det.group_levels = True
                    y_grouped = m(x)[0].clone()
                    det.group_levels = False
                    y_levels = m(x)[0].clone()
                    det.concurrent = True   # forked branches (the default eager walk)
                    y_forked = m(x)[0].clone()
                res[tag] = (y_grouped, y_levels, y_forked)
    finally:
        det.concurrent = saved
        det.group_levels = True
    torch.cuda.synchronize()
    for tag, (y_grouped, y_levels, y_forked) in res.items():
        assert torch.equal(y_grouped, y_levels) and torch.equal(y_grouped, y_forked), tag



def test_e2e_detect_walks_under_default_options_agree_to_the_last_bits():
    """Under LIBRARY-DEFAULT options the grouped walk stacks the 80 x 80 level's first convs on conv_big while the level-by-level and forked
    walks run the 64 -> 64 box conv on conv_ws3: two kernels with the same rounding points and another f32 summation order.  So with the defaults
    the Detect output depends on the walk in the last bits only (round-5 advisor finding: state it and test it): boxes within 0.05 px, scores
    within 2e-3 (a flipped bf16 tie of an intermediate), at most 1 % of the level's values touched at all; the other levels bit-identical."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    m = _build("yolov8n", torch.bfloat16, family="smooth:yolov8n")
    det = m.model[-1]
    det.keep_raw = False
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    saved = det.concurrent
    try:
        with torch.no_grad(), R.use_opts(L.Opts()):
            det.concurrent, det.group_levels = False, True
            y_grouped = m(x)[0].float().cpu().clone()
            det.group_levels = False
            y_levels = m(x)[0].float().cpu().clone()
            det.concurrent = True
            y_forked = m(x)[0].float().cpu().clone()
    finally:
        det.concurrent, det.group_levels = saved, True
    assert torch.equal(y_levels, y_forked)
    d = (y_grouped - y_levels).abs()
    a0 = 80 * 80
    assert d[:, :, a0:].max().item() == 0.0
    touched = (d[:, :, :a0] > 0).float().mean().item()
    print(f"default options, grouped vs level-by-level walk: box |d| max {d[:, :4].max():.4f} px, score |d| max {d[:, 4:].max():.5f}, values touched {touched:.4f}")
    assert d[:, :4].max().item() <= 0.05 and d[:, 4:].max().item() <= 2e-3 and touched <= 0.01


# ---- full-size comparisons with the oracle (round-3 review item 2): BASELINE.json's configurations at THEIR batch sizes ----------
def _oracle_full(name, batch, family=None):
    o = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(o, family=family)
    o.fuse()
    x = P.synthetic_images(batch)
    with torch.no_grad():
        y = o(x)[0]
    return x, y


def test_e2e_f32_headline_batch_vs_oracle():
    """The headline configuration itself - yolov8n, 32 x 3 x 640 x 640 - in f32 parity mode against the oracle on all 32 images:
    every anchor of the head output within 1e-3 (boxes in px, scores), post-NMS rows identical row by row."""
    from tests.hip_utils import DEV, rows_equivalent, rows_identical
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x, y_ref = _oracle_full("yolov8n", 32)
    ref = [r.numpy() for r in onms.non_max_suppression(y_ref, 0.25, 0.7, max_det=300)]
    m = _build("yolov8n", torch.float32)
    with torch.no_grad():
        y = m(x.to(DEV))[0]
        out = [o.cpu().numpy() for o in non_max_suppression(y, 0.25, 0.7, max_det=300)]
    d = (y.cpu() - y_ref).abs()
    eq, rb, rs = rows_identical(out, ref, TOL)
    print(f"yolov8n f32 bs 32 vs oracle: head max|box d| {d[:, :4].max():.3e} px max|score d| {d[:, 4:].max():.3e}; "
          f"{sum(map(len, out))} vs {sum(map(len, ref))} rows, row by row: equal={eq} box {rb:.3e} score {rs:.3e}")
    assert d[:, :4].max().item() <= TOL and d[:, 4:].max().item() <= TOL
    # 1872 rows over 32 images: a row whose score or whose IoU with a kept row sits within the tolerance of its threshold may be
    # present on one side only (measured on MI355X: one such row); every other row must have its partner within 1e-3
    rq = rows_equivalent(out, ref, TOL, 0.25, 0.7)
    print(f"  {rq}")
    assert rq["equivalent"] and rq["unmatched"] <= 8 and rq["matched"] >= 0.99 * sum(map(len, ref))


def test_e2e_bf16_headline_batch_smooth_family_vs_oracle():
    """The headline configuration in the mode the throughput is quoted in (bf16, throughput dispatch of the pipelined runner, one
    compiled graph: forward + fused decode + NMS with the key prefilter) on the smooth weight family against the f32 oracle on all 32
    images: the reference's AMP tolerance (0.5 px, utils/checks.py:780) on every matched row, detection sets equal outside the
    +-0.005 band around conf_thres."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.utils.nms import nms_raw
    x, y_ref = _oracle_full("yolov8n", 32, family="smooth:yolov8n")
    ref = [r.numpy() for r in onms.non_max_suppression(y_ref, 0.25, 0.7, max_det=300)]
    m = _build("yolov8n", torch.bfloat16, family="smooth:yolov8n")
    det = m.model[-1]
    det.keep_raw, det.nms_keys, det.concurrent = False, True, False
    xb = x.to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad(), _dispatch("throughput"):
        run = m.compile(xb, post=lambda o: nms_raw(o[0], 0.25, 0.7, max_det=300, key="full"))
        o_, c_, _ = run()
        torch.cuda.synchronize()
    oc, cc = o_.cpu().numpy(), c_.cpu().tolist()
    out = [oc[i, :int(cc[i])] for i in range(32)]
    a = detection_agreement(out, ref, 0.9)
    ref_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in ref]
    out_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in out]
    rec_x = detection_agreement(out, ref_x, 0.9)["recall"]
    prec_x = detection_agreement(out_x, ref, 0.9)["precision"]
    print(f"yolov8n smooth bf16 bs 32 vs oracle: {a['n_mine']} vs {a['n_ref']} rows, recall {a['recall']:.3f} precision {a['precision']:.3f} "
          f"(outside the band {rec_x:.4f} / {prec_x:.4f}), matched box max {a['box_max']:.3f} px score max {a['score_max']:.4f}")
    r9, p9 = SMOOTH_BOUNDS["yolov8n"]
    assert a["recall"] >= r9 and a["precision"] >= p9
    assert rec_x >= 0.995 and prec_x >= 0.995
    assert a["box_max"] <= 0.5 and a["score_max"] <= SMOOTH_BAND


def test_e2e_bf16_bot3_config_batch_smooth_family_vs_oracle():
    """Config 4 at ITS batch size in the mode its throughput is quoted in (yolov5-BoT3, 16 x 3 x 640 x 640, bf16: fused 6x6 stem, stacked
    C3 1x1 convs, MHSA on the matrix cores) on the smooth weight family against the f32 oracle on all 16 images: the reference's AMP
    tolerance (0.5 px, utils/checks.py:780) on every matched row, detection sets equal outside the +-0.005 band around conf_thres."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    fam = "smooth:yolov5-BoT3"
    x, y_ref = _oracle_full("yolov5-BoT3", 16, family=fam)
    ref = [r.numpy() for r in onms.non_max_suppression(y_ref, 0.25, 0.7, max_det=300)]
    m = _build("yolov5-BoT3", torch.bfloat16, family=fam)
    with torch.no_grad():
        y = m(x.to(DEV).to(torch.bfloat16).contiguous())[0]
        out = [o.cpu().numpy() for o in non_max_suppression(y, 0.25, 0.7, max_det=300)]
    a = detection_agreement(out, ref, 0.9)
    ref_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in ref]
    out_x = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in out]
    rec_x = detection_agreement(out, ref_x, 0.9)["recall"]
    prec_x = detection_agreement(out_x, ref, 0.9)["precision"]
    print(f"yolov5-BoT3 smooth bf16 bs 16 vs oracle: {a['n_mine']} vs {a['n_ref']} rows, recall {a['recall']:.3f} precision {a['precision']:.3f} "
          f"(outside the band {rec_x:.4f} / {prec_x:.4f}), matched box max {a['box_max']:.3f} px score max {a['score_max']:.4f}")
    r9, p9 = SMOOTH_BOUNDS["yolov5-BoT3"]
    assert a["recall"] >= r9 and a["precision"] >= p9
    assert rec_x >= 0.995 and prec_x >= 0.995
    assert a["box_max"] <= 0.5 and a["score_max"] <= SMOOTH_BAND


def test_e2e_bf16_rtdetr_config_batch_vs_oracle():
    """Config 5 at ITS batch size in the mode its throughput is quoted in (yolov3-rtdetr, 16 x 3 x 640 x 640, bf16 backbone + the
    decoder's perf mode: bf16-product linears, matrix-core self-attention, bf16 deformable-attention values) against the f32 oracle.
    The decoder picks its 300 queries by encoder score (head.py:2175), so the bf16 mode may pick other tokens near the cut: rows are
    matched as sets per image (nearest partner).  Reported: the fraction of oracle rows with a partner within the reference's AMP
    tolerance (0.5 px of 640 on the normalised box, utils/checks.py:780) and 0.01 in every class score; asserted: the measured level."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import rtdetr_postprocess
    fam = RTDETR_BF16_FAMILY
    x, y_ref, _ = _oracle_rtdetr_taps(16, fam)
    m = _build("yolov3-rtdetr", torch.bfloat16, family=fam)
    with torch.no_grad(), _dispatch("throughput"):
        y = m(x.to(DEV).to(torch.bfloat16).contiguous())[0]
    torch.cuda.synchronize()
    yc = y.float().cpu()
    fr, worst_box, worst_sc = [], 0.0, 0.0
    for i in range(yc.shape[0]):
        db = (yc[i][:, None, :4] - y_ref[i][None, :, :4]).abs().amax(2)   # (300 mine, 300 oracle)
        j = db.argmin(0)                                                   # my nearest row for every oracle row
        box = db.min(0).values
        sc = (yc[i][j, 4:] - y_ref[i][:, 4:]).abs().amax(1)
        ok = (box <= 0.5 / 640) & (sc <= 0.01)
        fr.append(float(ok.float().mean()))
        worst_box = max(worst_box, float(box[ok].max()) if ok.any() else 0.0)
        worst_sc = max(worst_sc, float(sc[ok].max()) if ok.any() else 0.0)
    mine = rtdetr_postprocess(y.float(), 0.25)
    ref = onms.rtdetr_postprocess(y_ref, 0.25)
    print(f"yolov3-rtdetr bf16 bs 16 vs oracle: oracle rows with a partner within 0.5 px / 0.01: mean {np.mean(fr):.4f} min {min(fr):.4f}; "
          f"among them max box {worst_box * 640:.3f} px score {worst_sc:.4f}; detections {sum(a.shape[0] for a in mine)} vs {sum(r.shape[0] for r in ref)}")
    # A REPORT since round 6 (the 0.10 gate it carried passed 90 % wrong rows): with its OWN query selection the bf16 mode decodes
    # mostly other tokens than the oracle - the 300 queries are the top of nearly flat random encoder scores - so this fraction
    # measures the selection's sensitivity, not the arithmetic.  The arithmetic is pinned by the two tests below, in front of and
    # behind the selection.  Asserted here: the detections the user sees are of the same number.
    n_mine, n_ref = sum(a.shape[0] for a in mine), sum(r.shape[0] for r in ref)
    assert 0.8 * n_ref <= n_mine <= 1.25 * n_ref + 4


RTDETR_BF16_FAMILY = "smooth:yolov3-rtdetr"


_RTDETR_ORACLE = {}


def _oracle_rtdetr_taps(batch, family):
    """The oracle's yolov3-rtdetr on `batch` images with the encoder-side tensors of `_get_decoder_input` (head.py:2143-2200) kept:
    enc_output features and enc_score_head logits of ALL tokens, the top-300 token indices (head.py:2175), and the encoder box of
    EVERY token, sigmoid(enc_bbox_head(features) + anchors) (head.py:2183-2185 applied to all 8400 tokens instead of the selected
    300: the same arithmetic, nothing selected).  Cached per (batch, family): one CPU forward of 16 x 257 GFLOP serves both tests."""
    key = (batch, family)
    if key in _RTDETR_ORACLE:
        return _RTDETR_ORACLE[key]
    o = ot.DetectionModel("yolov3-rtdetr.yaml")
    P.apply_procedural_weights(o, family=family)
    o.fuse()
    head = o.model[-1]
    kept = {}
    h1 = head.enc_output.register_forward_hook(lambda m, i, out: kept.__setitem__("features", out.detach()))
    h2 = head.enc_score_head.register_forward_hook(lambda m, i, out: kept.__setitem__("scores", out.detach()))
    x = P.synthetic_images(batch)
    with torch.no_grad():
        y = o(x)[0]
        h1.remove()
        h2.remove()
        sz = [640 // 8, 640 // 16, 640 // 32]
        anchors, valid = head._generate_anchors([[s, s] for s in sz])
        kept["topk"] = torch.topk(kept["scores"].max(-1).values, head.num_queries, dim=1).indices
        kept["enc_box"] = (head.enc_bbox_head(kept["features"]) + anchors).sigmoid()
        kept["valid"] = valid.view(-1)
    _RTDETR_ORACLE.clear()
    _RTDETR_ORACLE[key] = (x, y, kept)
    return _RTDETR_ORACLE[key]


# Gates of the two bf16 pins of config 5.  The reference's AMP self-check (utils/checks.py:780) allows 0.5 px between an fp32 and an
# fp16-autocast run: fp16 carries an 11-bit significand, bf16 an 8-bit one, so the same arithmetic in bf16 - the dtype BASELINE config 5
# is measured in - moves boxes 2^3 times as far.  Measured on MI355X (tools/experiments/r06_rtdetr_bf16_sources.py, bs 16, smooth family;
# gpurun_out -> profiles/r06_rtdetr_bf16_sources.txt): with ONLY the backbone in bf16 and the whole decoder in exact f32, 57 % of the
# decoder rows and 84 % of the encoder tokens are inside 0.5 px (row p50 0.42 / p99 1.75 / max 4.1 px); the quoted mode (bf16 projections,
# bf16-product linears, bf16 value rows) has 20 % / 71 % inside 0.5 px (row p50 0.89 / p99 3.0 / max 5.0 px), scores <= 0.005 either way.
# So the fp16 tolerance is out of reach for ANY bf16 backbone; the gate is the fp16 tolerance scaled by those three bits on the boxes
# (4 px of 640) and UNSCALED on the class probabilities (0.01) - and the fraction inside the unscaled 0.5 px is printed and floored at
# what was measured, so a regression of the arithmetic shows.  On the default (chaotic) weight family the decoder amplifies the
# backbone's bf16 noise to tens of pixels even on fixed queries (p99 35 px with an exact-f32 decoder too): a report, coarse floors only.
RTDETR_BOX_TOL_PX = 0.5 * 2 ** (11 - 8)
RTDETR_ENC_TOKENS_OK = 0.999   # encoder tokens inside (4 px, 0.01) on the smooth family; measured 1.0000
RTDETR_DEC_ROWS_OK = 0.99      # decoder rows on the oracle's queries inside (4 px, 0.01) on the smooth family; measured 0.9983
RTDETR_INSIDE_HALF_PX = {"enc": 0.65, "dec": 0.15}   # floors of the fractions inside the UNSCALED 0.5 px (measured 0.711 / 0.2015)


@pytest.mark.parametrize("family", [None, RTDETR_BF16_FAMILY])
def test_e2e_bf16_rtdetr_encoder_head_every_token_vs_oracle(family):
    """Config 5 in the mode its throughput is quoted in, pinned BEFORE the chaotic top-300 selection: the bf16 darknet53 backbone +
    bf16 input projections + enc_output / LayerNorm + enc_score_head + enc_bbox_head (head.py:2117-2172, 2183-2185) for EVERY one of
    the 16 x 8400 tokens against the f32 oracle - class probabilities sigmoid(enc_score logits) and encoder boxes
    sigmoid(enc_bbox_head(features) + anchors) in pixels of 640.  This comparison is token by token: no selection, nothing to flip."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.modules import rtdetr as RT
    x, _, ref = _oracle_rtdetr_taps(16, family)
    m = _build("yolov3-rtdetr", torch.bfloat16, family=family)
    head = m.model[-1]
    head.taps = {}
    with torch.no_grad(), _dispatch("throughput"):
        m(x.to(DEV).to(torch.bfloat16).contiguous())
        t = head.taps
        st, bs = t["static"], t["bs"]
        sc = head.level_major_to_image(t["enc_scores"], st, bs).float().cpu()
        # the encoder box of every token through the product's own enc_bbox_head in the perf mode's arithmetic (bf16 products)
        saved = RT._LINEAR_BF16[0]
        RT._LINEAR_BF16[0] = bool(head.linear_bf16)
        try:
            delta = head.enc_bbox_head(t["features"], key="all_tokens")
        finally:
            RT._LINEAR_BF16[0] = saved
        delta = head.level_major_to_image(delta, st, bs).float().cpu()
        torch.cuda.synchronize()
    head.taps = None
    anchors = st["anchors"].cpu().view(1, -1, 4)
    box = (delta + anchors).sigmoid()
    valid = ref["valid"]
    dp = (sc.sigmoid() - ref["scores"].sigmoid()).abs()                     # (16, 8400, 80) class probabilities
    db = (box - ref["enc_box"]).abs()[:, valid] * 640                       # (16, valid tokens, 4) px
    # invalid tokens (border anchors, head.py:2113): the anchor is +inf, the box 1.0 on both sides whatever the features are
    assert torch.equal(box[:, ~valid], ref["enc_box"][:, ~valid])
    tok_ok = ((dp.amax(2) <= 0.01)[:, valid] & (db.amax(2) <= 0.5)).float().mean(1)
    dl = (sc - ref["scores"]).abs()
    # the ranking statistic itself (max class logit per token) and how flat it is: the spread of the oracle's top-300 cut
    rank_ref = ref["scores"].max(-1).values
    cut = torch.topk(rank_ref, 300, dim=1).values
    print(f"yolov3-rtdetr bf16 encoder head, family {family}: tokens inside 0.5 px / 0.01: mean {tok_ok.mean():.5f} least image "
          f"{tok_ok.min():.5f}; class prob |d| max {dp.max():.5f} p99.9 {dp.flatten()[::7].quantile(0.999):.5f}; logit |d| max "
          f"{dl.max():.4f}; box |d| max {db.max():.4f} px p99.9 {db.flatten().quantile(0.999):.4f} px; oracle top-300 logit span "
          f"{(cut[:, 0] - cut[:, -1]).mean():.4f}, rank-stat |d| mean {(sc.max(-1).values - rank_ref).abs().mean():.4f}")
    tok_bf = ((dp.amax(2) <= 0.01)[:, valid] & (db.amax(2) <= RTDETR_BOX_TOL_PX)).float().mean(1)
    print(f"  inside ({RTDETR_BOX_TOL_PX} px, 0.01): mean {tok_bf.mean():.5f} least image {tok_bf.min():.5f}")
    if family is None:   # chaotic family: report + coarse floors (what a broken kernel would violate)
        assert dp.max().item() <= 0.05 and db.flatten()[::3].quantile(0.99).item() <= 4.0
        return
    assert tok_bf.mean().item() >= RTDETR_ENC_TOKENS_OK and tok_bf.min().item() >= RTDETR_ENC_TOKENS_OK - 0.002
    assert dp.max().item() <= 0.01 and db.flatten()[::3].quantile(0.99).item() <= 1.5
    assert tok_ok.mean().item() >= RTDETR_INSIDE_HALF_PX["enc"]


@pytest.mark.parametrize("family", [None, RTDETR_BF16_FAMILY])
def test_e2e_bf16_rtdetr_decoder_with_oracle_queries_vs_oracle(family):
    """Config 5's perf mode end to end with the ORACLE'S top-300 token indices injected (`RTDETRDecoder.query_override`, head.py:2175):
    bf16 backbone, bf16-product decoder (6 deformable layers, transformer.py:719-773), output (16, 300, 84) compared with the oracle's
    ROW BY ROW - same token in the same row on both sides, so no matching and no set semantics."""
    from tests.hip_utils import DEV
    x, y_ref, ref = _oracle_rtdetr_taps(16, family)
    m = _build("yolov3-rtdetr", torch.bfloat16, family=family)
    head = m.model[-1]
    head.query_override = ref["topk"]
    with torch.no_grad(), _dispatch("throughput"):
        y = m(x.to(DEV).to(torch.bfloat16).contiguous())[0]
    torch.cuda.synchronize()
    head.query_override = None
    yc = y.float().cpu()
    db = (yc[..., :4] - y_ref[..., :4]).abs().amax(2) * 640        # (16, 300) px
    ds = (yc[..., 4:] - y_ref[..., 4:]).abs().amax(2)
    ok = ((db <= 0.5) & (ds <= 0.01)).float().mean(1)
    print(f"yolov3-rtdetr bf16 decoder on the oracle's queries, family {family}: rows inside 0.5 px / 0.01: mean {ok.mean():.5f} least "
          f"image {ok.min():.5f}; box |d| max {db.max():.4f} px p99 {db.flatten().quantile(0.99):.4f}; score |d| max {ds.max():.5f} p99 "
          f"{ds.flatten().quantile(0.99):.5f}")
    ok_bf = ((db <= RTDETR_BOX_TOL_PX) & (ds <= 0.01)).float().mean(1)
    print(f"  inside ({RTDETR_BOX_TOL_PX} px, 0.01): mean {ok_bf.mean():.5f} least image {ok_bf.min():.5f}")
    assert torch.isfinite(yc).all()
    if family is None:   # chaotic family: the decoder amplifies the backbone's bf16 noise even on fixed queries - a report
        return
    assert ok_bf.mean().item() >= RTDETR_DEC_ROWS_OK and ok_bf.min().item() >= RTDETR_DEC_ROWS_OK - 0.02
    assert ds.max().item() <= 0.01 and db.flatten().quantile(0.99).item() <= 4.0
    assert ok.mean().item() >= RTDETR_INSIDE_HALF_PX["dec"]


def test_e2e_f32_bot3_config_batch_vs_oracle():
    """Config 4 at ITS batch size (yolov5-BoT3, 16 x 3 x 640 x 640), f32 vs the oracle: head output and post-NMS rows to 1e-3."""
    from tests.hip_utils import DEV, rows_equivalent, rows_identical
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x, y_ref = _oracle_full("yolov5-BoT3", 16)
    ref = [r.numpy() for r in onms.non_max_suppression(y_ref, 0.25, 0.7, max_det=300)]
    m = _build("yolov5-BoT3", torch.float32)
    with torch.no_grad():
        y = m(x.to(DEV))[0]
        out = [o.cpu().numpy() for o in non_max_suppression(y, 0.25, 0.7, max_det=300)]
    d = (y.cpu() - y_ref).abs()
    eq, rb, rs = rows_identical(out, ref, TOL)
    print(f"yolov5-BoT3 f32 bs 16 vs oracle: head max|box d| {d[:, :4].max():.3e} px max|score d| {d[:, 4:].max():.3e}; rows equal={eq}")
    rq = rows_equivalent(out, ref, TOL, 0.25, 0.7)
    print(f"  {rq}")
    assert d[:, :4].max().item() <= TOL and d[:, 4:].max().item() <= TOL
    assert rq["equivalent"] and rq["unmatched"] <= 8 and rq["matched"] >= 0.99 * sum(map(len, ref))


def test_e2e_f32_rtdetr_config_batch_vs_oracle():
    """Config 5 at ITS batch size (yolov3-rtdetr, 16 x 3 x 640 x 640), f32 vs the oracle: the (16, 300, 84) decoder output row by row
    (the top-300 query order must reproduce, head.py:2175) and the post-processed detections bit for bit given that output."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import rtdetr_postprocess
    x, y_ref = _oracle_full("yolov3-rtdetr", 16)
    m = _build("yolov3-rtdetr", torch.float32)
    with torch.no_grad():
        y = m(x.to(DEV))[0]
    torch.cuda.synchronize()
    yc = y.cpu()
    d = (yc - y_ref).abs()
    in_place = int((d.flatten(1).max(1).values <= TOL).sum())
    # `torch.topk` (head.py:2175) orders the 300 selected tokens by encoder score: two tokens whose scores agree to ~1e-6 may swap
    # places between two f32 implementations, and the decoder is equivariant to the order of its queries - the same 300 rows come out,
    # two of them in exchanged positions (measured on MI355X at bs 16: 6 of 16 images, exactly two rows each).  So the rows are
    # compared as a set per image: every oracle row has its own partner within 1e-3.
    worst = 0.0
    for i in range(yc.shape[0]):
        dist = (yc[i][:, None, :] - y_ref[i][None, :, :]).abs().amax(2)  # (300, 300)
        nearest = dist.argmin(1)
        assert len(set(nearest.tolist())) == yc.shape[1], f"image {i}: two rows share a partner"
        worst = max(worst, float(dist.min(1).values.max()))
    print(f"yolov3-rtdetr f32 bs 16 vs oracle: {in_place}/16 images identical in place (<= 1e-3); as sets: worst row-to-partner deviation "
          f"{worst:.3e} (normalised boxes, scores)")
    assert worst <= TOL and in_place >= 8
    # post-processing: bit-exact given the same decoder output; against the oracle's own output the row COUNT may differ by rows
    # whose score sits within 1e-3 of conf (reported, not asserted)
    mine = rtdetr_postprocess(y, 0.25)
    same = onms.rtdetr_postprocess(y.cpu(), 0.25)
    for a, r in zip(mine, same):
        assert torch.equal(a.cpu(), r)
    ref = onms.rtdetr_postprocess(y_ref, 0.25)
    print(f"  detections {sum(a.shape[0] for a in mine)} vs the oracle's {sum(r.shape[0] for r in ref)}")


def test_e2e_large_input_decode_index_split_vs_oracle():
    """1280 x 1280, batch 8 (204,800 anchors per image at stride 8): the fused decode epilogues split a flat pixel index into (image,
    anchor) with a multiply-high whose exactness the host checks (`upa_magic_exact`, csrc/detect_epi.h:107-117; shapes outside it
    take the unfused path).  Both numeric modes against the oracle: f32 to 1e-3 over every anchor; bf16 (fused branch tails + NMS key
    prefilter) per anchor to bf16 resolution - a wrong split would put whole images' anchors in the wrong rows."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    b, sz = 8, 1280
    x = P.synthetic_images(b, h=sz, w=sz)
    o = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(o)
    o.fuse()
    with torch.no_grad():
        y_ref = o(x)[0]
    assert y_ref.shape == (b, 84, 160 * 160 + 80 * 80 + 40 * 40)
    m = _build("yolov8n", torch.float32)
    with torch.no_grad():
        y = m(x.to(DEV))[0]
    d = (y.cpu() - y_ref).abs()
    print(f"1280 px bs 8 f32: max|box d| {d[:, :4].max():.3e} px max|score d| {d[:, 4:].max():.3e}")
    assert d[:, :4].max().item() <= 2e-3 and d[:, 4:].max().item() <= TOL  # boxes reach 1280 px: one more bit than at 640
    out = non_max_suppression(y, 0.25, 0.7)
    ref = onms.non_max_suppression(y_ref, 0.25, 0.7)
    assert [a.shape[0] for a in out] == [r.shape[0] for r in ref]
    del m, y
    mb = _build("yolov8n", torch.bfloat16)
    det = mb.model[-1]
    det.keep_raw, det.nms_keys = False, True
    with torch.no_grad():
        yb = mb(x.to(DEV).to(torch.bfloat16).contiguous())[0]
    db = (yb.float().cpu() - y_ref).abs()
    q = db[:, :4].flatten()[::11].quantile(0.99).item()
    print(f"1280 px bs 8 bf16: box |d| p99 {q:.3f} px max {db[:, :4].max():.3f}, score max {db[:, 4:].max():.4f}")
    assert q <= 4.0 and db[:, 4:].max().item() <= 0.05
    # anchor-exact placement: the box CENTRES of every image sit on their own anchors' grid cells (a mis-split image would carry
    # another image's centres: tens of pixels off)
    cx_ref, cx = y_ref[:, 0], yb.float().cpu()[:, 0]  # the decoded rows are (cx, cy, w, h, scores...)
    assert (cx - cx_ref).abs().flatten()[::11].quantile(0.999).item() <= 16.0


def test_e2e_throughput_dispatch_matches_default_dispatch():
    """The pipelined runner compiles its in-flight copies under `upa_opts {c2f: 4, conv_ws3: 1}` (engine/pipeline.py: the 40 x 40 C2f
    blocks as separate launches, conv_big instead of the persistent 3x3) while the serial legs and plain `model(x)` use the library
    defaults (whole-block c2f64, conv_ws3).  Both dispatches compute the same convolutions with the same bf16 rounding points; only
    f32 summation order differs.  End to end on the smooth family (where bf16 is meaningful): head outputs within the AMP
    tolerance of each other and the same detections outside the threshold band."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    m = _build("yolov8n", torch.bfloat16, family="smooth:yolov8n")
    m.model[-1].keep_raw = False
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        with R.use_opts(c2f64_max_px=0):  # the library default size rule (the test session's default lifts it)
            y_def = m(x)[0].float().clone()
        with R.use_opts(c2f=4, conv_ws3=1, c2f_stream_rows=-1, detect_stream=2, conv_big=2):
            y_thr = m(x)[0].float().clone()
        d_def = [o.cpu().numpy() for o in non_max_suppression(y_def, 0.25, 0.7, max_det=300)]
        d_thr = [o.cpu().numpy() for o in non_max_suppression(y_thr, 0.25, 0.7, max_det=300)]
    d = (y_def - y_thr).abs()
    a = detection_agreement(d_thr, d_def, 0.9)
    x_def = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in d_def]
    x_thr = [r[np.abs(r[:, 4] - 0.25) > SMOOTH_BAND] for r in d_thr]
    rec_x = detection_agreement(d_thr, x_def, 0.9)["recall"]
    prec_x = detection_agreement(x_thr, d_def, 0.9)["precision"]
    print(f"throughput vs default dispatch: head box max {d[:, :4].max():.3f} px score max {d[:, 4:].max():.4f}; rows {a['n_mine']} vs "
          f"{a['n_ref']}, recall {a['recall']:.3f} precision {a['precision']:.3f}, outside the band {rec_x:.4f} / {prec_x:.4f}")
    assert d[:, :4].max().item() <= 0.5 and d[:, 4:].max().item() <= SMOOTH_BAND
    assert rec_x >= 0.995 and prec_x >= 0.995 and a["box_max"] <= 0.5


# ---- bf16 pinned deterministically: HIP bf16 vs the rounding-point-exact CPU emulation (oracle/bf16_emul.py) ---------------------
EMUL_CASES = [("smooth:yolov8n", 2), ("smooth:yolov8n", 32), (None, 2), (None, 32)]


@pytest.mark.parametrize("family,batch", EMUL_CASES, ids=[f"{(f or 'procedural').split(':')[0]}-bs{b}" for f, b in EMUL_CASES])
def test_e2e_bf16_matches_rounding_point_emulation(family, batch):
    """The HIP bf16 mode against `oracle.bf16_emul.emulate_bf16`: the oracle with bf16 roundings at exactly the tensors the kernels
    round (input, folded weights, every Conv output, Bottleneck sums after the f32 add; Detect's last 1x1 and the decode in f32).
    What remains is f32 summation order inside a convolution and v_exp_f32 / v_rcp_f32: a value on a rounding boundary may come out
    one bf16 ulp away, and later layers see that.  Gates: per element of the (B, 84, A) head output, in bf16 ulps of the emulated
    value; and the detections as sets."""
    from oracle.bf16_emul import bf16_ulp, emulate_bf16
    from tests.hip_utils import DEV, detection_agreement, rows_equivalent
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x = P.synthetic_images(batch)
    o = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(o, family=family)
    o.eval()
    with torch.no_grad():
        y_em = emulate_bf16(o)(x)[0]
    ref = [r.numpy() for r in onms.non_max_suppression(y_em, 0.25, 0.7, max_det=300)]
    m = _build("yolov8n", torch.bfloat16, family=family)
    det = m.model[-1]
    det.keep_raw, det.nms_keys = False, True
    with torch.no_grad():
        y = m(x.to(DEV).to(torch.bfloat16).contiguous())[0]
        out = [t.cpu().numpy() for t in non_max_suppression(y, 0.25, 0.7, max_det=300)]
    yc = y.float().cpu()
    d = (yc - y_em).abs()
    u = d / bf16_ulp(y_em)
    ub, us = u[:, :4].flatten(), u[:, 4:].flatten()
    qs = lambda t: [float(t[::13].quantile(q)) for q in (0.5, 0.99, 0.999)]  # noqa: E731
    a = detection_agreement(out, ref, 0.9)
    rq = rows_equivalent(out, ref, 5e-3, 0.25, 0.7, iou_tol=5e-3)
    print(f"yolov8n {family or 'procedural'} bs {batch} HIP bf16 vs emulation: box |d| max {d[:, :4].max():.4f} px = {ub.max():.2f} bf16 ulp "
          f"(p50/p99/p99.9 {qs(ub)}); score |d| max {d[:, 4:].max():.5f} = {us.max():.2f} ulp (p50/p99/p99.9 {qs(us)}); within 1 ulp "
          f"{float((u <= 1).float().mean()):.5f}, within 2 ulp {float((u <= 2).float().mean()):.6f}; detections {a['n_mine']} vs {a['n_ref']} "
          f"recall {a['recall']:.4f} precision {a['precision']:.4f} matched box max {a['box_max']:.4f} score max {a['score_max']:.5f}; {rq}")
    # Measured on MI355X (round 4): smooth family bs 2 / 32: boxes <= 0.13 px (0.6 ulp), scores <= 0.0027 (4 ulp, p99.9 1.9 ulp), 99.92 /
    # 99.97 % of the elements within 2 ulp, every row without a partner excused by a threshold tie; the chaotic procedural family:
    # boxes p99.9 3.5 px, 96.9 % within 2 ulp - one-ulp flips (f32 summation order at a rounding boundary, ~1e-4 of the elements per
    # layer) are amplified by that family's weights exactly as the bf16 roundings themselves are.  The deviation from the emulation is
    # as large as the emulation's own deviation from f32 (tests/test_oracle_golden.py): in bf16 the rounding noise, not the rounding
    # points, is what separates two implementations, which is why the per-layer test below - not this one - is the bit-level pin.
    if family:
        assert float((u <= 2).float().mean()) >= 0.999 and us.max().item() <= 8 and d[:, :4].max().item() <= 0.25
        assert rq["unmatched"] == rq["explained_by_threshold_ties"] and rq["max_box_abs_px"] <= 0.05
        assert a["recall"] >= 0.93 and a["precision"] >= 0.93
    else:
        assert float((u <= 2).float().mean()) >= 0.95 and float(ub[::13].quantile(0.999)) <= 8
        assert a["recall"] >= 0.85 and a["precision"] >= 0.85


# per config: ({row: least fraction of bit-identical elements}, {row: least fraction within one floored bf16 ulp}); "rest" = every other
# row.  The first rows are the pin (a missed or misplaced rounding point drops them far below these bounds, a dropped channel to
# <= 1 - 1/C); later rows inherit the flips of the earlier ones as input noise.  BoT3: row 9 is the attention block (emulated since
# round 5: MHSA's q / k / v roundings and bf16 exponentials, oracle/bf16_emul.py).
# Measured on MI355X (round 5), bit-identical / within one ulp: yolov8n .99975 .9984 .994 .927 .844 .647 .534 ... .432 / ... .768;
# yolov5-BoT3 .99958 .9983 .9943 .965 .915 .678 .543 .472 .430 (SPPF) .446 (BoT3) ... .422 / ... .750; yolov3-tiny .99998 .99977 .9982
# .988 .949 .845 ...; yolov8s .99968 .9937 .980 .833 .698 .533 ... .453 / ... .777.
LAYER_BOUNDS = {
    "yolov8n": ({1: 0.999, 2: 0.995, 3: 0.985, 4: 0.88, "rest": 0.35}, {"rest": 0.70}),
    "yolov5-BoT3": ({1: 0.999, 2: 0.995, 3: 0.985, 4: 0.94, 5: 0.88, "rest": 0.35}, {"rest": 0.68}),
    "yolov3-tiny": ({0: 0.9999, 1: 0.9999, 2: 0.999, 3: 0.999, 4: 0.995, 5: 0.995, 6: 0.98, 7: 0.98, 8: 0.92, 9: 0.92, "rest": 0.35}, {"rest": 0.68}),
    "yolov8s": ({1: 0.999, 2: 0.985, 3: 0.97, 4: 0.78, "rest": 0.35}, {"rest": 0.70}),
}


@pytest.mark.parametrize("name", ["yolov8n", "yolov5-BoT3", "yolov3-tiny", "yolov8s"])
def test_e2e_bf16_layer_by_layer_vs_rounding_point_emulation(name):
    """Where the two bf16 implementations part: every layer output of ONE full HIP bf16 forward (yolov8n, smooth family, bs 2: fused
    stem, whole-block C2f kernels, virtual Upsample + Concat) against the same layer of the CPU emulation (oracle/bf16_emul.py).
    Both store bf16 at the same tensors, so the first layers must agree BIT FOR BIT except where an f32 sum lands on a rounding
    boundary (a one-ulp flip); the later layers inherit those flips as input noise.  The table this prints is the deterministic
    statement about the bf16 mode: fraction of bit-identical elements and the largest deviation in bf16 ulps, per layer."""
    from oracle.bf16_emul import bf16_ulp, emulate_bf16
    from tests.hip_utils import DEV
    fam = "smooth:" + name
    x = P.synthetic_images(2)
    o = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(o, family=fam)
    o.eval()
    e = emulate_bf16(o)
    em = {}
    for mod in e.model:
        mod.register_forward_hook(lambda m_, _i, out: em.__setitem__(m_.i, out) if torch.is_tensor(out) else None)
    with torch.no_grad():
        e(x)
    m = _build(name, torch.bfloat16, family=fam)
    m.model[-1].keep_raw = False
    m.fuse_pool = False  # (yolov3-tiny: Conv + MaxPool pairs as separate launches so that every row's output exists; the fused pairs are
    hip = {}             # bit-identical to them, test_e2e_yolov3_tiny_conv_pool_fusion_is_exact)
    m.fuse_down = False  # (yolov8n rows 2-3 as two launches for the same reason; the one-launch form's row 3 is compared below)

    def grab(i):
        return lambda _m, _i, out: hip.__setitem__(i, out.float().cpu()) if torch.is_tensor(out) else None
    for mod in m.model:
        mod.register_forward_hook(grab(mod.i))
    # rows 0-1 run as one fused kernel (no module call): their result is what row 2 receives
    m.model[2].register_forward_pre_hook(lambda _m, inp: hip.__setitem__(1, inp[0].float().cpu()))
    with torch.no_grad():
        m(x.to(DEV).to(torch.bfloat16).contiguous())
    torch.cuda.synchronize()
    rows = {}
    for i in sorted(set(hip) & set(em)):
        a, b = hip[i], em[i]
        if a.shape != b.shape:  # (yolov3-tiny row 11: the product runs nn.ZeroPad2d inside the MaxPool that follows it)
            continue
        same = float((a == b).float().mean())
        # one bf16 ulp of the element, floored at the ulp of 1/64 of the tensor's largest value (SiLU outputs near zero carry huge
        # RELATIVE differences that mean nothing: -4.2e-6 against -4.6e-6)
        ulp = bf16_ulp(b.abs().clamp_min(float(b.abs().max()) / 64))
        u = (a - b).abs() / ulp
        rows[i] = (same, float((u <= 1).float().mean()), float(u.max()))
        print(f"  layer {i:2d} {type(m.model[i]).__name__:8s} bit-identical {same:.5f}  within 1 ulp {rows[i][1]:.5f}  max {rows[i][2]:.1f} ulp")
    # Measured on MI355X (round 4): bit-identical 0.99975 (fused stem) -> 0.9984 (model.2) -> 0.994 -> 0.927 (model.4) -> 0.84 -> 0.65 ->
    # 0.53 -> ~0.45 from the SPPF on; within one ulp 0.99995 -> 0.9997 -> 0.9987 -> 0.98 -> 0.96 -> 0.90 -> ~0.77-0.80.  The first layers are
    # the pin: a missed or misplaced rounding point in the stem / model.2 / model.3 kernels would drop them far below these bounds.
    # (yolov5's 6x6 first conv sums 108 products per output: 99.958 % bit-identical, largest deviation 5.5 floored ulps)
    lo_id, lo_ulp = LAYER_BOUNDS[name]
    for i, (same, within, _mx) in rows.items():
        assert same >= lo_id.get(i, lo_id["rest"]) and within >= lo_ulp.get(i, lo_ulp["rest"]), (name, i, same, within)
    first = min(rows)
    assert rows[first][2] <= (4.0 if name == "yolov8n" else 8.0), "the first rows must reproduce the emulation up to boundary flips"
    if name == "yolov8n":  # rows 2-3 as ONE launch (upa_c2f16_down_fused): row 3 against the emulation, same bound as the two-launch row 3
        m.fuse_down = True
        got = []
        f = m.model[2].forward_down
        m.model[2].forward_down = lambda xx, dn, out=None: (got.append(f(xx, dn, out=out)), got[-1])[1]
        with torch.no_grad():
            m(x.to(DEV).to(torch.bfloat16).contiguous())
        torch.cuda.synchronize()
        assert len(got) == 1 and got[0] is not None, "rows 2-3 did not take the one-launch form"
        a, b = got[0].float().cpu(), em[3]
        same = float((a == b).float().mean())
        u = (a - b).abs() / bf16_ulp(b.abs().clamp_min(float(b.abs().max()) / 64))
        print(f"  layer  3 (rows 2-3 as one launch) bit-identical {same:.5f}  within 1 ulp {float((u <= 1).float().mean()):.5f}  max {float(u.max()):.1f} ulp")
        assert same >= lo_id.get(3, lo_id["rest"]) and float((u <= 1).float().mean()) >= lo_ulp.get(3, lo_ulp["rest"]), (same, float(u.max()))
        assert abs(same - rows[3][0]) <= 0.003, (same, rows[3][0])


def test_e2e_yolov3_tiny_conv_pool_fusion_is_exact():
    """yolov3-tiny in bf16: rows 0-7 alternate Conv and nn.MaxPool2d(2, 2, 0); `BaseModel.fuse_pool` runs each such pair whose conv
    output nobody else reads as one launch (rows 0-1, 2-3, 4-5, 6-7; row 8 feeds the route, so 8-9 stay apart).  Same arithmetic,
    same rounding points: the decoded head output must be bit-identical with the switch off, eagerly and in a replayed hipGraph."""
    from tests.hip_utils import DEV
    m = _build("yolov3-tiny", torch.bfloat16)
    x = P.synthetic_images(3).to(DEV).to(torch.bfloat16).contiguous()
    calls = []
    for row in (0, 2, 4, 6, 8):
        f = m.model[row].forward_pool2
        m.model[row].forward_pool2 = (lambda xx, f=f, row=row: (calls.append(row), f(xx))[1])
    try:
        with torch.no_grad():
            y_fused = m(x)[0].clone()
            run = m.compile(x)
            y_graph = run()[0].clone()
            type(m).fuse_pool = False
            y_plain = m(x)[0].clone()
    finally:
        type(m).fuse_pool = True
    torch.cuda.synchronize()
    assert sorted(set(calls)) == [0, 2, 4, 6], calls  # row 8's output is saved for the route: not offered
    assert torch.equal(y_fused, y_plain) and torch.equal(y_graph, y_plain)


@pytest.mark.parametrize("name", ["yolov8n", "yolov5-BoT3", "yolov3-tiny", "yolov8s"])  # (tiny / yolov8s: class tails on the streaming 1x1 kernel)
def test_e2e_bf16_keys_only_head_is_exact(name):
    """`Detect.scores_out = False` (`upa_opts.keys_only`): the fused class tails write ONLY every anchor's best-class NMS key - no
    (B, nc, A) score rows, one sigmoid per anchor instead of nc (the sigmoid of the largest logit when it is separated from the
    runner-up for certain, else all of them as before).  Keys, box rows and detections must be BIT-identical to the score-writing
    form at any confidence threshold, eagerly and in a replayed graph."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build(name, torch.bfloat16)
    det = m.model[-1]
    det.keep_raw, det.nms_keys, det.concurrent = False, True, False
    x = P.synthetic_images(5).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        det.scores_out = True
        y_full = m(x)[0]
        keys_full = y_full._upa_hot.clone()
        full = {c: [t.clone() for t in nms_raw(y_full, c, 0.7, key=("full", c))] for c in (0.25, 0.05, 0.6)}
        y_full = y_full.clone()
        det.scores_out = False
        y_keys = m(x)[0]
        assert torch.equal(y_keys._upa_hot, keys_full), "best-class keys differ"
        assert torch.equal(y_keys[:, :4], y_full[:, :4]), "box rows differ"
        for c in (0.25, 0.05, 0.6):
            for a, b in zip(full[c], nms_raw(y_keys, c, 0.7, key=("keys", c))):
                assert torch.equal(a, b)
        # a reader of the class rows (multi-label NMS: the validator's mode, engine/validator.py) must refuse the keys-only output
        # loudly - rows 4.. of it were never written - and accept the next score-writing forward again
        from ultralytics_pro_amd._lib import UpaError
        with pytest.raises(UpaError, match="keys-only"):
            nms_raw(y_keys, 0.001, 0.7, multi_label=True, key="ml")
        det.scores_out = True
        nms_raw(m(x)[0], 0.001, 0.7, multi_label=True, key="ml")
        det.scores_out = False
        m(x)
        run = m.compile(x, post=lambda o: nms_raw(o[0], 0.25, 0.7, key="graph"))
        for _ in range(2):
            out = run()
            torch.cuda.synchronize()
            for a, b in zip(full[0.25], out):
                assert torch.equal(a, b)
    det.scores_out = True


def test_e2e_detect_level_stream_follows_the_tile_form():
    """The opt-in line-buffer form of the 80 x 80 Detect level (`upa_opts.detect_stream = 2`, csrc/detect_stream.hip) inside the whole yolov8n
    step, against the default tile form on the same model and images (smooth family, bf16, batch 4): same rounding points, another f32 summation
    order inside the 3x3 convs - head outputs equal up to flipped bf16 ties of the two intermediates, detections the same."""
    from tests.hip_utils import DEV, detection_agreement
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    m = _build("yolov8n", torch.bfloat16, family="smooth:yolov8n")
    m.model[-1].concurrent = False
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    outs = {}
    for mode in (0, 2):
        with torch.no_grad(), _dispatch("throughput"):
            from ultralytics_pro_amd.engine import runtime as R
            with R.use_opts(detect_stream=mode if mode else 1):  # (1 = refuse: the tile form)
                y = m(x)[0]
                outs[mode] = (y.float().cpu().clone(), [o.cpu().numpy() for o in non_max_suppression(y, 0.25, 0.7, max_det=300)])
    a6400 = 80 * 80
    d = (outs[2][0] - outs[0][0]).abs()
    assert d[:, :, a6400:].max().item() == 0.0                 # the other levels run the same launches
    lvl = d[:, :, :a6400]
    flips = (lvl[:, 4:] > 1e-6).float().mean().item()
    print(f"detect level stream vs tile form: level-0 box |d| max {lvl[:, :4].max():.4f} px, score |d| max {lvl[:, 4:].max():.5f}, "
          f"scores moved {flips:.4f}")
    assert lvl[:, :4].max().item() <= 0.25 and lvl[:, 4:].max().item() <= 5e-3
    a = detection_agreement(outs[2][1], outs[0][1], 0.9)
    assert a["recall"] >= 0.97 and a["precision"] >= 0.97 and a["box_max"] <= 0.25
