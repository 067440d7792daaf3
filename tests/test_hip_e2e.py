"""-m gpu: whole hot path (model forward -> Detect decode -> NMS) on the HIP path vs the oracle and the committed
reference goldens (B=2 synthetic 640x640).  Tolerance from BASELINE.json north_star: 1e-3 on boxes/scores (f32 mode)."""

import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import tasks as ot
from ultralytics_pro_amd.utils import procedural as P

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _build(name, dtype):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m)
    m = m.to(DEV).eval()
    m.set_compute_dtype(dtype)
    return m


@pytest.mark.parametrize("name", ["yolov8n", "yolov3-tiny", "yolov5-BoT3"])
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
    assert d[:, :4].max() <= TOL and d[:, 4:].max() <= TOL
    out = non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    rows = torch.cat(out, 0).cpu().numpy()
    assert np.abs(rows[:, :5] - g["predict_rows"][:, :5]).max() <= TOL
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


def test_e2e_bf16_agrees_with_f32_detections():
    """Perf mode: bf16 storage cannot hold 1e-3 on 640-px boxes; gate on the reference's own AMP tolerance
    (utils/checks.py:780, atol 0.5) for matched detections and on detection-set agreement."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x = P.synthetic_images(2).to(DEV)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _build("yolov8n", dt)
        with torch.no_grad():
            y = m(x if dt == torch.float32 else x.to(torch.bfloat16))[0]
        outs[dt] = (y.cpu(), [o.cpu() for o in non_max_suppression(y, 0.25, 0.7)])
    y32, y16 = outs[torch.float32][0], outs[torch.bfloat16][0]
    dbox = (y32[:, :4] - y16[:, :4]).abs()
    dscore = (y32[:, 4:] - y16[:, 4:]).abs().max().item()
    print(f"bf16 vs f32: box median|d|={dbox.median().item():.3f} p99={dbox.flatten().quantile(0.99).item():.3f} "
          f"max={dbox.max().item():.3f} px; score max|d|={dscore:.4f}")
    assert dscore <= 0.05
    assert dbox.flatten().quantile(0.99).item() <= 4.0
    for a, b in zip(outs[torch.float32][1], outs[torch.bfloat16][1]):
        assert abs(a.shape[0] - b.shape[0]) <= max(3, int(0.15 * a.shape[0]))


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
    two concurrent sub-batches) - every copy reproduces the single-graph detections bit for bit, repeatedly."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.bfloat16)
    x = P.synthetic_images(4).to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
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
    from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
    from ultralytics_pro_amd.utils.nms import nms_raw
    m = _build("yolov8n", torch.bfloat16)
    x32 = P.synthetic_images(32).to(DEV)
    x = x32.to(torch.bfloat16).contiguous()
    with torch.no_grad():
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
        iou = _iou_matrix(d[:, :4])
        same = d[:, 5][:, None] == d[:, 5][None, :]
        iou = torch.where(same, iou, torch.zeros_like(iou)).triu(1)
        assert iou.max().item() <= 0.7 + 1e-6, "two kept boxes of one class overlap by more than iou_thres"
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
