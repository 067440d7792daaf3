"""Validation metrics: oracle vs the reference goldens on CPU; product (GPU IoU kernel + host AP)
vs the same goldens and end-to-end mAP of the HIP pipeline on the synthetic set."""

import numpy as np
import pytest
import torch

from oracle import metrics as omet
from ultralytics_pro_amd.utils import procedural as P


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "map_yolov8n.npz")


def _stats(G, process_batch, to=lambda t: t):
    tps, confs, pcls, tcls = [], [], [], []
    for i in range(4):
        d = torch.from_numpy(G[f"det{i}"])
        gb, gc = torch.from_numpy(G[f"gt_boxes{i}"]), torch.from_numpy(G[f"gt_cls{i}"])
        tp = process_batch(to(d[:, :4].contiguous()), to(d[:, 5].contiguous()), to(gb), to(gc))
        assert np.array_equal(tp, G[f"tp{i}"]), f"TP matrix of image {i} differs from the reference"
        tps.append(tp); confs.append(d[:, 4].numpy()); pcls.append(d[:, 5].numpy()); tcls.append(gc.numpy())
    return tuple(np.concatenate(v, 0) for v in (tps, confs, pcls, tcls))


def test_oracle_metrics_match_reference(G):
    tp, conf, pc, tc = _stats(G, omet.process_batch)
    p, r, f1, ap, uc = omet.ap_per_class(tp, conf, pc, tc)
    assert np.array_equal(ap, G["ap"]) and np.array_equal(p, G["p"]) and np.array_equal(r, G["r"])
    assert np.array_equal(uc, G["classes"])
    assert np.allclose(omet.mean_results(p, r, ap), G["mean"], rtol=0, atol=1e-12)
    # known answer: a perfect detector has AP 1 at every threshold
    tp1 = np.ones((5, 10), bool)
    _, _, _, ap1, _ = omet.ap_per_class(tp1, np.linspace(0.9, 0.5, 5), np.zeros(5), np.zeros(5))
    assert np.allclose(ap1, 1.0, atol=1e-2)  # 101-point interpolation


@pytest.mark.gpu
def test_product_metrics_match_reference(G):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils import metrics as pmet
    tp, conf, pc, tc = _stats(G, pmet.process_batch, to=lambda t: t.to(DEV))
    p, r, f1, ap, uc = pmet.ap_per_class(tp, conf, pc, tc)
    assert np.array_equal(ap, G["ap"]) and np.array_equal(p, G["p"]) and np.array_equal(r, G["r"])
    assert np.allclose(pmet.mean_results(p, r, ap), G["mean"], rtol=0, atol=1e-12)
    # empty edge cases (val.py:284-285)
    e = torch.zeros((0, 4), device=DEV)
    assert pmet.process_batch(e, torch.zeros(0, device=DEV), e, torch.zeros(0, device=DEV)).shape == (0, 10)
    assert pmet.box_iou(e, torch.zeros((3, 4), device=DEV)).shape == (0, 3)


@pytest.mark.gpu
def test_hip_pipeline_map_matches_reference(G):
    """Model forward + val-mode NMS on the HIP path, scored against the synthetic ground truth: mAP equals the
    reference's (detections differ by <= 1e-3 px, which can only move a match sitting exactly on an IoU threshold)."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import metrics as pmet
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m = m.to(DEV).eval()
    with torch.no_grad():
        y = m(P.synthetic_images(4).to(DEV))[0]
    dets = non_max_suppression(y, conf_thres=0.001, iou_thres=0.7, max_det=300, multi_label=True)
    tps, confs, pcls, tcls = [], [], [], []
    for i, d in enumerate(dets):
        gb, gc = torch.from_numpy(G[f"gt_boxes{i}"]).to(DEV), torch.from_numpy(G[f"gt_cls{i}"]).to(DEV)
        assert d.shape[0] == G[f"det{i}"].shape[0]
        tps.append(pmet.process_batch(d[:, :4].contiguous(), d[:, 5].contiguous(), gb, gc))
        confs.append(d[:, 4].cpu().numpy()); pcls.append(d[:, 5].cpu().numpy()); tcls.append(gc.cpu().numpy())
    tp, conf, pc, tc = (np.concatenate(v, 0) for v in (tps, confs, pcls, tcls))
    p, r, f1, ap, uc = pmet.ap_per_class(tp, conf, pc, tc)
    got = np.array(pmet.mean_results(p, r, ap))
    print("HIP mAP:", got, "reference:", G["mean"])
    assert np.abs(got - G["mean"]).max() <= 5e-3


@pytest.mark.gpu
def test_validator_single_rank_equals_reference_map(G):
    """engine.validator.DetectionValidator on one GPU: model -> val-mode NMS -> batched matching kernel -> statistics ->
    class metrics, one batch of 4 images and the same images as two batches of 2: the TP matrices equal the reference's
    for its own detections, and the pipeline's mAP equals tests/golden/map_yolov8n.npz."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.engine.validator import DetectionValidator
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import metrics as pmet
    # (1) the kernel on the reference's own detections: TP matrices bit for bit, batched with ragged label counts
    det = torch.zeros(4, 300, 6)
    gt = torch.zeros(4, 64, 5)
    cnt, ngt = [], []
    for i in range(4):
        d, gb, gc = torch.from_numpy(G[f"det{i}"]), torch.from_numpy(G[f"gt_boxes{i}"]), torch.from_numpy(G[f"gt_cls{i}"])
        det[i, : d.shape[0]] = d
        gt[i, : gc.shape[0], 0] = gc
        gt[i, : gc.shape[0], 1:] = gb
        cnt.append(d.shape[0]); ngt.append(gc.shape[0])
    tp = pmet.match_predictions_batched(det.to(DEV), torch.tensor(cnt, dtype=torch.int32, device=DEV), gt.to(DEV),
                                        torch.tensor(ngt, dtype=torch.int32, device=DEV)).cpu().numpy().astype(bool)
    for i in range(4):
        assert np.array_equal(tp[i, : cnt[i]], G[f"tp{i}"]) and not tp[i, cnt[i]:].any()
    # (2) the whole validate path
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m = m.to(DEV).eval()
    x = P.synthetic_images(4).to(DEV)
    gtd, ngd = gt.to(DEV), torch.tensor(ngt, dtype=torch.int32, device=DEV)
    for split in ((0, 4),), ((0, 2), (2, 4)):
        v = DetectionValidator(m)
        with torch.no_grad():
            for a, b in split:
                v.update(m(x[a:b].contiguous()), gtd[a:b].contiguous(), ngd[a:b].contiguous())
        st = v.get_stats()
        got = np.array(st["mean"])
        print("validator mAP:", got, "reference:", G["mean"])
        assert np.abs(got - G["mean"]).max() <= 5e-3
        assert st["tp"].shape[0] == sum(cnt)
