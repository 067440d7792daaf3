"""-m gpu: weights interchange on the HIP path.  Reference-keyed checkpoints - plain f32, half
(`model.half()`, tasks.py:2291-2406) and fused (`model.fuse()`: conv.weight = W', conv.bias = b', no bn.* keys) - are
loaded with `load_weights()` into a product model that ALREADY ran with other weights (so every packed-weight cache must
be dropped) and must reproduce the reference golden detections of tests/golden/e2e_yolov8n.npz (f32 mode, 1e-3)."""

import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import tasks as ot
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.utils.weights import load_weights, save_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _product_with_other_weights():
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m, seed=11)  # NOT the golden weights
    m = m.to(DEV).eval()
    with torch.no_grad():
        m(P.synthetic_images(1, h=64, w=64).to(DEV))  # warm-up: packs and caches every conv's folded weights
    torch.cuda.synchronize()
    return m


def _check_against_golden(m, g):
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    x = P.synthetic_images(2).to(DEV)
    with torch.no_grad():
        y = m(x)[0]
    torch.cuda.synchronize()
    d = np.abs(y.cpu()[:, :, g["anchor_sel"]].numpy() - g["y_sel"])
    assert d[:, :4].max() <= TOL and d[:, 4:].max() <= TOL, (d[:, :4].max(), d[:, 4:].max())
    out = non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    rows = torch.cat(out, 0).cpu().numpy()
    assert np.abs(rows[:, :5] - g["predict_rows"][:, :5]).max() <= TOL
    assert np.array_equal(rows[:, 5], g["predict_rows"][:, 5])


def _source():
    src = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(src)  # the weights the goldens were captured with
    return src


def test_plain_checkpoint_reproduces_golden(tmp_path, golden_dir):
    g = np.load(golden_dir / "e2e_yolov8n.npz")
    save_state_dict(_source(), tmp_path / "w.pt")
    m = _product_with_other_weights()
    rep = load_weights(m, tmp_path / "w.pt", strict=True)
    assert rep["loaded"] == rep["total"]
    _check_against_golden(m, g)


def test_fused_checkpoint_reproduces_golden(tmp_path, golden_dir):
    g = np.load(golden_dir / "e2e_yolov8n.npz")
    src = _source()
    src.fuse()
    sd = src.state_dict()
    assert "model.0.bn.weight" not in sd and "model.0.conv.bias" in sd
    torch.save({"epoch": -1, "model": dict(sd)}, tmp_path / "fused.pt")
    m = _product_with_other_weights()
    rep = load_weights(m, tmp_path / "fused.pt")
    assert not rep["unexpected"]
    _check_against_golden(m, g)


def test_half_checkpoint_matches_oracle_with_half_rounded_weights(tmp_path):
    """A `model.half()` checkpoint: weights carry 11 significant bits, so the golden (f32 weights) is not the expectation;
    the oracle run with the same half-rounded weights is."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.utils.nms import non_max_suppression
    src = _source()
    half = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in src.state_dict().items()}
    torch.save({"ema": half}, tmp_path / "half.pt")
    src.load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in half.items()})
    src.fuse()
    x = P.synthetic_images(2)
    with torch.no_grad():
        y_ref = src(x)[0]
    m = _product_with_other_weights()
    rep = load_weights(m, tmp_path / "half.pt")
    assert rep["loaded"] == rep["total"]
    with torch.no_grad():
        y = m(x.to(DEV))[0]
    torch.cuda.synchronize()
    d = (y.cpu() - y_ref).abs()
    assert d[:, :4].max().item() <= TOL and d[:, 4:].max().item() <= TOL
    out = non_max_suppression(y, 0.25, 0.7)
    ref = onms.non_max_suppression(y_ref, 0.25, 0.7)
    assert [a.shape[0] for a in out] == [r.shape[0] for r in ref]


def test_load_state_dict_invalidates_every_packed_cache():
    """a bare `model.load_state_dict()` (not `load_weights`) after a warm-up forward must not leave stale packed
    weights anywhere - Conv, the Detect head's final 1x1 convs, MHSA q/k/v (yolov5-BoT3), the packed nn.Linear weights of
    MLP / MSDeformAttn / the decoder layers and the input projections of RTDETRDecoder (yolov3-rtdetr)."""
    from tests.hip_utils import DEV
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    for name in ("yolov8n", "yolov5-BoT3", "yolov3-rtdetr"):
        a = DetectionModel(name + ".yaml")
        P.apply_procedural_weights(a, seed=11)
        a = a.to(DEV).eval()
        x = P.synthetic_images(1, h=320, w=320).to(DEV)
        with torch.no_grad():
            a(x)
        b = DetectionModel(name + ".yaml")
        P.apply_procedural_weights(b)
        a.load_state_dict(b.state_dict())
        b = b.to(DEV).eval()
        with torch.no_grad():
            ya, yb = a(x)[0].clone(), b(x)[0].clone()
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), name
