"""CPU (-m "not gpu"): the PRODUCT's `parse_model` / `DetectionModel` (ultralytics_pro_amd/nn/tasks.py) reproduces the
builder tables captured from the imported reference (tests/golden/builder_*.json, written by oracle/gen_golden.py) for
all five configs of BASELINE.json: state_dict keys and shapes in order, the per-layer (i, f, type,
parameter count) table (tasks.py:3121-3134), the save list, the strides and the parameter total.  No GPU needed: the
product's modules are parameter containers until `forward`."""

import json

import pytest

CONFIGS = ["yolov8n", "yolov8s", "yolov3-tiny", "yolov5-BoT3", "yolov3-rtdetr"]


@pytest.mark.parametrize("name", CONFIGS)
def test_product_builder_matches_reference(name, golden_dir):
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    g = json.loads((golden_dir / f"builder_{name}.json").read_text())
    m = DetectionModel(name + ".yaml")
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g["state_dict"]
    table = [dict(i=l.i, f=l.f, type=l.type.split(".")[-1], np=int(sum(p.numel() for p in l.parameters())))
             for l in m.model]
    assert table == g["layers"]
    assert list(m.save) == g["save"]
    assert [float(s) for s in m.stride] == g["stride"]
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]


def test_product_refuses_cpu_forward():
    """The product path has no CPU fallback: a CPU tensor raises instead of silently running torch ops."""
    import torch

    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8n.yaml")
    with pytest.raises(L.UpaError):
        m(torch.zeros(1, 3, 64, 64))
