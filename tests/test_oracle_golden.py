"""CPU (-m "not gpu"): the oracle restatement reproduces the golden vectors that oracle/gen_golden.py captured from the
imported reference.  Inputs are regenerated from the counter hash; expected outputs are the fixtures."""

import json

import numpy as np
import pytest
import torch

from oracle import modules as om
from oracle import nms as onms
from oracle import tasks as ot
from ultralytics_pro_amd.utils import procedural as P

CONFIGS = ["yolov8n", "yolov8s", "yolov3-tiny", "yolov5-BoT3", "yolov3-rtdetr"]


def unit_input(name, shape, lo=-1.0, hi=1.0):
    return P.uniform(f"unit:{name}", shape, lo, hi)


def bn_fix(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return m.eval()


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "ops_unit.npz")


def test_hash_is_pinned():
    v = P.hash_uniform("abc", 4)
    assert [int(round(float(x) * (1 << 24))) for x in v] == [14048116, 430347, 11779619, 16698018]


@pytest.mark.parametrize("name", CONFIGS)
def test_builder_matches_reference(name, golden_dir):
    g = json.loads((golden_dir / f"builder_{name}.json").read_text())
    m = ot.DetectionModel(name + ".yaml")
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g["state_dict"]
    table = [dict(i=l.i, f=l.f, type=l.type.split(".")[-1], np=int(sum(p.numel() for p in l.parameters())))
             for l in m.model]
    assert table == g["layers"]
    assert list(m.save) == g["save"]
    assert [float(s) for s in m.stride] == g["stride"]
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]


UNIT = [
    ("conv_k1", om.Conv, (16, 32, 1, 1), (2, 16, 12, 12)),
    ("conv_k3s1", om.Conv, (16, 32, 3, 1), (2, 16, 12, 12)),
    ("conv_k3s2", om.Conv, (16, 32, 3, 2), (2, 16, 13, 13)),
    ("conv_k6s2p2", om.Conv, (3, 16, 6, 2, 2), (2, 3, 20, 20)),
    ("conv_stem", om.Conv, (3, 16, 3, 2), (2, 3, 16, 16)),
    ("bottleneck", om.Bottleneck, (16, 16, True, 1, (3, 3), 1.0), (2, 16, 10, 10)),
    ("bottleneck_noadd", om.Bottleneck, (16, 32, False), (2, 16, 10, 10)),
    ("c2f_n2", om.C2f, (32, 32, 2, True), (2, 32, 10, 10)),
    ("c2f_n1_noshortcut", om.C2f, (48, 32, 1, False), (2, 48, 10, 10)),
    ("c3_n1", om.C3, (32, 32, 1, True), (2, 32, 10, 10)),
    ("sppf", om.SPPF, (32, 32, 5), (2, 32, 9, 11)),
    ("mhsa", om.MHSA, (32, 6, 6, 4), (2, 32, 6, 6)),
    ("bot3", om.BoT3, (32, 32, 1, 0.5, 1, 6, 6), (2, 32, 6, 6)),
    ("mlp", om.MLP, (16, 32, 4, 3), (2, 10, 16)),
]


@pytest.mark.parametrize("name,cls,args,xshape", UNIT, ids=[u[0] for u in UNIT])
def test_unit_modules(name, cls, args, xshape, G):
    m = bn_fix(cls(*args))
    P.apply_procedural_weights(m, family="default")
    x = unit_input(name, xshape)
    with torch.no_grad():
        y = m(x)
    assert np.abs(y.numpy() - G[name]).max() <= 1e-6
    if name.startswith("conv_"):
        m.conv = om.fuse_conv_and_bn(m.conv, m.bn)
        with torch.no_grad():
            yf = m.forward_fuse(x)
        assert np.array_equal(m.conv.weight.numpy(), G[name + "_fused_w"])
        assert np.array_equal(m.conv.bias.numpy(), G[name + "_fused_b"])
        assert np.abs(yf.numpy() - G[name + "_fused"]).max() <= 1e-6


def test_unit_functions(G):
    a, b = unit_input("up_a", (2, 8, 5, 5)), unit_input("up_b", (2, 4, 10, 10))
    y = om.Concat(1)([torch.nn.Upsample(None, 2, "nearest")(a), b])
    assert np.array_equal(y.numpy(), G["upsample_concat"])
    assert np.abs(om.DFL(16)(unit_input("dfl", (2, 64, 21), -4, 4)).detach().numpy() - G["dfl"]).max() <= 1e-6
    feats = [torch.zeros(1, 1, 80, 80), torch.zeros(1, 1, 40, 40), torch.zeros(1, 1, 20, 20)]
    ar, sr = om.make_anchors(feats, torch.tensor([8.0, 16.0, 32.0]), 0.5)
    assert np.array_equal(ar[:100].numpy(), G["anchors_head"])
    assert [float(ar.double().sum()), float(sr.double().sum())] == list(G["anchors_sum"])
    d = unit_input("dist", (2, 4, 50), 0, 15)
    ap = unit_input("dist_anchor", (1, 2, 50), 0, 80)
    assert np.array_equal(om.dist2bbox(d, ap, xywh=True, dim=1).numpy(), G["dist2bbox"])
    # known answers from the reference docstrings (nn/modules/utils.py:49-51, 93-95)
    v = om.inverse_sigmoid(torch.tensor([0.2, 0.5, 0.8]))
    assert np.array_equal(v.numpy(), G["inverse_sigmoid"])
    assert np.allclose(v.numpy(), [-1.3863, 0.0, 1.3863], atol=1e-4)
    assert abs(om.bias_init_with_prob(0.01) - (-4.5951)) < 1e-4


def test_unit_detect(G):
    d = bn_fix(om.Detect(80, (16, 32, 64)))
    d.stride = torch.tensor([8.0, 16.0, 32.0])
    P.apply_procedural_weights(d, family="yolov8n")
    xs = [unit_input(f"det{i}", s) for i, s in enumerate([(2, 16, 8, 8), (2, 32, 4, 4), (2, 64, 2, 2)])]
    with torch.no_grad():
        y, raw = d(xs)
    assert y.shape == (2, 84, 84)
    assert np.abs(y.numpy() - G["detect_y"]).max() <= 1e-4
    assert np.abs(raw[0].numpy() - G["detect_raw0"]).max() <= 1e-5


def test_unit_msdeform_and_rtdetr(G):
    o = om.MSDeformAttn(32, 3, 4, 4).eval()
    P.apply_procedural_weights(o)
    with torch.no_grad():
        y = o(unit_input("msda_q", (2, 10, 32)), unit_input("msda_ref", (2, 10, 1, 4), 0.1, 0.9),
              unit_input("msda_v", (2, 84, 32)), [[8, 8], [4, 4], [2, 2]])
    assert np.abs(y.numpy() - G["msdeform_attn"]).max() <= 1e-5
    r = bn_fix(om.RTDETRDecoder(80, (16, 32, 64), 32, 10, 4, 4, 2, 64))
    P.apply_procedural_weights(r)
    xs = [unit_input(f"rtd{i}", s) for i, s in enumerate([(2, 16, 8, 8), (2, 32, 4, 4), (2, 64, 2, 2)])]
    with torch.no_grad():
        y = r(xs)[0]
    assert np.abs(y.numpy() - G["rtdetr_decoder_small"]).max() <= 1e-5


def _nms_case_names(path):
    g = np.load(path)
    return sorted(k[:-5] for k in g.files if k.endswith("_pred"))


def test_nms_docstring_case():
    b = torch.tensor([[0.0, 0, 10, 10], [5, 5, 15, 15]])
    assert onms.greedy_nms(b, torch.tensor([0.9, 0.8]), 0.5).tolist() == [0, 1]  # IoU 25/175: keep both


def test_nms_cases(golden_dir):
    g = np.load(golden_dir / "nms_cases.npz")
    names = _nms_case_names(golden_dir / "nms_cases.npz")
    assert len(names) >= 19
    for name in names:
        kw = json.loads(str(g[name + "_kw"]))
        out, keep = onms.non_max_suppression(torch.from_numpy(g[name + "_pred"]), return_idxs=True, **kw)
        assert [o.shape[0] for o in out] == list(g[name + "_n"]), name
        rows = torch.cat(out, 0).numpy() if sum(o.shape[0] for o in out) else np.zeros((0, 6), "f4")
        assert np.array_equal(rows, g[name + "_out"]), name
        assert np.array_equal(torch.cat([k.view(-1) for k in keep]).numpy(), g[name + "_keep"]), name


@pytest.mark.parametrize("name", ["yolov3-tiny", "yolov8n", "yolov5-BoT3"])
def test_e2e_detect(name, golden_dir):
    g = np.load(golden_dir / f"e2e_{name}.npz")
    m = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m)
    m.fuse()
    with torch.no_grad():
        y = m(P.synthetic_images(2))[0]
    assert np.abs(y[:, :, g["anchor_sel"]].numpy() - g["y_sel"]).max() <= 1e-4
    assert np.allclose(y.double().mean(dim=(0, 2)).numpy(), g["y_chan_mean"], rtol=1e-6, atol=1e-7)
    out = onms.non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    assert np.abs(torch.cat(out, 0).numpy() - g["predict_rows"]).max() <= 1e-3


def test_e2e_detect_smooth_family(golden_dir):
    """The "smooth" weight family (utils/procedural.py SMOOTH_RECIPE) is pinned the same way: the oracle on those weights
    equals the imported reference's recorded output (tests/golden/e2e_yolov8n_smooth.npz)."""
    g = np.load(golden_dir / "e2e_yolov8n_smooth.npz")
    m = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m, family="smooth:yolov8n")
    m.fuse()
    with torch.no_grad():
        y = m(P.synthetic_images(2))[0]
    assert np.abs(y[:, :, g["anchor_sel"]].numpy() - g["y_sel"]).max() <= 1e-4
    out = onms.non_max_suppression(y, conf_thres=0.25, iou_thres=0.7, max_det=300)
    assert [o.shape[0] for o in out] == list(g["predict_n"])
    assert np.abs(torch.cat(out, 0).numpy() - g["predict_rows"]).max() <= 1e-3


def test_e2e_rtdetr(golden_dir):
    g = np.load(golden_dir / "e2e_yolov3-rtdetr.npz")
    m = ot.DetectionModel("yolov3-rtdetr.yaml")
    P.apply_procedural_weights(m)
    m.fuse()
    with torch.no_grad():
        y = m(P.synthetic_images(2))[0]
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-4
    outs = onms.rtdetr_postprocess(y, 0.25)
    assert [o.shape[0] for o in outs] == list(g["post_n"])


def test_oracle_train_step_matches_reference_golden(golden_dir):
    """oracle/train.py (train-mode forward, v8DetectionLoss + TaskAlignedAssigner, backward, clip,
    SGD nesterov, EMA) reproduces what the imported reference recorded in tests/golden/train_yolov8n.npz."""
    from oracle import tasks as ot
    from oracle import train as otr
    G = np.load(golden_dir / "train_yolov8n.npz")
    bs, imgsz = int(G["bs"][0]), int(G["imgsz"][0])
    torch.set_num_threads(4)
    m = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    state = otr.TrainState(m)
    x = P.synthetic_images(bs, h=imgsz, w=imgsz, seed=0)
    items, norm = otr.train_step(m, state, {"img": x, **P.synthetic_labels(bs, seed=0)})
    np.testing.assert_allclose(items.numpy(), G["loss_items_0"], rtol=2e-5)
    assert abs(norm - float(G["grad_norm_0"][0])) <= 2e-4 * norm
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    keys = [str(k) for k in G["param_keys"]]
    assert list(grads) == keys
    l2 = np.array([float(grads[k].double().norm()) for k in keys])
    np.testing.assert_allclose(l2, G["grad_l2_0"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(grads["model.22.cv3.0.2.bias"].numpy(), G["grad_cls_bias_0"], rtol=1e-3, atol=1e-6)
    sd = m.state_dict()
    np.testing.assert_allclose(sd["model.0.conv.weight"].numpy(), G["w_stem_0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd["model.2.cv1.bn.running_var"].numpy(), G["bn_rv_0"], rtol=1e-5)
    fk = [str(k) for k in G["state_keys"]]
    np.testing.assert_allclose(np.array([float(state.ema[k].double().sum()) for k in fk]), G["ema_sum_0"], rtol=1e-4, atol=1e-4)


def test_reference_f32_noise_floor_justifies_the_yolov8s_box_gate():
    """BASELINE.json's gate is 1e-3 on boxes / scores.  The GPU tests widen the BOX gate for yolov8s to 3e-3 px
    (tests/test_hip_e2e.py BOX_TOL): this test is the justification, measured, not argued - the reference's OWN f32 arithmetic
    (oracle == reference bit for bit, pinned by the e2e goldens above) against a float64 run of the same model on the same images
    moves the decoded boxes by more than 1e-3 px on that config (DFL expectation x stride 32 on coordinates up to 640 px), so
    1e-3 px is below what f32 can reproduce there; scores stay far inside 1e-3.  yolov8n - the headline config - stays below
    1e-3 and keeps the stated gate."""
    torch.set_num_threads(min(8, torch.get_num_threads()))
    x = P.synthetic_images(2)
    floors = {}
    for name in ("yolov8n", "yolov8s"):
        m = ot.DetectionModel(name + ".yaml")
        P.apply_procedural_weights(m)
        m.fuse()
        with torch.no_grad():
            y32 = m(x)[0]
            y64 = m.double()(x.double())[0]
        d = (y32.double() - y64).abs()
        floors[name] = (float(d[:, :4].max()), float(d[:, 4:].max()))
    print("f32 vs f64 of the reference arithmetic (box px, score):", floors)
    assert floors["yolov8n"][0] < 1e-3 and floors["yolov8n"][1] < 1e-4      # the headline config keeps the 1e-3 gate
    assert floors["yolov8s"][0] > 1e-3, "yolov8s box noise floor fell below 1e-3: tighten BOX_TOL in tests/test_hip_e2e.py"
    # ... and the widened gate (3e-3 px) is the floor itself (2.2e-3 .. 2.9e-3 px by thread count and host; HIP f32 measures 1.7e-3)
    assert floors["yolov8s"][0] < 5e-3 and floors["yolov8s"][1] < 1e-4


@pytest.mark.parametrize("name", ["yolov8n", "yolov8s", "yolov3-tiny"])
def test_bf16_emulation_meets_the_amp_gate_on_the_smooth_goldens(name, golden_dir):
    """oracle/bf16_emul.py (bf16 roundings at the product's rounding points, CPU) against the REFERENCE's f32 output on the smooth
    weight family (tests/golden/e2e_<cfg>_smooth.npz): the emulation must itself sit inside the gate the HIP bf16 mode is held to
    against those goldens (0.5 px, the reference's AMP self-check utils/checks.py:780; scores 0.005) - otherwise it could not serve
    as the deterministic stand-in for that mode - and its values are bf16-exact where the product stores bf16."""
    from oracle.bf16_emul import bf16_round, emulate_bf16
    g = np.load(golden_dir / f"e2e_{name}_smooth.npz")
    m = ot.DetectionModel(name + ".yaml")
    P.apply_procedural_weights(m, family="smooth:" + name)
    m.eval()
    e = emulate_bf16(m)
    grabbed = {}
    first = next(mod for mod in e.modules() if isinstance(mod, om.Conv))
    first.register_forward_hook(lambda _m, _i, out: grabbed.setdefault("y0", out))
    with torch.no_grad():
        y = e(P.synthetic_images(2))[0]
    assert torch.equal(grabbed["y0"], bf16_round(grabbed["y0"]))  # a Conv output is stored as bf16
    d = np.abs(y[:, :, g["anchor_sel"]].numpy() - g["y_sel"])
    print(f"{name} smooth: emulated bf16 vs reference f32 golden: box max {d[:, :4].max():.4f} px, score max {d[:, 4:].max():.5f}")
    assert d[:, :4].max() <= 0.5 and d[:, 4:].max() <= 0.005
