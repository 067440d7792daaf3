"""CPU (-m "not gpu"): host-side target packing of the training step (utils/loss.py:445-461, 489): run-time gt capacity,
zero-sum boxes masked like the reference's `mask_gt`, xywh(normalised) -> xyxy(pixels)."""

import torch


def test_pack_targets_matches_reference_preprocess():
    from ultralytics_pro_amd.engine.trainer import pack_targets
    labels = {"batch_idx": torch.tensor([0., 0., 2., 2., 2., 1.]), "cls": torch.tensor([5., 7., 1., 2., 3., 9.]),
              "bboxes": torch.tensor([[.5, .5, .2, .2], [0, 0, 0, 0], [.3, .4, .1, .2], [.6, .6, .3, .3], [.2, .8, .1, .1],
                                      [0, 0, 0, 0]])}
    gt, ngt = pack_targets(labels, 3, 320, 640)
    assert tuple(gt.shape) == (3, 64, 5)
    assert ngt.tolist() == [1, 0, 3]  # the two all-zero boxes are dropped (mask_gt)
    assert torch.allclose(gt[0, 0], torch.tensor([5., 256., 128., 384., 192.]))
    assert torch.allclose(gt[2, 1], torch.tensor([2., 288., 144., 480., 240.]))
    assert float(gt[1].abs().sum()) == 0.0


def test_pack_targets_grows_past_64_rows():
    from ultralytics_pro_amd.engine.trainer import pack_targets
    n = 130
    labels = {"batch_idx": torch.zeros(n), "cls": torch.arange(n).float() % 80,
              "bboxes": torch.rand(n, 4).clamp(0.05, 0.9)}
    gt, ngt = pack_targets(labels, 2, 640, 640)
    assert tuple(gt.shape) == (2, 192, 5) and ngt.tolist() == [n, 0]
