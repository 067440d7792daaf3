"""What the validate-path NMS (conf 0.001, multi_label, max_det 300) sees per image on the synthetic validation batches: candidates,
length of the sorted first prefix, boxes kept, and whether the greedy pass flagged the image for the longer stages.
    python3 tools/experiments/r05_val_nms_counts.py [--batches 2]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ultralytics_pro_amd.engine import runtime as R  # noqa: E402
from ultralytics_pro_amd.nn.tasks import DetectionModel  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402
from ultralytics_pro_amd.utils.nms import nms_raw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m = m.to(dev).eval()
    m.set_compute_dtype(torch.bfloat16)
    for bi in range(a.batches):
        x = P.synthetic_images(32, first=32 * bi).to(dev).to(torch.bfloat16)
        with torch.no_grad():
            y = m(x)
            y = (y[0] if isinstance(y, (tuple, list)) else y).float().contiguous()
            out, counts, _ = nms_raw(y, 0.001, 0.7, multi_label=True, max_det=300, key="cnt")
            torch.cuda.synchronize()
            ws = R.alloc_plain((1,), torch.uint8, dev, key=("cnt", "nms_ws")) if False else None
        # the counters sit at the front of the workspace (csrc/nms.hip: count, nsorted, partial, redo, redo2, mode, pcount, count2)
        from ultralytics_pro_amd import _lib as L
        nb = L.lib().upa_nms_workspace_bytes(32, 80, y.shape[2], 1, 30000)
        wsb = R.alloc_plain((nb,), torch.uint8, dev, key=("cnt", "nms_ws"))
        c = wsb[: 8 * 32 * 4].view(torch.int32).view(8, 32).cpu()
        print(f"batch {bi}: candidates/image min {int(c[0].min())} mean {int(c[0].float().mean())} max {int(c[0].max())}; "
              f"sorted list (last stage that ran) mean {int(c[1].float().mean())}; first prefix (emit) mean {int(c[6].float().mean())}; "
              f"kept min {int(counts.min())} mean {int(counts.float().mean())}; flagged after stage 1: {int(c[3].sum())}, after stage 2: {int(c[4].sum())}")
        sc = out[..., 4]
        k = counts.cpu()
        low = torch.stack([sc[i, k[i] - 1] for i in range(32)])
        used = (y[:, 4:] >= low.view(32, 1, 1)).sum((1, 2))
        print("   candidates the greedy pass walked before it had max_det boxes: min", int(used.min()), "mean", int(used.float().mean()), "max", int(used.max()))
        print("   classes among the kept boxes (first 8 images):", [int(out[i, :k[i], 5].unique().numel()) for i in range(8)])
        print("   lowest kept score per image (first 8):", [round(float(sc[i, k[i] - 1]), 4) if k[i] else None for i in range(8)])


if __name__ == "__main__":
    main()
