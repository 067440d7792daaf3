#!/usr/bin/env python3
"""Run in an environment where the REFERENCE (`ultralytics`) is importable: turns one of its checkpoints (a pickle of the
model object, engine/trainer.py:579-618) into a plain float32 state_dict that ultralytics_pro_amd.utils.weights.load_weights
reads.  usage: python tools/export_reference_state_dict.py best.pt best_state.pt"""
import sys
from collections import OrderedDict

import torch


def main(src, dst):
    from ultralytics.nn.tasks import torch_safe_load  # the reference's loader (nn/tasks.py:2291)

    ckpt, _ = torch_safe_load(src)
    model = ckpt.get("ema") or ckpt["model"]
    sd = model.float().state_dict()
    torch.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), dst)
    print(f"{dst}: {len(sd)} tensors, {sum(v.numel() for v in sd.values())} values")


if __name__ == "__main__":
    main(*sys.argv[1:3])
