"""Experiment: the in-flight lanes of PipelinedRunner on CU-masked streams (hipExtStreamCreateWithCUMask): each lane gets its own
slice of the chip instead of competing for all 256 CUs.  usage: python tools/experiments/cu_mask_lanes.py [lanes] [interleave]"""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from ultralytics_pro_amd.engine.pipeline import PipelinedRunner
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
from ultralytics_pro_amd.utils.nms import nms_raw

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
interleave = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")
m = DetectionModel("yolov8n.yaml"); P.apply_procedural_weights(m); m = m.to(dev).eval(); m.set_compute_dtype(torch.bfloat16)
det = m.model[-1]; det.keep_raw = False; det.nms_keys = True
xs = [P.synthetic_images(32, first=32 * j).to(dev).to(torch.bfloat16).contiguous() for j in range(8)]
post = lambda o: nms_raw(o[0], 0.25, 0.7, max_det=300, key="b")
with torch.no_grad():
    r = PipelinedRunner(m, xs, post, micro_batches=1, in_flight=lanes, linear=True)
    def rate(tag):
        for _ in range(50): r.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000): r.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{tag}: {32 * 1000 / dt:9.1f} img/s  {dt:.3f} ms/step", flush=True)
    rate("plain lanes")
    ncu = 256
    words = ncu // 32
    new = []
    for i in range(lanes):
        mask = (C.c_uint32 * words)()
        for cu in range(ncu):
            owner = (cu % lanes) if interleave else (cu * lanes // ncu)
            if owner == i:
                mask[cu // 32] |= 1 << (cu % 32)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
        assert rc == 0, rc
        new.append(torch.cuda.ExternalStream(st.value, device=dev))
    r.lanes = new
    rate(f"{lanes} CU-masked lanes ({'interleaved' if interleave else 'contiguous'} {ncu // lanes} CUs each)")
