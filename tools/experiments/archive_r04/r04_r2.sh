#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
# ring vs register-staged weight-gradient kernels, per call (kernel + reduce), a few layer shapes; K1 = workgroup budget of the pointwise form
run() {
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "tools")
import bench_wgrad as B
for cfg in [(768,512,1,1,20,32),(384,256,1,1,40,32),(96,64,1,1,160,32),(64,64,1,1,160,32),(128,128,1,1,80,32),(256,128,1,1,80,32),(128,80,1,1,80,32),(64,64,1,1,20,32),
            (64,64,3,1,80,32),(256,256,3,1,20,32),(128,128,3,1,80,32),(32,32,3,1,160,32),(128,256,3,2,80,32),(32,64,3,2,320,32)]:
    B.run(*cfg)
PY
}
echo "== register-staged"; UPA_WGRAD_RING=0 run
echo "== ring, 128 workgroups"; run
echo "== ring, 256 workgroups"; UPA_WGRAD_K1_WGS=256 run
