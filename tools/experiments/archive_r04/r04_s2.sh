#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
run() {
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "tools")
import bench_wgrad as B
for cfg in [(64,64,3,1,80,32),(256,256,3,1,20,32),(128,128,3,1,80,32),(32,32,3,1,160,32),(256,128,3,1,40,32),(128,128,3,1,20,32),(64,64,3,1,20,32),
            (128,256,3,2,80,32),(32,64,3,2,320,32),(256,512,3,2,40,32),(64,128,3,2,160,32),(128,128,3,2,80,32)]:
    B.run(*cfg)
PY
}
echo "== register-staged"; UPA_WGRAD_RING=0 run
echo "== ring"; run
