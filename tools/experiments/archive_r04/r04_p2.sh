#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
# weight-gradient kernel variants (WG_EXP builds of train.hip: 0 product, 1 conflict-free K order, 2 = 1 + no refetch, 3 = 1 + no MFMA)
for v in ${VARIANTS:-0 1 2 3}; do
  echo "== variant $v"
  UPA_HIP_LIB=ultralytics_pro_amd/libupa_exp$v.so python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "tools")
import bench_wgrad as B
for cfg in [(64,64,3,1,80,32),(256,256,3,1,20,32),(128,128,3,1,80,32),(32,32,3,1,160,32),(128,256,3,2,80,32),(32,64,3,2,320,32),
            (768,512,1,1,20,32),(384,256,1,1,40,32),(96,64,1,1,160,32),(128,128,1,1,80,32)]:
    B.run(*cfg)
PY
done
