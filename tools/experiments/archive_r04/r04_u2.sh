#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
# training step with / without the ring weight-gradient kernels, pointwise workgroup budget 128 / 256
b() { echo "$* : $(env "$@" python bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b UPA_WGRAD_RING=0
b UPA_WGRAD_RING=1
b UPA_WGRAD_RING=1 UPA_WGRAD_K1_WGS=256
b UPA_WGRAD_RING=1 UPA_WGRAD_K1_WGS=64
