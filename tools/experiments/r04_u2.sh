#!/bin/bash
# training step with / without the ring weight-gradient kernels, pointwise workgroup budget 128 / 256
b() { echo "$* : $(env "$@" python bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b UPA_WGRAD_RING=0
b UPA_WGRAD_RING=1
b UPA_WGRAD_RING=1 UPA_WGRAD_K1_WGS=256
b UPA_WGRAD_RING=1 UPA_WGRAD_K1_WGS=64
