#!/bin/bash
# NOTE: the UPA_WGRAD_* / WG_EXP switches this script sets existed only in the A/B builds of round 4 (git history: "3x3 weight gradient: 12-wave
# LDS-DMA ring kernel" .. "Narrow-input (stem) weight gradient"); the library reads no environment, so they were removed afterwards.
# training step vs the workgroup budgets of the weight-gradient kernels (side stream): fewer workgroups leave CUs to the main chain
b() { echo "$* : $(env "$@" python bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b UPA_WGRAD_K3_WGS=256 UPA_WGRAD_K1_WGS=128
b UPA_WGRAD_K3_WGS=128 UPA_WGRAD_K1_WGS=128
b UPA_WGRAD_K3_WGS=64 UPA_WGRAD_K1_WGS=64
b UPA_WGRAD_K3_WGS=128 UPA_WGRAD_K1_WGS=64
b UPA_WGRAD_K3_WGS=192 UPA_WGRAD_K1_WGS=96
b UPA_WGRAD_K3_WGS=96 UPA_WGRAD_K1_WGS=96
