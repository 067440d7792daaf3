#!/bin/bash
# training step vs the workgroup budgets of the weight-gradient kernels (side stream): fewer workgroups leave CUs to the main chain
b() { echo "$* : $(env "$@" python bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b UPA_WGRAD_K3_WGS=256 UPA_WGRAD_K1_WGS=128
b UPA_WGRAD_K3_WGS=128 UPA_WGRAD_K1_WGS=128
b UPA_WGRAD_K3_WGS=64 UPA_WGRAD_K1_WGS=64
b UPA_WGRAD_K3_WGS=128 UPA_WGRAD_K1_WGS=64
b UPA_WGRAD_K3_WGS=192 UPA_WGRAD_K1_WGS=96
b UPA_WGRAD_K3_WGS=96 UPA_WGRAD_K1_WGS=96
