#!/bin/bash
# training step vs the workgroup budgets of the ring weight-gradient kernels (compile-time variants -DUPA_WGRAD_K3_BUDGET / _K1_BUDGET
# built as libupa_exp_<k3>_<k1>.so beside the product library: 256 / 128), interleaved on one box
b() { echo "$1: $(env UPA_HIP_LIB=$2 python bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for rep in 1 2; do
  b "256/128 (product)" ultralytics_pro_amd/libupa_hip.so
  for v in 160_128 128_128 96_128 128_192; do b "$v" ultralytics_pro_amd/libupa_exp_$v.so; done
done
