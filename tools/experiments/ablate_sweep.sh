#!/bin/bash
# needs the ablation build: make -C ultralytics_pro_amd/csrc ablate (the product library compiles the switches out)
# which resource bounds the saturated (4 steps in flight) run?  ablation bits: 1 no input loads, 2 no weight loads, 4 no stores,
# 8 no MFMA, 32 return at once (launch + workgroup dispatch only)
b() { echo "$* : $(env UPA_HIP_LIB=ultralytics_pro_amd/libupa_hip_ablate.so "$@" python bench.py --opts env --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)  serial $(env UPA_HIP_LIB=ultralytics_pro_amd/libupa_hip_ablate.so "$@" python bench.py --opts env --serial --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for a in ${@:-0 15 32}; do
b UPA_CONV_ABLATE=$a UPA_PIPE_ABLATE=$a UPA_C1_ABLATE=$a
done
