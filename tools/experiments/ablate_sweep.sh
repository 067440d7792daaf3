#!/bin/bash
# which resource bounds the saturated (4 steps in flight) run?  ablation bits: 1 no input loads, 2 no weight loads, 4 no stores,
# 8 no MFMA, 32 return at once (launch + workgroup dispatch only)
b() { echo "$* : $(env "$@" python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)  serial $(env "$@" python bench.py --serial --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for a in ${@:-0 15 32}; do
b UPA_CONV_ABLATE=$a UPA_PIPE_ABLATE=$a UPA_C1_ABLATE=$a
done
