#!/bin/bash
b() { echo "$* : $(env "$@" python bench.py --opts env --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)  serial $(env "$@" python bench.py --opts env --serial --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b A=1
b UPA_CONV_CKT=1
b UPA_CONV_CKT=2
b UPA_CONV_FORCE=2,2,2,2
b UPA_CONV_FORCE=4,1,2,4
b UPA_CONV_FORCE=4,2,2,2
b UPA_PIPE_MIN_TILES=256
b UPA_C1_MT=1
b UPA_C1_MT=2
