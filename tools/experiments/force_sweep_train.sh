#!/bin/bash
b() { echo "$* : $(env "$@" python bench.py --opts env --workload train --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
b A=1
for f in 2,2,4,2 4,1,2,4 4,1,4,2 4,2,2,2 4,1,4,4 2,2,8,2 8,1,2,2 4,2,4,2; do b UPA_CONV_FORCE=$f; done
b UPA_CONV_CKT=2
