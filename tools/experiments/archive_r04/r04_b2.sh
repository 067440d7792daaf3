#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04b2; mkdir -p $O
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for a in "" "--opts conv_p8=1" "--opts conv_p8=1" ""; do
  echo "train ARGS $a"; python bench.py --workload train --no-cpu-baseline --no-kernel-profile $a 2>/dev/null | j
done | tee $O/ab_train.txt
