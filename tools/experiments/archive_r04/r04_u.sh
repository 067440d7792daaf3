#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04u; mkdir -p $O
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for m in yolov3-tiny yolov8s; do
for a in "" "--opts conv_p8=1" "--opts conv_p8=1" ""; do
  echo "$m ARGS $a"; python bench.py --model $m --no-cpu-baseline --no-kernel-profile --no-parity $a 2>/dev/null | j
done; done | tee $O/ab_other.txt
