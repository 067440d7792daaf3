#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04k2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "keys_only or detect" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.log
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for m in yolov8s yolov3-tiny; do for a in "" "--full-scores" "--full-scores" ""; do echo "$m ARGS $a"; python bench.py --model $m --no-cpu-baseline --no-kernel-profile --no-parity $a 2>/dev/null | j; done; done | tee $O/ab.txt
