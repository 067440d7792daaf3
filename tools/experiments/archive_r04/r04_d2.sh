#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04d2; mkdir -p $O
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for b in 16 32 64; do echo "rtdetr batch $b"; python bench.py --model yolov3-rtdetr --batch $b --no-cpu-baseline --no-kernel-profile --no-parity 2>/dev/null | j; done | tee $O/batch_sweep.txt
