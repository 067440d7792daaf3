#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04e2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "fused_stem or rtdetr" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.log
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for i in 1 2; do python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline --no-kernel-profile --no-parity 2>/dev/null | j; done | tee $O/bench.txt
