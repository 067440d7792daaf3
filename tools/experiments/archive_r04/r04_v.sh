#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04v; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "fused_stem or (e2e and yolov8s)" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.log
python tools/experiments/archive_r04/r04_s.py 2>/dev/null | grep -E "yolov8s|yolov8n" | tee $O/stem.txt
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for i in 1 2; do python bench.py --model yolov8s --no-cpu-baseline --no-kernel-profile --no-parity 2>/dev/null | j; done | tee $O/v8s.txt
