#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "stacked or blocks_match or mhsa or BoT3 or bot3 or c3_virtual or rtdetr" 2>&1 | grep -v amdgpu.ids | tail -6 > $O/tests.log
cat $O/tests.log
for i in 1 2; do python bench.py --model yolov5-BoT3 --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BoT3', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"; done
