#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "fused_stem or BoT3 or bot3 or stem" 2>&1 | grep -v amdgpu.ids | tail -12 > $O/tests.log
cat $O/tests.log
for i in 1 2; do python bench.py --model yolov5-BoT3 --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BoT3', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"; done
python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v8n', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
