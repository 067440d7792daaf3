#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "conv_maxpool or conv_pool_fusion or test_conv or yolov3-tiny or c16" 2>&1 | grep -v amdgpu.ids | tail -25 > $O/tests.log
cat $O/tests.log
for i in 1 2; do python bench.py --model yolov3-tiny --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"; done
