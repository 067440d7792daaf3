#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04c2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -k "msdeform or rtdetr" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests.log
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for i in 1 2; do python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline --no-kernel-profile --no-parity 2>/dev/null | j; done | tee $O/bench.txt
bash tools/experiments/archive_r04/r04_z.sh yolov3-rtdetr 16 | grep -E "msdeform|topk|linear|layer_norm|rows_kernel" | tee $O/kernels.txt
