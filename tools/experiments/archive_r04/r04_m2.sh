#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04m2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "nms or map or validator or val" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
bash tools/experiments/archive_r04/r04_l2.sh | head -6
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do python bench.py --workload val --no-cpu-baseline --steps 300 2>/dev/null | j; done | tee $O/val.txt
