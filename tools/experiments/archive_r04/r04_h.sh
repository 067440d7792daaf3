#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04h; mkdir -p $O
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'), d['config'].get('autotune_ms_per_step'))"; }
for i in 1 2 3 4 5 6; do echo "short $i"; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | j; done > $O/short.txt
for i in 1 2; do echo "long $i"; python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | j; done >> $O/short.txt
for i in 1 2 3; do echo "short200 $i"; python bench.py --gpus 1 --steps 20 --warmup 200 --no-cpu-baseline --no-kernel-profile 2>/dev/null | j; done >> $O/short.txt
cat $O/short.txt
