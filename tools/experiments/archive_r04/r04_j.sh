#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial_rtdetr -- python3 $R/bench.py --model yolov3-rtdetr --batch 16 --serial --no-cpu-baseline --no-kernel-profile --steps 30 --warmup 3 > $R/$O/bench_serial_rtdetr.json 2>/dev/null
cd $R; find $O -name "*kernel_trace.csv" -delete
python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline --no-kernel-profile > $O/bench_rtdetr.json 2>/dev/null
python -c "import json;d=json.load(open('$O/bench_rtdetr.json'));print('rtdetr', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
