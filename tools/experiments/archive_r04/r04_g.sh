#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s -k "conv_maxpool or conv_pool_fusion or yolov3-tiny or maxpool" 2>&1 | grep -v amdgpu.ids > $O/tests.log
grep -E "passed|failed|Error|assert|FAILED" $O/tests.log | cut -c1-300
python bench.py --model yolov3-tiny --no-cpu-baseline --no-kernel-profile --steps 300 > $O/bench_tiny.json 2>/dev/null
python -c "import json;d=json.load(open('$O/bench_tiny.json'));print('tiny', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -- python3 $R/bench.py --model yolov3-tiny --serial --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 5 > /dev/null 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete
head -12 $O/prof_serial/*/*kernel_stats.csv | cut -c1-160
