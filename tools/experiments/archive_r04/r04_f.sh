#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s -k "mhsa or bot3 or BoT3 or layer_by_layer" 2>&1 | grep -v amdgpu.ids > $O/tests.log
grep -E "mhsa bf16|passed|failed|Error|assert" $O/tests.log | cut -c1-300
python bench.py --model yolov5-BoT3 --no-cpu-baseline --no-kernel-profile --steps 300 > $O/bench_bot3.json 2>/dev/null
python -c "import json;d=json.load(open('$O/bench_bot3.json'));print('BoT3', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -- python3 $R/bench.py --model yolov5-BoT3 --serial --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 5 > /dev/null 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete
grep -h "mhsa" $O/prof_serial/*/*kernel_stats.csv | cut -c1-200
