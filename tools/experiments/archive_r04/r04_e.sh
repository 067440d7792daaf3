#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s -k "headline_batch or config_batch or rounding_point or layer_by_layer" 2>&1 | grep -v amdgpu.ids > $O/tests.log
grep -E "layer |vs emulation|vs oracle|passed|failed|Error|assert" $O/tests.log | cut -c1-600
# kernel mix of the secondary configs (in flight and serial)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in yolov5-BoT3 yolov3-tiny yolov8s; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$m -- python3 $R/bench.py --model $m --no-cpu-baseline --no-kernel-profile --steps 200 --warmup 10 > $R/$O/bench_$m.json 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial_$m -- python3 $R/bench.py --model $m --serial --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 5 > $R/$O/bench_serial_$m.json 2>/dev/null
done
cd $R
find $O -name "*kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
