#!/bin/bash
# the yolov3-rtdetr rows of tools/round_profiles.sh alone (after the pair / query-selection kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/rtdetr_r04; mkdir -p $O
cd $R
timeout 600 python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline > $O/bench_yolov3-rtdetr.json 2>/dev/null
timeout 600 python bench.py --model yolov3-rtdetr --batch 16 --serial --no-cpu-baseline --no-kernel-profile > $O/bench_yolov3-rtdetr_serial.json 2>/dev/null
timeout 300 python tools/bench_conv.py --model yolov3-rtdetr --batch 16 > $O/conv_layers_yolov3-rtdetr.txt 2>/dev/null
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_yolov3-rtdetr.json $O/bench_yolov3-rtdetr_serial.json
tail -1 $O/conv_layers_yolov3-rtdetr.txt
