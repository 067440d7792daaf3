#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "conv_p8" 2>&1 | grep -v amdgpu.ids | tail -15 > $O/tests.log
cat $O/tests.log
for o in "conv_p8=2" "conv_p8=1"; do
  echo "== rtdetr $o"
  timeout 600 python tools/bench_conv.py --model yolov3-rtdetr --batch 16 --opts $o 2>/dev/null | grep -E "^ *[0-9]+ +[0-9]+ 3 1 |TOTAL" | head -16
done > $O/conv_layers.txt
cat $O/conv_layers.txt
