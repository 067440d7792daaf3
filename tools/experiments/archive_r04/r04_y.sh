#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04y; mkdir -p $O
for m in yolov3-tiny yolov8s; do
for o in "conv_p8=2" "conv_p8=1"; do
  echo "== $m $o"
  timeout 600 python tools/bench_conv.py --model $m --batch 32 --opts $o 2>/dev/null | grep -E "^ *[0-9]+ +[0-9]+ 3 1 |TOTAL" | awk '$1>=128' | head -14
done; done > $O/conv_layers.txt
cat $O/conv_layers.txt
