#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04t; mkdir -p $O
for o in "conv_p8=2" "conv_p8=3" "conv_p8=4"; do
  echo "== rtdetr $o"
  timeout 600 python tools/bench_conv.py --model yolov3-rtdetr --batch 16 --opts $o 2>/dev/null | grep -E "^ *[0-9]+ +[0-9]+ 3 1 " | awk '$1>=128' | head -8
done > $O/p8_dma_ablation.txt
cat $O/p8_dma_ablation.txt
