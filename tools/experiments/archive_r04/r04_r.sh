#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "conv_mm" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests.log
cat $O/tests.log
for o in "conv_mm=1" "conv_mm=0"; do
  echo "== rtdetr $o"
  python tools/bench_conv.py --model yolov3-rtdetr --batch 16 --opts $o 2>/dev/null | grep -E " 3 1 .*(82|81|41) " | head -9
done > $O/conv_layers.txt
cat $O/conv_layers.txt
