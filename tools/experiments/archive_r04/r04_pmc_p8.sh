#!/bin/bash
# L2 hit rate and instruction mix of the 512 -> 256 @40x40 layer (yolov3-rtdetr bs 16) on conv_p8 and on conv_big
cd "$(dirname "$0")/../.." || exit 1
EXTRA_ARGS="--model yolov3-rtdetr --batch 16 --opts conv_p8=2" bash tools/pmc_layer.sh "512,256,3,40" gpurun_out/pmc_p8 > /dev/null 2>&1
EXTRA_ARGS="--model yolov3-rtdetr --batch 16 --opts conv_p8=1" bash tools/pmc_layer.sh "512,256,3,40" gpurun_out/pmc_big > /dev/null 2>&1
echo "== conv_p8"; cat gpurun_out/pmc_p8/summary.txt
echo "== conv_big"; cat gpurun_out/pmc_big/summary.txt
