#!/bin/bash
# every bench mode once on the current build (values only)
v() { grep -o '"value": [0-9.]*' | head -1; }
echo "yolov8s $(python bench.py --model yolov8s --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | v)"
echo "yolov3-tiny $(python bench.py --model yolov3-tiny --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | v)"
echo "yolov5-BoT3 $(python bench.py --model yolov5-BoT3 --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | v)"
echo "f32 $(python bench.py --dtype f32 --no-cpu-baseline --no-kernel-profile --steps 200 2>/dev/null | v)"
echo "val $(python bench.py --workload val --no-cpu-baseline --no-kernel-profile --steps 300 2>/dev/null | v)"
echo "serial $(python bench.py --serial --no-cpu-baseline --no-kernel-profile 2>/dev/null | v)"
