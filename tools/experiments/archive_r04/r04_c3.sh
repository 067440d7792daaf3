#!/bin/bash
# validate path vs steps in flight
for n in 4 6 8; do echo "in-flight $n: $(python bench.py --workload val --no-cpu-baseline --no-kernel-profile --steps 300 --in-flight $n 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; done
for n in 4 6 8; do echo "rtdetr in-flight $n: $(python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline --no-kernel-profile --no-parity --in-flight $n 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; done
for n in 4 6; do echo "yolov8s in-flight $n: $(python bench.py --model yolov8s --no-cpu-baseline --no-kernel-profile --no-parity --steps 300 --in-flight $n 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; done
