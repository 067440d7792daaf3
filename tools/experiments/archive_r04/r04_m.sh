#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04m; mkdir -p $O
for m in yolov8s yolov3-tiny yolov5-BoT3; do
  timeout 600 python bench.py --model $m --no-cpu-baseline --steps 300 > $O/bench_$m.json 2> $O/bench_$m.err
  python -c "
import json;d=json.load(open('$O/bench_$m.json'));r=d['roofline'];print('$m', d['value'], r and (r['kernel'][:60], r['frac'], r['bound'], r['avg_launch_us']))"
done
timeout 600 python bench.py --model yolov3-rtdetr --batch 16 --no-cpu-baseline > $O/bench_yolov3-rtdetr.json 2> $O/bench_yolov3-rtdetr.err
python -c "
import json;d=json.load(open('$O/bench_yolov3-rtdetr.json'));r=d['roofline'];print('rtdetr', d['value'], r and (r['kernel'][:60], r['frac'], r['bound'], r['avg_launch_us']))"
tail -3 $O/*.err
