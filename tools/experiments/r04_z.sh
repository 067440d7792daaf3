#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
root=$(pwd)
O=gpurun_out/r04z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/prof -- python3 $root/bench.py --model yolov8s --no-cpu-baseline --no-kernel-profile --no-parity --steps 300 > $root/$O/bench.json 2>/dev/null
cd $root
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/v8s_kernel_stats.csv; rm -rf $O/prof
python3 - $O/v8s_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]: print(f"{r['Name'][:120]:120s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
PY
