#!/bin/bash
# in-flight kernel stats of a secondary config: usage r04_z.sh <model> [batch]
cd "$(dirname "$0")/../.." || exit 1
root=$(pwd)
m=${1:-yolov8s}; b=${2:-32}
O=gpurun_out/r04z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/prof_$m -- python3 $root/bench.py --model $m --batch $b --no-cpu-baseline --no-kernel-profile --no-parity --steps 300 > $root/$O/bench_$m.json 2>/dev/null
cd $root
f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); cp $f $O/${m}_kernel_stats.csv; rm -rf $O/prof_$m
python3 - $O/${m}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]: print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
PY
