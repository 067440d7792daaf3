#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
root=$(pwd); O=gpurun_out/r04l2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/prof -- python3 $root/bench.py --workload val --no-cpu-baseline --steps 100 > $root/$O/bench_val.json 2>/dev/null
cd $root
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/val_kernel_stats.csv; rm -rf $O/prof
python3 - $O/val_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
PY
