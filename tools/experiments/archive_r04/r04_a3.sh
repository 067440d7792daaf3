#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/a3; mkdir -p $O
python3 $R/tools/experiments/archive_r04/r04_a3.py 2>&1 | grep "wall"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/experiments/archive_r04/r04_a3.py 2>&1 | grep "wall"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/main_kernel_stats.csv; rm -rf $O/prof
python3 - $O/main_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=24
for r in rows[:30]: print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:6.1f} {float(r['AverageNs'])/1e3:8.1f} {float(r['TotalDurationNs'])/steps/1e6:7.3f}")
print("kernel sum ms/step", sum(float(r['TotalDurationNs']) for r in rows)/steps/1e6, "launches/step", sum(int(r['Calls']) for r in rows)/steps)
PY
