#!/bin/bash
# in-flight kernel stats of the training step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/y2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 4 > $O/bench.json 2>/dev/null
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/train_kernel_stats.csv; rm -rf $O/prof
python3 - $O/train_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=24
for r in rows[:34]: print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:6.1f} {float(r['AverageNs'])/1e3:8.1f} {float(r['TotalDurationNs'])/steps/1e6:7.3f}")
print("total ms/step", sum(float(r['TotalDurationNs']) for r in rows)/steps/1e6)
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.json
