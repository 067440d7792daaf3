#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d3; mkdir -p $O; rm -rf $O/prof
python3 $R/tools/experiments/archive_r04/r04_d3.py 2>&1 | grep -v amdgpu.ids | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/tools/experiments/archive_r04/r04_d3.py > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'nms_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-12:]:
    print(r['Kernel_Name'][:50], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, "us")
PY
rm -rf $O/prof
