#!/bin/bash
# rocprofv3 kernel stats of a model's bench run: usage prof_model.sh <outdir> <bench args...>
out=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/bench.py --no-cpu-baseline --no-kernel-profile "$@" > $root/$out/bench.json 2>/dev/null
cd $root
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
rm -rf $out/prof
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:86]:86s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
tail -c 400 $out/bench.json
