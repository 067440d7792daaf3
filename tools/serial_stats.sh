#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --serial` (per-kernel average durations comparable with roofline.avg_launch_us);
# usage (GPU box): tools/serial_stats.sh <outdir> [extra bench args]  ->  <outdir>/serial_kernel_stats.csv
out=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/bench.py --serial --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 5 "$@" > $root/$out/bench_serial_profiled.json 2>/dev/null
cd $root
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp $f $out/serial_kernel_stats.csv
rm -rf $out/prof
python3 - "$out/serial_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 65.0 + 2  # warmup + steps replays (+ the eager walks of compile())
tot = 0.0
for r in rows[:40]:
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    print(f'{r["Name"][:80]:80s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.2f} us  per-step {per:8.2f} us')
print("sum of listed per-step us:", round(tot, 1))
PY
