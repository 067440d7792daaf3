# one-step-at-a-time kernel table of the headline config (rocprofv3 --kernel-trace --stats of bench.py --serial): the kernels >= 15 us.  Usage: r06_serial_kernel_stats.sh [tag] [bench opts...]
R=$(pwd); TAG=${1:-serial}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -- python3 $R/bench.py --serial --no-cpu-baseline --no-kernel-profile --no-parity "$@" > $R/gpurun_out/ks_$TAG.log 2>&1
cd $R
grep '^{' gpurun_out/ks_$TAG.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'])"
f=$(find gpurun_out/ks_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    if float(r['AverageNs'])>=15e3: print(f"{r['Name'][:86]:86s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f}")
PY
cp $f gpurun_out/ks_$TAG.csv; rm -rf gpurun_out/ks_$TAG
