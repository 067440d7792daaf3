# Validate-path A/Bs on one box (upa_opts.nms_stages / nms_first_prefix), then a kernel trace of the default.
#   gpurun -- 'bash tools/experiments/r05_val_ab.sh'
mkdir -p gpurun_out/val_ab
run() { l=$1; shift; python bench.py --workload val "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  run "radix select in the first stage                     " --opts nms_stages=2
  run "coarse histogram, all keys, first prefix 16384      " --opts nms_stages=1,nms_first_prefix=-1
  run "coarse histogram, all keys, first prefix 4096       " --opts nms_stages=1
  run "coarse histogram, prefix keys only, prefix 8192     " --opts nms_first_prefix=8192
  run "coarse histogram, prefix keys only, prefix 4096     "
  run "coarse histogram, prefix keys only, prefix 2048     " --opts nms_first_prefix=2048
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/val_ab/prof -- python3 $R/bench.py --workload val --steps 100 > $R/gpurun_out/val_ab/prof.log 2>&1
cd $R
f=$(ls -t gpurun_out/val_ab/prof/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nb = max(int(r['Calls']) for r in rows if 'candidates' in r['Name'])
for r in [r for r in rows if 'nms' in r['Name'] or 'zero' in r['Name']] + rows[:6]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} calls {int(r['TotalDurationNs'])/1e3/nb:8.1f} us/batch  avg {float(r['AverageNs'])/1e3:7.1f} us")
print("all kernels:", round(sum(int(r['TotalDurationNs']) for r in rows) / 1e3 / nb, 1), "us/batch")
PY
