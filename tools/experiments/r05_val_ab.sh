# Validate-path A/Bs on one box (environment switches of csrc/nms.hip), then a kernel trace of the default.
#   gpurun -- 'bash tools/experiments/r05_val_ab.sh'
mkdir -p gpurun_out/val_ab
run() { python bench.py --workload val 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  UPA_NMS_NO_COARSE=1 run "radix select in the first stage                     "
  UPA_NMS_NO_EMIT=1 UPA_NMS_FIRST_PREFIX=0 run "coarse histogram, all keys, first prefix 16384      "
  UPA_NMS_NO_EMIT=1 run "coarse histogram, all keys, first prefix 4096       "
  UPA_NMS_FIRST_PREFIX=8192 run "coarse histogram, prefix keys only, prefix 8192     "
  run "coarse histogram, prefix keys only, prefix 4096     "
  UPA_NMS_FIRST_PREFIX=2048 run "coarse histogram, prefix keys only, prefix 2048     "
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
