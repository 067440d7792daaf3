R=$PWD; mkdir -p gpurun_out/rtd; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rtd -- python3 $R/bench.py --model yolov3-rtdetr --batch 16 --serial --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 5 > $R/gpurun_out/rtd/bench.json 2>/dev/null
cd $R
f=$(ls -t gpurun_out/rtd/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 45
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("all kernels us/step:", round(tot / 1e3 / steps, 1))
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:7.1f} us = {int(r['TotalDurationNs'])/1e3/steps:8.1f} us/step")
PY
