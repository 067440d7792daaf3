# Validate-path A/B on one box: first-stage NMS prefix from the coarse score histogram (default) vs the radix select
# (UPA_NMS_NO_COARSE=1), then a kernel trace of the default.   gpurun -- 'bash tools/experiments/r05_val_ab.sh'
mkdir -p gpurun_out/val_ab
for r in 1 2 3; do
  UPA_NMS_NO_COARSE=1 python bench.py --workload val 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('radix select    ', d['value'], d['ms_per_step'])"
  python bench.py --workload val 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('coarse histogram', d['value'], d['ms_per_step'])"
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/val_ab/prof -- python3 $R/bench.py --workload val --steps 100 > $R/gpurun_out/val_ab/prof.log 2>&1
cd $R
f=$(ls -t gpurun_out/val_ab/prof/*/*kernel_stats.csv | head -1)
head -16 $f | cut -d, -f1-4 | cut -c1-150
