#!/bin/bash
# the training rows of tools/round_profiles.sh alone: bench line + kernel stats + the per-layer weight-gradient table
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/train_r04; mkdir -p $O; rm -rf $O/prof_train
cd $R
timeout 600 python bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err
python tools/experiments/wgrad_pmc_table.py 2>/dev/null > $O/wgrad_layers_yolov8s.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
cd $R
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp $f $O/train_kernel_stats.csv; rm -rf $O/prof_train
cat $O/bench_train.json | head -c 3000
