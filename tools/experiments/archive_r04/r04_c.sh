#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s -k "headline_batch or config_batch or large_input or eval_after_training or throughput_dispatch or trainer_bucketed or linear_f32 or conv_ws3 or c2f32 or c2f_fused" > $O/new_tests.log 2>&1
echo "pytest rc=$?" >> $O/new_tests.log
S=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so
for layer in 4 15; do
  for o in "no_prefetch=1" "no_prefetch=0" "no_prefetch=1" "no_prefetch=0"; do
    echo "== layer $layer $o" >> $O/stamps.txt
    UPA_HIP_LIB=$S python tools/experiments/c2f_stamps.py --layer $layer --opts $o 2>&1 | grep -v amdgpu.ids >> $O/stamps.txt
  done
done
bash tools/experiments/ab_opts.sh "--opts no_prefetch=1" "--opts no_prefetch=0" > $O/ab_prefetch.txt 2>&1
grep -E "passed|failed|rc=" $O/new_tests.log | tail -5
grep -E "==|graph replay|x halo|x chunk|workgroup life" $O/stamps.txt
cat $O/ab_prefetch.txt
