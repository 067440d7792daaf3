#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04k; mkdir -p $O
S=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so
for layer in 4 15; do
  for o in "no_xcd=1" "no_xcd=0" "no_xcd=1" "no_xcd=0"; do
    echo "== layer $layer $o" >> $O/stamps.txt
    UPA_HIP_LIB=$S python tools/experiments/c2f_stamps.py --layer $layer --opts $o 2>&1 | grep -v amdgpu.ids >> $O/stamps.txt
  done
done
grep -E "==|graph replay|x halo|x chunk|cv1|workgroup life" $O/stamps.txt
