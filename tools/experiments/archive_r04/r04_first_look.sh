#!/bin/bash
# round 4, first GPU call: where this round starts on this box + phase stamps of the chip-filling whole-block kernels
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r04a
unset UPA_HIP_LIB
python bench.py --no-cpu-baseline > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err
for layer in 2 4 15; do
  UPA_HIP_LIB=$PWD/ultralytics_pro_amd/libupa_hip_stamp.so python tools/experiments/c2f_stamps.py --layer $layer > gpurun_out/r04a/stamps_layer$layer.txt 2>&1
done
tail -n 30 gpurun_out/r04a/stamps_layer4.txt
