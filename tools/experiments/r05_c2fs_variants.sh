#!/bin/bash
# Timing variants of csrc/c2f_stream.hip (-DC2FS_EXP = 11 / 12: SiLU replaced by v/2 / by one transcendental, 13: no MFMA, 14: no tap reads, 15 = 11 + 13 - NOT correct kernels): builds
# libupa_hip_exp<k>.so next to the product library (build container), then on the GPU box: tools/experiments/r05_c2fs_variants.sh run
cd "$(dirname "$0")/../.."
if [ "$1" = run ]; then
  for k in 0 11 12 13 14 15; do
    lib=$PWD/ultralytics_pro_amd/libupa_hip_exp$k.so
    [ $k = 0 ] && lib=$PWD/ultralytics_pro_amd/libupa_hip.so
    echo "variant $k"; UPA_HIP_LIB=$lib python tools/experiments/c2f_stamps.py --layer 4 --no-stamps | tail -1
  done
  exit 0
fi
cd ultralytics_pro_amd/csrc
for k in 11 12 13 14 15; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -DC2FS_EXP=$k -c c2f_stream.hip -o /tmp/c2fs_exp$k.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libupa_hip_exp$k.so $(ls *.o | grep -v "abl\|stamp\|c2f_stream.o") /tmp/c2fs_exp$k.o
done
