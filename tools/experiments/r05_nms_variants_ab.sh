# Same-box A/B of compile-time variants of csrc/nms.hip (each built into /tmp/libupa_hip_<name>.so on the box, loaded through UPA_HIP_LIB):
#   chain:  UPA_GREEDY_ROWS_MIN=65 - phase 2 of the greedy kernel always as the serial walk (the form before the suppression columns)
#   w8:     8 waves per image (default 16);  "rows8" in the output = the default build (columns from 8 alive candidates per chunk)
# (measured with this script earlier and no longer build switches: histogram bins 4096 / 1024 vs 2048 (UPA_COARSE_SHIFT still is one), the sort
#  kernel's scan forms, columns for 1 / 4 candidates per trip, phase-1 unroll)
#   gpurun -- 'bash tools/experiments/r05_nms_variants_ab.sh'
cd ultralytics_pro_amd/csrc
build() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off $2 -c nms.hip -o /tmp/nms_$1.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libupa_hip_$1.so $(ls *.o | grep -v "abl\|stamp\|^nms.o") /tmp/nms_$1.o
}
build chain -DUPA_GREEDY_ROWS_MIN=65
build w8 -DUPA_GREEDY_NT=512
cd ../..
python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py -x -q -m gpu -k "nms" 2>&1 | tail -1
for v in chain w8; do UPA_HIP_LIB=/tmp/libupa_hip_$v.so python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "nms" 2>&1 | tail -1; done
run() { l=$1; shift; python bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for r in 1 2 3; do
  for v in chain rows8 w8; do
    lib=/tmp/libupa_hip_$v.so; [ $v = rows8 ] && lib=$PWD/ultralytics_pro_amd/libupa_hip.so
    UPA_HIP_LIB=$lib run "serial $v" --serial --no-cpu-baseline --no-kernel-profile
  done
  for v in chain rows8; do
    lib=/tmp/libupa_hip_$v.so; [ $v = rows8 ] && lib=$PWD/ultralytics_pro_amd/libupa_hip.so
    UPA_HIP_LIB=$lib run "infer  $v" --no-cpu-baseline --no-kernel-profile
    UPA_HIP_LIB=$lib run "val    $v" --workload val
  done
done
EXTRA= bash tools/experiments/r05_greedy_phases.sh 2>&1 | grep -A7 "batch 0"
