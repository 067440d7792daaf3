# Same-box A/B of compile-time variants of csrc/nms.hip (each built into /tmp/libupa_hip_<name>.so on the box, loaded through UPA_HIP_LIB):
#   bins4096 / bins1024: coarse histogram bins (UPA_COARSE_SHIFT 16 / 18; default 17 = 2048 bins)     -> validate path
# (the greedy kernel with 16 waves per image was measured with this script too - GREEDY_NT 1024, equal - and is no longer a build switch)
#   gpurun -- 'bash tools/experiments/r05_nms_variants_ab.sh'
cd ultralytics_pro_amd/csrc
build() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off $2 -c nms.hip -o /tmp/nms_$1.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libupa_hip_$1.so $(ls *.o | grep -v "abl\|stamp\|^nms.o") /tmp/nms_$1.o
}
build bins4096 -DUPA_COARSE_SHIFT=16
build bins1024 -DUPA_COARSE_SHIFT=18
cd ../..
run() { l=$1; shift; python bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for r in 1 2 3; do
  run "val  2048 bins (default)" --workload val
  UPA_HIP_LIB=/tmp/libupa_hip_bins4096.so run "val  4096 bins          " --workload val
  UPA_HIP_LIB=/tmp/libupa_hip_bins1024.so run "val  1024 bins          " --workload val
done
for v in bins4096 bins1024; do UPA_HIP_LIB=/tmp/libupa_hip_$v.so python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "nms" 2>&1 | tail -1; done
