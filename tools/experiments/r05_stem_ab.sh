# The same-box A/B that measured the fused stem kernel's stage-2 software pipeline (a compile-time switch UPA_STEM_PIPE that csrc/stem.hip no
# longer has: the pipeline was slower and is gone - profiles/r05_stem_phases.txt); kept as the recipe: tests, phase profile of both builds,
# kernel time under rocprofv3 --stats, serial step.
#   gpurun -- 'bash tools/experiments/r05_stem_ab.sh'
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "stem" 2>&1 | tail -1
bash tools/experiments/r05_stem_phases.sh
EXTRA=-DUPA_STEM_PIPE=0 bash tools/experiments/r05_stem_phases.sh
cd ultralytics_pro_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off -DUPA_STEM_PIPE=0 -c stem.hip -o /tmp/stem_nopipe.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libupa_hip_nopipe.so $(ls *.o | grep -v "abl\|stamp\|^stem.o") /tmp/stem_nopipe.o
cd ../..
run() { l=$1; shift; python bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
for r in 1 2 3; do
  UPA_HIP_LIB=/tmp/libupa_hip_nopipe.so run "serial one segment at a time" --serial --no-cpu-baseline --no-kernel-profile
  run "serial pipelined            " --serial --no-cpu-baseline --no-kernel-profile
  UPA_HIP_LIB=/tmp/libupa_hip_nopipe.so run "infer  one segment at a time" --no-cpu-baseline --no-kernel-profile
  run "infer  pipelined            " --no-cpu-baseline --no-kernel-profile
done
bash tools/experiments/r05_stem_time.sh
