# Builds csrc/nms.hip with -DUPA_GREEDY_PROF into /tmp/libupa_hip_prof.so (on the GPU box) and prints the greedy kernel's cycles per phase.
#   gpurun -- 'bash tools/experiments/r05_greedy_phases.sh'
cd ultralytics_pro_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off -DUPA_GREEDY_PROF ${EXTRA} -c nms.hip -o /tmp/nms_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libupa_hip_prof.so $(ls *.o | grep -v "abl\|stamp\|^nms.o") /tmp/nms_prof.o
cd ../..
UPA_HIP_LIB=/tmp/libupa_hip_prof.so python3 tools/experiments/r05_greedy_phases.py 2>&1 | grep -v amdgpu.ids
