# Builds csrc/stem.hip with -DUPA_STEM_PROF into /tmp/libupa_hip_stemprof.so (on the GPU box) and prints the fused stem kernel's cycles per phase.
#   gpurun -- 'bash tools/experiments/r05_stem_phases.sh'
cd ultralytics_pro_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off -DUPA_STEM_PROF ${EXTRA} -c stem.hip -o /tmp/stem_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libupa_hip_stemprof.so $(ls *.o | grep -v "abl\|stamp\|^stem.o") /tmp/stem_prof.o
cd ../..
UPA_HIP_LIB=/tmp/libupa_hip_stemprof.so python3 tools/experiments/r05_stem_phases.py 2>&1 | grep -v amdgpu.ids
