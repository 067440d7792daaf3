// What does it cost a launch when every workgroup ends with 2C same-address-class atomics instead of 2C plain stores to its own row?
// (Would int64 fixed-point accumulators - order-independent, so deterministic - in S slots per channel replace the per-workgroup statistics
// rows + combine kernel of the training step?)  G workgroups of 256 threads each spin ~W us, then thread t < 2C either stores a float to
// row[blockIdx][t] (MODE 0), does a non-returning 64-bit atomic add to slot[(blockIdx % S)][t] (MODE 1), or nothing (MODE 2).  Prints us per launch
// over 200 launches (hipEvents) for C = 64, 256, 512 and S = 1, 4, 16, 64.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/atomic_slots.hip -o /tmp/atomic_slots
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* rows, unsigned long long* slots, int C2, int S, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) {}
  __syncthreads();
  for (int t = threadIdx.x; t < C2; t += 256) {
    if (MODE == 0) rows[(size_t)blockIdx.x * C2 + t] = (float)t;
    if (MODE == 1) atomicAdd(&slots[(size_t)(blockIdx.x % S) * C2 + t], (unsigned long long)(t + 1));
  }
}

int main() {
  float* rows; unsigned long long* slots;
  (void)hipMalloc(&rows, (size_t)4096 * 1024 * 4);
  (void)hipMalloc(&slots, (size_t)64 * 1024 * 8);
  (void)hipMemset(slots, 0, (size_t)64 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {400, 800, 1600};
  const int spin = 500;  // s_memtime ticks (shader cycles here: ~0.2 us - the workgroups arrive together, the worst case for contention)
  printf("%5s %5s | %8s %8s | %8s %8s %8s %8s\n", "G", "C", "nothing", "rows", "S=1", "S=4", "S=16", "S=64");
  for (int G : grids)
    for (int C : {64, 256, 512}) {
      float us[6];
      int col = 0;
      auto time = [&](auto kern, int S) {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, 0, rows, slots, 2 * C, S, spin);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, 0, rows, slots, 2 * C, S, spin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        us[col++] = ms * 1e3f / 200;
      };
      time(k<2>, 1); time(k<0>, 1);
      for (int S : {1, 4, 16, 64}) time(k<1>, S);
      printf("%5d %5d | %8.2f %8.2f | %8.2f %8.2f %8.2f %8.2f\n", G, C, us[0], us[1], us[2], us[3], us[4], us[5]);
    }
  return 0;
}
