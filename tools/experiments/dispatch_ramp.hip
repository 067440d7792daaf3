// How fast does the MI355X place the workgroups of ONE launch?  Every workgroup records s_memtime (shader clock) when its first wave starts and when
// its last wave ends; the kernel does `spin` dependent VALU operations in between.  Printed per configuration: launch duration by HIP events and the
// median workgroup life (the counters of different XCDs are not synchronised, so start times are only comparable inside an XCD: the ramp is read
// off as launch - empty launch - life).  Round 6, MI355X: an empty launch is 6.0 us event to event whatever the grid (256 .. 2048 workgroups, 0 ..
// 70 KB of LDS); with a 10 k-cycle body 256 x 8 waves take 8.5 us, 448 x 8 waves 11.7, 448 x 4 waves 8.5, 448 x 16 waves 18.0, 1024 x 8 waves 18.7
// (life 22.5 k: 8 waves per SIMD): the launch grows by ~13-16 shader cycles per WAVE and XCD beyond the body - 3200 waves (a 400-workgroup
// conv_big launch on a 40 x 40 map) pay ~2.5 us of wave launch, LDS size does not matter.
// Build: hipcc -O3 --offload-arch=gfx950 -o dispatch_ramp dispatch_ramp.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int VG>
__global__ __launch_bounds__(1024) void ramp_kernel(unsigned long long* out, int spin, float* sink) {
  extern __shared__ char sm[];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float v[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) v[i] = threadIdx.x + i;
  for (int s = 0; s < spin; ++s)
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = v[i] * 1.0001f + 0.5f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < VG; ++i) acc += v[i];
  if (sm && spin < 0) sm[threadIdx.x] = (char)acc;
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1; }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int VG>
static void run(int grid, int threads, int lds, int spin) {
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, grid * 16);
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)ramp_kernel<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  std::vector<unsigned long long> h(2 * grid);
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(ramp_kernel<VG>, dim3(grid), dim3(threads), lds, 0, d, spin, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
  unsigned long long s0 = ~0ull, s1 = 0, e_last = 0;
  std::vector<long> life;
  for (int i = 0; i < grid; ++i) {
    s0 = std::min(s0, h[2 * i]); s1 = std::max(s1, h[2 * i]); e_last = std::max(e_last, h[2 * i + 1]);
    life.push_back((long)(h[2 * i + 1] - h[2 * i]));
  }
  std::sort(life.begin(), life.end());
  (void)s0; (void)s1; (void)e_last;
  printf("grid %5d x %4d threads, LDS %6d B, %3d chains, spin %5d: launch %.2f us, median workgroup life %ld cycles\n", grid, threads, lds, VG, spin,
         best * 1e3, life[life.size() / 2]);
  hipFree(d); hipFree(sink);
}

int main() {
  for (int spin : {0, 200}) {
    for (int lds : {0, 32 * 1024, 70 * 1024}) {
      run<8>(256, 512, lds, spin);
      run<8>(448, 512, lds, spin);
      run<8>(448, 256, lds, spin);
      run<8>(896, 256, lds, spin);
      run<8>(1024, 512, lds, spin);
    }
    run<64>(448, 512, 70 * 1024, spin);
    run<8>(448, 1024, 70 * 1024, spin);
    run<8>(2048, 64, 0, spin);
    run<8>(32, 1024, 64 * 1024, spin);
  }
  return 0;
}
