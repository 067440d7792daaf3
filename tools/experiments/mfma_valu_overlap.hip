// Do matrix and vector instructions of DIFFERENT waves on one SIMD overlap on gfx950?  One workgroup of 1024 threads (4 waves per SIMD):
// waves 0-7 (two per SIMD) run `nm` v_mfma_f32_16x16x32_bf16 per iteration, waves 8-15 run `nv` v_fma_f32 (or v_exp_f32) per iteration.
// Prints cycles per iteration for matrix only, vector only and both together.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_valu_overlap.hip -o tools/experiments/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int TRANS>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, int do_m, int do_v) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 8) {
    if (do_m)
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
  } else if (do_v) {
    for (int i = 0; i < iters; ++i) {
      if (TRANS)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
      else
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(1.0001f), "v"(1e-4f));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int TRANS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 16 * 8);
  const int iters = 4000;
  for (int mode = 0; mode < 3; ++mode) {
    const int dm = mode != 1, dv = mode != 0;
    k<TRANS><<<1, 1024>>>(out, cyc, iters, dm, dv);
    k<TRANS><<<1, 1024>>>(out, cyc, iters, dm, dv);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mm = 0, mv = 0;
    for (int i = 0; i < 8; ++i) mm = h[i] > mm ? h[i] : mm;
    for (int i = 8; i < 16; ++i) mv = h[i] > mv ? h[i] : mv;
    printf("%-10s %-12s matrix waves %.1f cycles / iteration (8 MFMA per SIMD), vector waves %.1f cycles / iteration (16 instr per SIMD)\n", name,
           mode == 0 ? "matrix only" : mode == 1 ? "vector only" : "both", (double)mm / iters, (double)mv / iters);
  }
}
int main() {
  run<0>("v_fma_f32");
  run<1>("v_exp_f32");
  return 0;
}
