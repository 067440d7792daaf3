// Issue rate of plain / packed / transcendental f32 VALU on gfx950 with 1, 2 and 4 waves per SIMD (one workgroup of 256 / 512 / 1024 threads on
// one CU; every wave runs the same unrolled stream of independent instructions).  Prints shader cycles per wave-instruction per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/valu_rate.hip -o tools/experiments/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const f32x2 c = {1.0001f, 0.9999f}, d = {1e-4f, -1e-4f};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) {  // 8 independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c[0]), "v"(d[0]));
      } else if (MODE == 1) {  // 4 v_pk_fma_f32 (8 results)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(d));
      } else if (MODE == 2) {  // 8 v_exp_f32
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 3) {  // 4 v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
      } else if (MODE == 4) {  // 8 v_rcp_f32
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 5) {  // 8 v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                     "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 6) {  // 8 s_mul_i32-free SALU: s_add
        asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                     ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
}

template <int MODE>
void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 16 * 8);
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    k<MODE><<<1, threads>>>(out, cyc, iters);
    k<MODE><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    const int waves = threads / 64;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    const int wps = waves >= 4 ? waves / 4 : 1;  // waves per SIMD
    const double instr_per_simd = (double)iters * 8 * per_iter * wps;
    printf("%-22s %4d threads (%d wave%s/SIMD): %.2f cycles per wave-instruction per SIMD\n", name, threads, wps, wps > 1 ? "s" : "", (double)mx / instr_per_simd);
  }
}

int main() {
  run<0>("v_fma_f32", 8);
  run<1>("v_pk_fma_f32", 4);
  run<3>("v_pk_mul_f32", 4);
  run<2>("v_exp_f32", 8);
  run<4>("v_rcp_f32", 8);
  run<5>("v_cvt_pk_bf16_f32", 8);
  run<6>("s_add_u32", 8);
  return 0;
}
