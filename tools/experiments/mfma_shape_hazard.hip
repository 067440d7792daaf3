// Reproducer for the accumulator hazard csrc/c2f_stream.hip works around (gfx950, ROCm 7.2): a 4-pass v_mfma_f32_16x16x16_bf16 whose
// result is the srcC of the NEXT instruction, an 8-pass v_mfma_f32_16x16x32_bf16, with no wait state between them.
// Three variants of acc = MFMA32(A, B, MFMA16(a, b, c)) on the same random operands in every lane:
//   back to back (inline asm, nothing between the two instructions), with s_nop 3 / 7 / 15 between them, and the compiler's own code for the
//   two intrinsics; each compared element by element with the separated form (16-wide chain, s_nop 15 x 2, 32-wide chain).
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_shape_hazard.hip -o tools/experiments/mfma_shape_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ void k(const u32x2* a16, const u32x2* b16, const u32x4* a32, const u32x4* b32, const f32x4* c, f32x4* out) {
  const int l = threadIdx.x;
  u32x2 x16 = a16[l], y16 = b16[l];
  u32x4 x32 = a32[l], y32 = b32[l];
  f32x4 acc = c[l];
  if (MODE == 0) {  // reference: separated
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 15\n s_nop 15\n v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else if (MODE == 1) {  // back to back
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else if (MODE == 2) {
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 3\n v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else if (MODE == 3) {
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 7\n v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else if (MODE == 4) {  // what hipcc emits for the two intrinsics
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<s16x4*>(&x16), *reinterpret_cast<s16x4*>(&y16), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&x32), *reinterpret_cast<bf16x8*>(&y32), acc, 0, 0, 0);
  } else if (MODE == 7) {  // the sequence of csrc/c2f_stream.hip's first cv2 form: two 16-wide steps, then the 32-wide chain, as hipcc schedules it
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<s16x4*>(&x16), *reinterpret_cast<s16x4*>(&y16), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<s16x4*>(&y16), *reinterpret_cast<s16x4*>(&x16), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&x32), *reinterpret_cast<bf16x8*>(&y32), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&y32), *reinterpret_cast<bf16x8*>(&x32), acc, 0, 0, 0);
  } else if (MODE == 8) {  // ... and its reference (every step separated)
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 15\n v_mfma_f32_16x16x16_bf16 %0, %2, %1, %0\n s_nop 15\n s_nop 15\n"
                 "v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15\n v_mfma_f32_16x16x32_bf16 %0, %4, %3, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else if (MODE == 5) {  // the reverse order, back to back: 8-pass result as srcC of a 4-pass instruction
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  } else {  // reference for MODE 5
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_nop 15\n s_nop 15\n v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n s_nop 15\n s_nop 15"
                 : "+v"(acc) : "v"(x16), "v"(y16), "v"(x32), "v"(y32));
  }
  out[l] = acc;
}

static unsigned short rbf() { float f = (float)rand() / RAND_MAX - 0.5f; unsigned u; ::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main() {
  unsigned short h16a[64 * 4], h16b[64 * 4], h32a[64 * 8], h32b[64 * 8];
  float hc[64 * 4];
  srand(1);
  for (auto& v : h16a) v = rbf();
  for (auto& v : h16b) v = rbf();
  for (auto& v : h32a) v = rbf();
  for (auto& v : h32b) v = rbf();
  for (auto& v : hc) v = (float)rand() / RAND_MAX;
  void *a16, *b16, *a32, *b32, *c, *out;
  (void)hipMalloc(&a16, sizeof h16a); (void)hipMalloc(&b16, sizeof h16b); (void)hipMalloc(&a32, sizeof h32a); (void)hipMalloc(&b32, sizeof h32b);
  (void)hipMalloc(&c, sizeof hc); (void)hipMalloc(&out, sizeof hc);
  (void)hipMemcpy(a16, h16a, sizeof h16a, hipMemcpyHostToDevice); (void)hipMemcpy(b16, h16b, sizeof h16b, hipMemcpyHostToDevice);
  (void)hipMemcpy(a32, h32a, sizeof h32a, hipMemcpyHostToDevice); (void)hipMemcpy(b32, h32b, sizeof h32b, hipMemcpyHostToDevice);
  (void)hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
  float ref[256], ref5[256], ref7[256], got[256];
  auto run = [&](int mode, float* dst) {
    switch (mode) {
      case 0: k<0><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 1: k<1><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 2: k<2><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 3: k<3><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 4: k<4><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 5: k<5><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 7: k<7><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      case 8: k<8><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
      default: k<6><<<1, 64>>>((u32x2*)a16, (u32x2*)b16, (u32x4*)a32, (u32x4*)b32, (f32x4*)c, (f32x4*)out); break;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(dst, out, sizeof hc, hipMemcpyDeviceToHost);
  };
  run(0, ref);
  run(6, ref5);
  run(8, ref7);
  const char* names[] = {"", "16x16x16 -> 16x16x32 back to back", "... s_nop 3 between", "... s_nop 7 between", "... as hipcc schedules the two intrinsics",
                         "16x16x32 -> 16x16x16 back to back", "", "16, 16, 32, 32 chain as hipcc schedules it"};
  for (int m = 1; m <= 7; ++m) {
    if (m == 6) continue;
    run(m, got);
    const float* r = m == 5 ? ref5 : m == 7 ? ref7 : ref;
    int bad = 0, bad_e[4] = {0, 0, 0, 0};
    float worst = 0;
    for (int i = 0; i < 256; ++i)
      if (got[i] != r[i]) { ++bad; ++bad_e[i & 3]; float d = got[i] - r[i]; if (d < 0) d = -d; if (d > worst) worst = d; }
    printf("%-46s wrong elements %3d of 256 (by accumulator row 0..3: %d %d %d %d), max |d| %.4g\n", names[m], bad, bad_e[0], bad_e[1], bad_e[2], bad_e[3], worst);
  }
  return 0;
}
