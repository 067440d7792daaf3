// Detect head decode, one kernel per level: DFL softmax-expectation over reg_max bins -> ltrb distances ->
// dist2bbox around the cell-centre anchor -> xywh * stride; sigmoid on the class logits; written channel-major
// into the reference's (B, 4+nc, A) f32 layout.
// Replaces Detect._inference + DFL.forward + make_anchors + dist2bbox
// (ultralytics/nn/modules/head.py:151-169, nn/modules/block.py:250-253, utils/tal.py:352-376).
//
// HBM-bound: per anchor 4*reg_max + nc logits in, 4 + nc floats out.  One lane = one anchor: its logits are
// contiguous in NHWC (whole 128/160-byte lines per lane), the (B, C, A) stores are coalesced along the anchor axis.
#include "common.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One wave = 64 consecutive anchors of one level.  Their logits are brought into LDS by LDS-DMA as whole 16-byte groups
// (lanes sweep the channel groups of consecutive anchors, so global reads are full contiguous lines), XOR-swizzled by
// anchor so that the per-lane row reads that follow are bank-conflict free; outputs are written channel-major with the
// anchor on the lane (256-byte coalesced runs).  The one-lane-per-anchor direct form touched 64 different cache lines
// per load instruction and ran at 2 TB/s.
template <typename T, int REG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void detect_decode_kernel(const char* box, int ldb, const char* cls, int ldc, int N,
                                                                  int H, int W, int nc, float stride_px, float* y,
                                                                  int a_total, int a0) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int E = 16 / sizeof(T);
  constexpr int GB = 4 * REG / E;               // 16-byte groups per anchor, box part (8 bf16 / 16 f32)
  const int gc = (nc + E - 1) / E;              // groups per anchor, class part
  int gcp = 1;
  while (gcp < gc) gcp <<= 1;                   // padded to a power of two (XOR swizzle stays inside the row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HW = H * W;
  const long total = (long)N * HW;
  const long wbase = ((long)blockIdx.x * WAVES + wave) * 64;  // first anchor (flattened over images) of this wave
  char* bsm = sm + (size_t)wave * 64 * (GB + gcp) * 16;
  char* csm = bsm + 64 * GB * 16;
  // ---- DMA: box rows
  for (int k = 0; k < 64 * GB; k += 64) {
    const int item = k + lane;
    const int row = item / GB, slot = item % GB;
    const int grp = slot ^ (row & (GB - 1));
    long gid = wbase + row;
    if (gid >= total) gid = total - 1;
    const char* src = box + ((size_t)gid * ldb + grp * E) * sizeof(T);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bsm + k * 16), 16, 0, 0);
  }
  for (int k = 0; k < 64 * gcp; k += 64) {
    const int item = k + lane;
    const int row = item / gcp, slot = item & (gcp - 1);
    int grp = slot ^ (row & (gcp - 1));
    long gid = wbase + row;
    if (gid >= total) gid = total - 1;
    if (grp >= gc) grp = gc - 1;  // padding slots: any valid address (never read back)
    const char* src = cls + ((size_t)gid * ldc + grp * E) * sizeof(T);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(csm + k * 16), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const long gid = wbase + lane;
  if (gid >= total) return;
  const int n = (int)(gid / HW);
  const int a = (int)(gid - (long)n * HW);
  const int ay = a / W, ax = a - ay * W;
  float* yb = y + (size_t)n * (4 + nc) * a_total + a0 + a;

  auto unpack = [](const u32x4 t, float* v) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(t[i] << 16);
        v[2 * i + 1] = __uint_as_float(t[i] & 0xFFFF0000u);
      }
    }
  };
  // ---- box: 4 sides x REG bins (softmax expectation, block.py:250-253)
  float dist[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[REG];
#pragma unroll
    for (int q = 0; q < REG / E; ++q) {
      const int grp = s * (REG / E) + q;
      unpack(*reinterpret_cast<const u32x4*>(bsm + (lane * GB + (grp ^ (lane & (GB - 1)))) * 16), v + q * E);
    }
    float m = v[0];
#pragma unroll
    for (int i = 1; i < REG; ++i) m = fmaxf(m, v[i]);
    float sum = 0.f, e = 0.f;
#pragma unroll
    for (int i = 0; i < REG; ++i) {
      // bf16 perf mode: v_exp_f32 / v_rcp_f32 (1 ulp) - the logits carry 8 bits; f32 parity mode: ocml expf + IEEE divide
      const float ex = sizeof(T) == 2 ? __builtin_amdgcn_exp2f((v[i] - m) * 1.44269504088896340736f) : expf(v[i] - m);
      sum += ex;
      e += ex * (float)i;
    }
    dist[s] = sizeof(T) == 2 ? e * __builtin_amdgcn_rcpf(sum) : e / sum;
  }
  const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f;
  const float x1 = cx - dist[0], y1 = cy - dist[1], x2 = cx + dist[2], y2 = cy + dist[3];
  yb[0 * (size_t)a_total] = ((x1 + x2) / 2.f) * stride_px;
  yb[1 * (size_t)a_total] = ((y1 + y2) / 2.f) * stride_px;
  yb[2 * (size_t)a_total] = (x2 - x1) * stride_px;
  yb[3 * (size_t)a_total] = (y2 - y1) * stride_px;
  // ---- classes
  for (int grp = 0; grp < gc; ++grp) {
    float v[E];
    unpack(*reinterpret_cast<const u32x4*>(csm + (lane * gcp + (grp ^ (lane & (gcp - 1)))) * 16), v);
#pragma unroll
    for (int i = 0; i < E; ++i)
      if (grp * E + i < nc)
        yb[(size_t)(4 + grp * E + i) * a_total] =
            sizeof(T) == 2 ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i] * -1.44269504088896340736f))
                           : 1.0f / (1.0f + expf(-v[i]));
  }
}

extern "C" int upa_detect_decode(const void* box, int ldb, const void* cls, int ldc, int n, int h, int w, int reg_max,
                                 int nc, float stride_px, float* y, int a_total, int a0, int dtype, void* stream) {
  UPA_CHECK_ARG(box && cls && y, "detect_decode: null pointer");
  UPA_CHECK_ARG(reg_max == 16, "detect_decode: reg_max must be 16 (head.py:87)");
  const int E = 16 / upa_elem_size(dtype);
  // nc itself may be any value: the class row is read in whole 16-byte groups, so it must be readable (not meaningful) up to
  // the next multiple of 16 bytes - Detect pads its class slice accordingly
  UPA_CHECK_ARG(ldb % E == 0 && ldc % E == 0 && nc >= 1, "detect_decode: row strides must be multiples of 16 bytes");
  UPA_CHECK_ARG(ldc >= (nc + E - 1) / E * E, "detect_decode: class row shorter than nc rounded up to 16 bytes");
  UPA_CHECK_ARG(a0 >= 0 && a0 + h * w <= a_total, "detect_decode: level does not fit a_total");
  UPA_CHECK_ARG(nc <= 512, "detect_decode: nc too large for the LDS row");
  const long total = (long)n * h * w;
  constexpr int WAVES = 2;
  int gcp = 1;
  while (gcp < (nc + E - 1) / E) gcp <<= 1;
  const int gb = 4 * 16 / E;
  const size_t lds = (size_t)WAVES * 64 * (gb + gcp) * 16;
  dim3 grid((unsigned)((total + WAVES * 64 - 1) / (WAVES * 64)));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UPA_BF16) {
    auto kern = detect_decode_kernel<bf16_t, 16, WAVES>;
    (void)upa_full_lds<detect_decode_kernel<bf16_t, 16, WAVES>>();
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, (const char*)box, ldb, (const char*)cls, ldc, n, h, w, nc,
                       stride_px, y, a_total, a0);
  } else {
    auto kern = detect_decode_kernel<float, 16, WAVES>;
    (void)upa_full_lds<detect_decode_kernel<float, 16, WAVES>>();
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, (const char*)box, ldb, (const char*)cls, ldc, n, h, w, nc,
                       stride_px, y, a_total, a0);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
