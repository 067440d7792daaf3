// Detect head decode, one kernel per level: DFL softmax-expectation over reg_max bins -> ltrb distances ->
// dist2bbox around the cell-centre anchor -> xywh * stride; sigmoid on the class logits; written channel-major
// into the reference's (B, 4+nc, A) f32 layout.
// Replaces Detect._inference + DFL.forward + make_anchors + dist2bbox
// (ultralytics/nn/modules/head.py:151-169, nn/modules/block.py:250-253, utils/tal.py:352-376).
//
// HBM-bound: per anchor 4*reg_max + nc logits in, 4 + nc floats out.  One lane = one anchor: its logits are
// contiguous in NHWC (whole 128/160-byte lines per lane), the (B, C, A) stores are coalesced along the anchor axis.
#include "common.h"

template <typename T, int REG>
__global__ __launch_bounds__(256) void detect_decode_kernel(const char* box, int ldb, const char* cls, int ldc, int N, int H,
                                                            int W, int nc, float stride_px, float* y, int a_total, int a0) {
  constexpr int E = 16 / sizeof(T);
  const int HW = H * W;
  const long total = (long)N * HW;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int n = (int)(gid / HW);
  const int a = (int)(gid - (long)n * HW);
  const int ay = a / W, ax = a - ay * W;
  float* yb = y + (size_t)n * (4 + nc) * a_total + a0 + a;

  // ---- box: 4 sides x REG bins
  const char* bp = box + (size_t)gid * ldb * sizeof(T);
  float dist[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[REG];
#pragma unroll
    for (int q = 0; q < REG / E; ++q) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(bp + (s * REG + q * E) * sizeof(T));
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[q * 4 + i] = __uint_as_float(t[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[q * 8 + 2 * i] = __uint_as_float(t[i] << 16);
          v[q * 8 + 2 * i + 1] = __uint_as_float(t[i] & 0xFFFF0000u);
        }
      }
    }
    float m = v[0];
#pragma unroll
    for (int i = 1; i < REG; ++i) m = fmaxf(m, v[i]);
    float sum = 0.f, e = 0.f;
#pragma unroll
    for (int i = 0; i < REG; ++i) {
      const float ex = expf(v[i] - m);
      sum += ex;
      e += ex * (float)i;
    }
    dist[s] = e / sum;
  }
  const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f;
  const float x1 = cx - dist[0], y1 = cy - dist[1], x2 = cx + dist[2], y2 = cy + dist[3];
  yb[0 * (size_t)a_total] = ((x1 + x2) / 2.f) * stride_px;
  yb[1 * (size_t)a_total] = ((y1 + y2) / 2.f) * stride_px;
  yb[2 * (size_t)a_total] = (x2 - x1) * stride_px;
  yb[3 * (size_t)a_total] = (y2 - y1) * stride_px;

  // ---- classes
  const char* cp = cls + (size_t)gid * ldc * sizeof(T);
  for (int c0 = 0; c0 < nc; c0 += E) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(cp + c0 * sizeof(T));
    float v[E];
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
#pragma unroll
    for (int i = 0; i < E; ++i)
      if (c0 + i < nc) yb[(size_t)(4 + c0 + i) * a_total] = 1.0f / (1.0f + expf(-v[i]));
  }
}

extern "C" int upa_detect_decode(const void* box, int ldb, const void* cls, int ldc, int n, int h, int w, int reg_max,
                                 int nc, float stride_px, float* y, int a_total, int a0, int dtype, void* stream) {
  UPA_CHECK_ARG(box && cls && y, "detect_decode: null pointer");
  UPA_CHECK_ARG(reg_max == 16, "detect_decode: reg_max must be 16 (head.py:87)");
  const int E = 16 / upa_elem_size(dtype);
  UPA_CHECK_ARG(ldb % E == 0 && ldc % E == 0 && nc % E == 0, "detect_decode: strides / nc must be multiples of 16 bytes");
  UPA_CHECK_ARG(a0 >= 0 && a0 + h * w <= a_total, "detect_decode: level does not fit a_total");
  const long total = (long)n * h * w;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL((detect_decode_kernel<bf16_t, 16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)box, ldb,
                       (const char*)cls, ldc, n, h, w, nc, stride_px, y, a_total, a0);
  else
    hipLaunchKernelGGL((detect_decode_kernel<float, 16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)box, ldb,
                       (const char*)cls, ldc, n, h, w, nc, stride_px, y, a_total, a0);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
