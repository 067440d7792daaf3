// First-layer convolution: reads the model input in the reference's NCHW layout (f32 or bf16, <= 4 channels) and writes
// NHWC, so no separate layout-conversion pass touches the largest tensor of the network.
// Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 (k=3 s=2 for yolov8, k=6 s=2 p=2 for yolov5, k=3 s=1 v3).
//
// HBM-bound (K = cin*k*k <= 108 is too shallow for MFMA tiles): one thread = one output pixel x CO_T output channels,
// weights ([tap][ci][co] f32) broadcast from LDS, f32 FMA chain in (kh, kw, ci) order, coalesced reads along x,
// one contiguous CO_T*esize store per thread.
#include "common.h"

struct StemParams {
  const void* x;
  const float* w;
  const float* bias;
  char* y;
  int N, Cin, H, W, OH, OW, Cout, ldy, KS, stride, pad, act, x_bf16;
};

template <typename TO, int CO_T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];  // [tap][ci][CO_T] + bias[CO_T]
  const int co0 = blockIdx.y * CO_T;
  const int taps = p.KS * p.KS;
  for (int i = threadIdx.x; i < taps * p.Cin * CO_T; i += 256) {
    const int co = i % CO_T;
    const int ci = (i / CO_T) % p.Cin;
    const int tap = i / (CO_T * p.Cin);
    const int kh = tap / p.KS, kw = tap % p.KS;
    float v = 0.f;
    if (co0 + co < p.Cout) v = p.w[(((size_t)(co0 + co) * p.Cin + ci) * p.KS + kh) * p.KS + kw];
    wsh[i] = v;
  }
  if (threadIdx.x < CO_T) {
    const int co = co0 + threadIdx.x;
    wsh[taps * p.Cin * CO_T + threadIdx.x] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
  }
  __syncthreads();
  const long total = (long)p.N * p.OH * p.OW;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ox = (int)(gid % p.OW);
  const int oy = (int)((gid / p.OW) % p.OH);
  const int n = (int)(gid / ((long)p.OW * p.OH));
  float acc[CO_T];
#pragma unroll
  for (int c = 0; c < CO_T; ++c) acc[c] = 0.f;
  const size_t plane = (size_t)p.H * p.W;
  for (int kh = 0; kh < p.KS; ++kh) {
    const int iy = oy * p.stride - p.pad + kh;
    if (iy < 0 || iy >= p.H) continue;
    for (int kw = 0; kw < p.KS; ++kw) {
      const int ix = ox * p.stride - p.pad + kw;
      if (ix < 0 || ix >= p.W) continue;
      const float* wt = wsh + (kh * p.KS + kw) * p.Cin * CO_T;
      for (int ci = 0; ci < p.Cin; ++ci) {
        const size_t off = ((size_t)n * p.Cin + ci) * plane + (size_t)iy * p.W + ix;
        const float xv = p.x_bf16 ? bf16_to_f32(((const bf16_t*)p.x)[off]) : ((const float*)p.x)[off];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = fmaf(xv, wt[ci * CO_T + c], acc[c]);
      }
    }
  }
  const float* bsh = wsh + taps * p.Cin * CO_T;
  char* dst = p.y + ((size_t)gid * p.ldy + co0) * sizeof(TO);
  constexpr bool F32 = sizeof(TO) == 4;
#pragma unroll
  for (int c = 0; c < CO_T; c += 4) {
    if (co0 + c >= p.Cout) break;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = acc[c + q] + bsh[c + q];
      if (p.act == UPA_ACT_SILU) t = F32 ? t / (1.0f + expf(-t)) : t * __frcp_rn(1.0f + __expf(-t));
      v[q] = t;
    }
    if constexpr (F32)
      *reinterpret_cast<f32x4*>(dst + c * 4) = f32x4{v[0], v[1], v[2], v[3]};
    else
      *reinterpret_cast<u32x2*>(dst + c * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}

extern "C" int upa_conv2d_stem_nchw(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt,
                                    const float* bias, void* y, int cout, int ldy, int k, int stride, int pad, int act,
                                    int dtype, void* stream) {
  UPA_CHECK_ARG(x && wt && y, "stem: null pointer");
  UPA_CHECK_ARG(cin >= 1 && cin <= 4 && cout % 4 == 0 && ldy % 4 == 0, "stem: cin must be <=4, cout %% 4 == 0");
  UPA_CHECK_ARG(k >= 1 && k <= 7 && stride >= 1 && pad >= 0, "stem: bad k/s/p");
  StemParams p;
  p.x = x; p.w = wt; p.bias = bias; p.y = (char*)y;
  p.N = n; p.Cin = cin; p.H = h; p.W = w;
  p.OH = (h + 2 * pad - k) / stride + 1;
  p.OW = (w + 2 * pad - k) / stride + 1;
  p.Cout = cout; p.ldy = ldy; p.KS = k; p.stride = stride; p.pad = pad; p.act = act; p.x_bf16 = (x_dtype == UPA_BF16);
  constexpr int CO_T = 16;
  const long total = (long)n * p.OH * p.OW;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)cdiv(cout, CO_T));
  const size_t lds = ((size_t)k * k * cin * CO_T + CO_T) * sizeof(float);
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL((stem_conv_kernel<bf16_t, CO_T>), grid, dim3(256), lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((stem_conv_kernel<float, CO_T>), grid, dim3(256), lds, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
