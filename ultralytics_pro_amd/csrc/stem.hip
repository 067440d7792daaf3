// First-layer convolution: reads the model input in the reference's NCHW layout (f32 or bf16, <= 4 channels) and writes
// NHWC, so no separate layout-conversion pass touches the largest tensor of the network.
// Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 (k=3 s=2 for yolov8, k=6 s=2 p=2 for yolov5, k=3 s=1 v3).
//
// HBM-bound (K = cin*k*k <= 108 is too shallow for MFMA tiles): one thread = one output pixel x CO_T output channels,
// weights ([tap][ci][co] f32) broadcast from LDS, f32 FMA chain in (kh, kw, ci) order, coalesced reads along x,
// one contiguous CO_T*esize store per thread.
#include <stdlib.h>

#include "common.h"

struct StemParams {
  const void* x;
  const float* w;     // repacked [tap][ci][co16 group][16] f32 (device)
  const float* bias;
  char* y;
  int N, Cin, H, W, OH, OW, Cout, ldy, KS, stride, pad, act, x_bf16;
  int TH, TW, tilesX, tilesY, PR, PC, PCS;  // output tile, patch rows/cols, padded LDS row stride
  int ablate;
};

// One workgroup = TH x TW output pixels (TH*TW = 256, one per lane) x 16 output channels of one image.
// The input patch ((TH-1)s+k) x ((TW-1)s+k) x cin is staged in LDS as f32 by coalesced row reads (bf16 input: aligned
// 4-byte pairs), so every input element is fetched from HBM once per workgroup instead of k*k/s^2 times with 2-byte
// accesses; weights are read through wave-uniform addresses (scalar loads), the FMA chain runs in (kh, kw, ci) order.
template <typename TO, int CO_T, bool SILU>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) float patch[];  // [cin][PR][PCS]
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
  const int co0 = blockIdx.y * CO_T;
  const size_t plane = (size_t)p.H * p.W;
  // ---- stage.  A thread owns one patch column (bf16: one aligned pair of columns) and walks the cin*PR patch lines in
  // steps of LSTEP, four lines in flight: no per-element division (only incremental line -> (ci,row) counters).
  if (p.x_bf16 == 2) {
    // uint8 HWC BGR frames (what the reference's predictor receives, engine/predictor.py:151-173): BGR->RGB,
    // HWC->CHW, uint8->float and /255 all happen in this load - the separate preprocessing passes disappear.
    const unsigned char* xu = (const unsigned char*)p.x;
    const int npix = p.PR * p.PC;
    for (int base = 0; base < npix; base += 256 * 4) {
      unsigned v[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + q * 256 + tid;
        v[q][0] = v[q][1] = v[q][2] = 0u;
        if (i < npix) {
          const int row = i / p.PC, col = i - row * p.PC;
          const int iy = iy0 + row, ix = ix0 + col;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && !(p.ablate & 1)) {
            const unsigned char* px = xu + (((size_t)n * p.H + iy) * p.W + ix) * 3;
            v[q][0] = px[2]; v[q][1] = px[1]; v[q][2] = px[0];  // channel reversal
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + q * 256 + tid;
        if (i < npix) {
          const int row = i / p.PC, col = i - row * p.PC;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci)
            patch[(ci * p.PR + row) * p.PCS + col] = (float)v[q][ci] / 255.0f;  // `im.float(); im /= 255`
        }
      }
    }
  } else {
    const bool bf = p.x_bf16 != 0;
    const int ixa = bf ? (ix0 & ~1) : ix0;   // bf16: even start -> 4-byte aligned pairs (two's complement floors)
    const int shift = ix0 - ixa;             // 0 or 1
    const int ncols = bf ? (p.PC + shift + 1) / 2 : p.PC;   // columns (pairs) per line
    int span = 32;
    while (span < ncols) span <<= 1;         // lanes per line: 32, 64, 128 or 256
    const int LSTEP = 256 / span;            // lines covered per pass
    const int colid = tid & (span - 1);
    const int line0 = tid / span;            // span is a power of two: a shift
    const int nlines = p.Cin * p.PR;
    const bool colOK = colid < ncols;
    const int ix = bf ? ixa + 2 * colid : ix0 + colid;
    const bool evenW = (p.W & 1) == 0;
    const bool pairFast = bf && ix >= 0 && ix + 1 < p.W && evenW;
    const bf16_t* xb = (const bf16_t*)p.x;
    const float* xf = (const float*)p.x;
    int ci = 0, row = line0;                  // line0 < PR always (LSTEP <= 8 <= PR for k >= 3 ... guarded below)
    while (row >= p.PR) { row -= p.PR; ++ci; }
    for (int line = line0; line < nlines; line += 4 * LSTEP) {
      unsigned u[4];
      int lci[4], lrow[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u[q] = 0u;
        lci[q] = ci;
        lrow[q] = row;
        const int iy = iy0 + row;
        if (colOK && line + q * LSTEP < nlines && iy >= 0 && iy < p.H && !(p.ablate & 1)) {
          const size_t rb = ((size_t)n * p.Cin + ci) * plane + (size_t)iy * p.W;
          if (bf) {
            if (pairFast) {
              u[q] = *reinterpret_cast<const unsigned*>(xb + rb + ix);
            } else {
              unsigned lo = 0, hi = 0;
              if (ix >= 0 && ix < p.W) lo = xb[rb + ix];
              if (ix + 1 >= 0 && ix + 1 < p.W) hi = xb[rb + ix + 1];
              u[q] = lo | (hi << 16);
            }
          } else if (ix >= 0 && ix < p.W) {
            u[q] = __float_as_uint(xf[rb + ix]);
          }
        }
        row += LSTEP;
        while (row >= p.PR) { row -= p.PR; ++ci; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!colOK || line + q * LSTEP >= nlines) continue;
        float* dst = patch + (lci[q] * p.PR + lrow[q]) * p.PCS;
        if (bf) {
          const int c0 = 2 * colid - shift;
          if (c0 >= 0 && c0 < p.PC) dst[c0] = __uint_as_float(u[q] << 16);
          if (c0 + 1 >= 0 && c0 + 1 < p.PC) dst[c0 + 1] = __uint_as_float(u[q] & 0xFFFF0000u);
        } else {
          dst[colid] = __uint_as_float(u[q]);
        }
      }
    }
  }
  __syncthreads();
  // ---- compute: lane = one output pixel
  const int ty = tid / p.TW, tx = tid - ty * p.TW;
  const int oy = oy0 + ty, ox = ox0 + tx;
  float acc[CO_T];  // starts from the bias (fetched once, all loads in flight together)
#pragma unroll
  for (int c = 0; c < CO_T; ++c) acc[c] = (p.bias && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
  const float* wg = p.w + (size_t)blockIdx.y * CO_T;  // [tap][ci][groups*16]: uniform addresses -> scalar loads
  const int wstride = ((p.Cout + CO_T - 1) / CO_T) * CO_T;
  for (int kh = 0; kh < ((p.ablate & 2) ? 0 : p.KS); ++kh) {
    for (int kw = 0; kw < p.KS; ++kw) {
      for (int ci = 0; ci < p.Cin; ++ci) {
        const float xv = patch[(ci * p.PR + ty * p.stride + kh) * p.PCS + tx * p.stride + kw];
        const float* wt = wg + (size_t)((kh * p.KS + kw) * p.Cin + ci) * wstride;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = fmaf(xv, wt[c], acc[c]);
      }
    }
  }
  if (oy >= p.OH || ox >= p.OW || (p.ablate & 4)) return;
  const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
  char* dst = p.y + (pix * p.ldy + co0) * sizeof(TO);
  constexpr bool F32 = sizeof(TO) == 4;
  float v[CO_T];
#pragma unroll
  for (int c = 0; c < CO_T; ++c) {
    float t = acc[c];
    if constexpr (SILU) t = F32 ? t / (1.0f + expf(-t)) : t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
    v[c] = t;
  }
  if constexpr (F32) {
#pragma unroll
    for (int c = 0; c < CO_T; c += 4)
      if (co0 + c < p.Cout) *reinterpret_cast<f32x4*>(dst + c * 4) = f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]};
  } else {
#pragma unroll
    for (int c = 0; c < CO_T; c += 8) {
      if (co0 + c + 4 < p.Cout)
        *reinterpret_cast<u32x4*>(dst + c * 2) = u32x4{pack_bf16x2(v[c], v[c + 1]), pack_bf16x2(v[c + 2], v[c + 3]),
                                                       pack_bf16x2(v[c + 4], v[c + 5]), pack_bf16x2(v[c + 6], v[c + 7])};
      else if (co0 + c < p.Cout)
        *reinterpret_cast<u32x2*>(dst + c * 2) = u32x2{pack_bf16x2(v[c], v[c + 1]), pack_bf16x2(v[c + 2], v[c + 3])};
    }
  }
}

// Host-side repack of OIHW f32 weights into [tap][ci][co padded to 16] (HOST memory in, HOST memory out).
extern "C" size_t upa_stem_packed_weight_bytes(int cout, int cin, int k) {
  return (size_t)k * k * cin * ((cout + 15) / 16 * 16) * sizeof(float);
}
extern "C" int upa_pack_stem_weight(const float* w_oihw, int cout, int cin, int k, float* out) {
  UPA_CHECK_ARG(w_oihw && out && cout > 0 && cin > 0 && k >= 1 && k <= 7, "pack_stem_weight: bad args");
  const int cop = (cout + 15) / 16 * 16;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw)
      for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cop; ++co)
          out[((size_t)(kh * k + kw) * cin + ci) * cop + co] =
              co < cout ? w_oihw[(((size_t)co * cin + ci) * k + kh) * k + kw] : 0.f;
  return UPA_OK;
}

extern "C" int upa_conv2d_stem_nchw(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt,
                                    const float* bias, void* y, int cout, int ldy, int k, int stride, int pad, int act,
                                    int dtype, void* stream) {
  UPA_CHECK_ARG(x && wt && y, "stem: null pointer");
  UPA_CHECK_ARG(x_dtype == UPA_F32 || x_dtype == UPA_BF16 || (x_dtype == UPA_U8_BGR_HWC && cin == 3),
                "stem: input must be NCHW f32/bf16 or NHWC uint8 BGR with 3 channels");
  UPA_CHECK_ARG(cin >= 1 && cin <= 4 && cout % 4 == 0 && ldy % 8 == 0 && (uintptr_t)y % 16 == 0,
                "stem: cin must be <=4, cout %% 4 == 0, output view 16-byte aligned");
  UPA_CHECK_ARG(k >= 1 && k <= 7 && stride >= 1 && stride <= 2 && pad >= 0, "stem: bad k/s/p");
  StemParams p;
  p.x = x; p.w = wt; p.bias = bias; p.y = (char*)y;
  p.N = n; p.Cin = cin; p.H = h; p.W = w;
  p.OH = (h + 2 * pad - k) / stride + 1;
  p.OW = (w + 2 * pad - k) / stride + 1;
  p.Cout = cout; p.ldy = ldy; p.KS = k; p.stride = stride; p.pad = pad; p.act = act; p.x_bf16 = x_dtype == UPA_BF16 ? 1 : (x_dtype == UPA_U8_BGR_HWC ? 2 : 0);
  p.TW = p.OW >= 64 ? 64 : (p.OW >= 32 ? 32 : 16);
  p.TH = 256 / p.TW;
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  p.PR = (p.TH - 1) * stride + k;
  p.PC = (p.TW - 1) * stride + k;
  p.PCS = p.PC + 1;
  static const int ablate = getenv("UPA_STEM_ABLATE") ? atoi(getenv("UPA_STEM_ABLATE")) : 0;
  p.ablate = ablate;
  constexpr int CO_T = 16;
  dim3 grid((unsigned)((long)p.tilesX * p.tilesY * n), (unsigned)cdiv(cout, CO_T));
  const size_t lds = (size_t)cin * p.PR * p.PCS * sizeof(float);
  UPA_CHECK_ARG(lds <= 64 * 1024, "stem: patch does not fit LDS");
  UPA_CHECK_ARG(act == UPA_ACT_SILU || act == UPA_ACT_NONE, "stem: activation must be SiLU or none");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UPA_BF16) {
    if (act == UPA_ACT_SILU) hipLaunchKernelGGL((stem_conv_kernel<bf16_t, CO_T, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((stem_conv_kernel<bf16_t, CO_T, false>), grid, dim3(256), lds, st, p);
  } else {
    if (act == UPA_ACT_SILU) hipLaunchKernelGGL((stem_conv_kernel<float, CO_T, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((stem_conv_kernel<float, CO_T, false>), grid, dim3(256), lds, st, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
