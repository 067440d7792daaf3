// Training-step kernels (SURVEY 8f rank 2 / BASELINE config 3): what the reference obtains from torch autograd for
// Conv = conv2d -> BatchNorm2d(batch statistics) -> SiLU (ultralytics/nn/modules/conv.py:177-186), SPPF max pools
// (nn/modules/block.py:402-406), nn.Upsample, and the optimizer step of engine/trainer.py:674-682.
//   * upa_pack_conv_weight_dev    device-side repack of the f32 master weights (forward layout, or transposed + flipped
//                                 for the data gradient, which then runs on the forward conv kernels)
//   * upa_bn_stats / upa_bn_finalize   per-channel batch mean / biased variance (f64 combination), running-stat update
//   * upa_bn_act_fwd              y = act(gamma * (z - mean) * rstd + beta) (+ residual)
//   * upa_bn_act_bwd_reduce/_apply   d(gamma), d(beta) and dz through activation + batch-stat BN
//   * upa_conv2d_wgrad            dW = sum over pixels of dz (x) x on exact-f32 MFMA (v_mfma_f32_16x16x4_f32): both
//                                 operands are pixel-major (NHWC), which is exactly the k-per-lane layout of the f32
//                                 MFMA - no transposition anywhere; persistent workgroups accumulate in registers
//   * upa_channel_sum             bias gradient of the plain nn.Conv2d head outputs
//   * upa_dilate2x, upa_upsample2x_bwd, upa_maxpool_bwd
//   * upa_sumsq, upa_sgd_nesterov_ema   gradient norm, clipped SGD(nesterov) + weight decay + EMA in one pass
// All HBM-bound except wgrad (f32 MFMA bound).
#include <stdlib.h>

#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ void load16(const char* p, float* v);
template <> __device__ __forceinline__ void load16<float>(const char* p, float* v) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load16<bf16_t>(const char* p, float* v) {
  const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(t[i] << 16);
    v[2 * i + 1] = __uint_as_float(t[i] & 0xFFFF0000u);
  }
}
// 16 raw bytes of a view now, their E floats later (software prefetch: the loads of the next loop trip are in flight while this one's
// values are worked on)
template <typename T> __device__ __forceinline__ void cvt16(const u32x4& t, float* v);
template <> __device__ __forceinline__ void cvt16<float>(const u32x4& t, float* v) {
  v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
}
template <> __device__ __forceinline__ void cvt16<bf16_t>(const u32x4& t, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(t[i] << 16);
    v[2 * i + 1] = __uint_as_float(t[i] & 0xFFFF0000u);
  }
}
template <typename T> __device__ __forceinline__ void store16(char* p, const float* v);
template <> __device__ __forceinline__ void store16<float>(char* p, const float* v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store16<bf16_t>(char* p, const float* v) {
  *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                       pack_bf16x2(v[6], v[7])};
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_weight_body(const float* w, int cout, int cin, int k, int E, int transpose_flip, void* out,
                                                 long total) {
  // logical weight V[co'][ci'][kh][kw]; transpose_flip: V[a][b][kh][kw] = W[b][a][k-1-kh][k-1-kw] (data gradient)
  const int lc_out = transpose_flip ? cin : cout, lc_in = transpose_flip ? cout : cin;
  const int ktch = 4 * E;
  const int ktt = (lc_in + ktch - 1) / ktch, ntn = (lc_out + 15) / 16;
  // 32-bit index arithmetic (a conv weight has far fewer than 2^31 packed elements; E and 64 are powers of two): with `long`
  // the five divisions per element made the whole-model repack 107 us at the head of every training step
  const unsigned tot = (unsigned)total, eshift = E == 8 ? 3u : 2u;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += gridDim.x * blockDim.x) {
    unsigned t = idx;
    const int j = (int)(t & (unsigned)(E - 1)); t >>= eshift;
    const int lane = (int)(t & 63u); t >>= 6;
    const int nt = (int)(t % (unsigned)ntn); t /= (unsigned)ntn;
    const int kt = (int)(t % (unsigned)ktt); t /= (unsigned)ktt;
    const int tap = (int)t;
    const int kh = tap / k, kw = tap - kh * k;
    const int g = lane >> 4, r = lane & 15;
    const int co = nt * 16 + r, ci = kt * ktch + g * E + j;
    float v = 0.f;
    if (co < lc_out && ci < lc_in) {
      if (transpose_flip) v = w[(((size_t)ci * cin + co) * k + (k - 1 - kh)) * k + (k - 1 - kw)];
      else v = w[(((size_t)co * cin + ci) * k + kh) * k + kw];
    }
    if (E == 8) ((bf16_t*)out)[idx] = f32_to_bf16(v);
    else ((float*)out)[idx] = v;
  }
}

__global__ void pack_weight_dev_kernel(const float* w, int cout, int cin, int k, int E, int transpose_flip, void* out,
                                       long total) {
  pack_weight_body(w, cout, cin, k, E, transpose_flip, out, total);
}

// every conv of a model in one launch: blockIdx.y = descriptor (the per-layer form cost 147 launches of 4.5 us per
// training step of yolov8s)
__global__ void pack_weight_batched_kernel(const UpaPackDesc* descs) {
  const UpaPackDesc d = descs[blockIdx.y];
  const int E = d.dtype == UPA_BF16 ? 8 : 4;
  const int lc_out = d.transpose_flip ? d.cin : d.cout, lc_in = d.transpose_flip ? d.cout : d.cin;
  const long total = (long)d.k * d.k * ((lc_in + 4 * E - 1) / (4 * E)) * ((lc_out + 15) / 16) * 64 * E;
  pack_weight_body(d.w_oihw, d.cout, d.cin, d.k, E, d.transpose_flip, d.out, total);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-channel reductions over an NHWC view.  MODE 0: sum(z), sum(z^2).  MODE 1 (BN+act backward): sum(du),
// sum(du * xhat) with du = dy * act'(gamma*xhat+beta).  MODE 2: sum(z) only (bias gradient).
// Thread = one 16-byte channel group of a strided set of pixels; block partials meet in LDS (f64), one f64 atomic per
// channel per block reaches HBM.
struct ReduceParams {
  const char* z; const char* dy;
  long npix; int c, ldz, lddy;
  const float* mean; const float* var; const float* gamma; const float* beta;
  float eps; int act;
  double* out0; double* out1;
};

// FAST (bf16 perf mode): v_exp_f32 / v_rcp_f32, 1 ulp - the operands carry 8 bits; the f32 parity mode keeps ocml expf
// and the IEEE divide.  (These element-wise passes are VALU-bound, not HBM-bound, with the accurate forms.)
template <bool FAST>
__device__ __forceinline__ float sigmoid_t(float u) {
  if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896340736f));
  else return 1.0f / (1.0f + expf(-u));
}
template <bool FAST>
__device__ __forceinline__ float silu_grad(float u) {
  const float s = sigmoid_t<FAST>(u);
  return s * (1.0f + u * (1.0f - s));
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void channel_reduce_kernel(const ReduceParams p) {
  extern __shared__ double red[];  // [2][c]
  constexpr int E = 16 / sizeof(T);
  const int cg = p.c / E;
  const int ppb = 256 / cg;  // pixels per pass (>= 1: c <= 256*E checked by the launcher)
  const int tid = threadIdx.x;
  const int grp = tid % cg, sub = tid / cg;
  for (int i = tid; i < 2 * p.c; i += 256) red[i] = 0.0;
  __syncthreads();
  float s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s0[e] = s1[e] = 0.f;
  float mean[E], rstd[E], gam[E], bet[E];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int ch = grp * E + e;
      mean[e] = p.mean[ch]; rstd[e] = 1.0f / sqrtf(p.var[ch] + p.eps); gam[e] = p.gamma[ch]; bet[e] = p.beta[ch];
    }
  }
  if (sub < ppb) {
    const long chunk = (p.npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * chunk;
    long p1 = p0 + chunk;
    if (p1 > p.npix) p1 = p.npix;
    int cnt = 0;
    constexpr int U = 4;  // pixels per thread and loop trip
    // Software-pipelined: the raw 16-byte packets of trip t + 1 are requested before trip t's values are worked on.  With two
    // workgroups per CU (512 blocks) a wave otherwise alternates between one memory round trip and ~500 vector instructions
    // (MODE 1: sigmoid and its derivative per element) and the launch runs at a third of the HBM rate (yolov8s training step:
    // 1.8 TB/s against the 4.3 TB/s of the apply pass that follows it).
    u32x4 rz[U], rd[U];
    auto request = [&](long px) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long q = px + (long)u * ppb;
        const long qq = q < p1 ? q : p0 + sub;  // past the end: any valid pixel (its values are masked at use)
        rz[u] = *reinterpret_cast<const u32x4*>(p.z + ((size_t)qq * p.ldz + grp * E) * sizeof(T));
        if (MODE == 1) rd[u] = *reinterpret_cast<const u32x4*>(p.dy + ((size_t)qq * p.lddy + grp * E) * sizeof(T));
      }
    };
    const long step = (long)U * ppb;
    long px = p0 + sub;
    if (px < p1) request(px);
    for (; px < p1; px += step) {
      float v[U][E], d[U][E];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long q = px + (long)u * ppb;
        cvt16<T>(rz[u], v[u]);
        if (MODE == 1) cvt16<T>(rd[u], d[u]);
        if (q >= p1) {
#pragma unroll
          for (int e = 0; e < E; ++e) { v[u][e] = MODE == 1 ? mean[e] : 0.f; d[u][e] = 0.f; }
        }
      }
      if (px + step < p1) request(px + step);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) { s0[e] += v[u][e]; s1[e] += v[u][e] * v[u][e]; }
        } else if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < E; ++e) s0[e] += v[u][e];
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float xh = (v[u][e] - mean[e]) * rstd[e];
            float du = d[u][e];
            if (p.act == UPA_ACT_SILU) du *= silu_grad<sizeof(T) == 2>(gam[e] * xh + bet[e]);
            s0[e] += du;
            s1[e] += du * xh;
          }
        }
      }
      if (++cnt == 16) {  // bound the f32 partial sums (64 pixels): fold into the f64 block accumulators
#pragma unroll
        for (int e = 0; e < E; ++e) {
          atomicAdd(&red[grp * E + e], (double)s0[e]);
          if (MODE != 2) atomicAdd(&red[p.c + grp * E + e], (double)s1[e]);
          s0[e] = s1[e] = 0.f;
        }
        cnt = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      atomicAdd(&red[grp * E + e], (double)s0[e]);
      if (MODE != 2) atomicAdd(&red[p.c + grp * E + e], (double)s1[e]);
    }
  }
  __syncthreads();
  // per-block partial sums, plain stores: [block][2][c] (a second kernel adds the blocks in a fixed order - with f64
  // atomics every extra block made the contention on the c addresses worse: 2048 blocks were the optimum at 1.2 TB/s)
  double* part = p.out0 + (size_t)blockIdx.x * 2 * p.c;
  for (int i = tid; i < p.c; i += 256) {
    part[i] = red[i];
    part[p.c + i] = MODE != 2 ? red[p.c + i] : 0.0;
  }
}

// Second stage: one workgroup per channel adds the block partials of both sums in a fixed order, then applies the
// epilogue of the operation (MODE 0: batch statistics + running update; 1: dgamma / dbeta + the sums kept as f64 for the
// dz pass; 2: bias gradient).
struct CombineParams {
  const double* part; int nblocks, c; double* fin;
  long npix; float momentum; float* mean; float* var; float* running_mean; float* running_var;  // MODE 0
  float* dgamma; float* dbeta; int accumulate;                                                   // MODE 1 / 2 (dbeta = out)
};

template <int MODE>
__global__ __launch_bounds__(256) void combine_finalize_kernel(const CombineParams p) {
  __shared__ double red[2][256];
  const int ch = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < p.nblocks; b += 256) {
    const double* q = p.part + (size_t)b * 2 * p.c;
    s0 += q[ch];
    if (MODE != 2) s1 += q[p.c + ch];
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  s0 = red[0][0]; s1 = red[1][0];
  if (MODE == 0) {
    const double m = s0 / (double)p.npix;
    double v = s1 / (double)p.npix - m * m;
    if (v < 0) v = 0;
    p.mean[ch] = (float)m;
    p.var[ch] = (float)v;
    if (p.running_mean) {  // nn.BatchNorm2d: running_var takes the unbiased estimate
      const double unb = p.npix > 1 ? v * (double)p.npix / (double)(p.npix - 1) : v;
      p.running_mean[ch] = (1.0f - p.momentum) * p.running_mean[ch] + p.momentum * (float)m;
      p.running_var[ch] = (1.0f - p.momentum) * p.running_var[ch] + p.momentum * (float)unb;
    }
  } else if (MODE == 1) {
    p.fin[ch] = s0; p.fin[p.c + ch] = s1;
    if (p.accumulate) { p.dbeta[ch] += (float)s0; p.dgamma[ch] += (float)s1; }
    else { p.dbeta[ch] = (float)s0; p.dgamma[ch] = (float)s1; }
  } else {
    p.dbeta[ch] = p.accumulate ? p.dbeta[ch] + (float)s0 : (float)s0;
  }
}

// The same second stage for the rows a convolution's statistics epilogue leaves (conv_big.hip TAIL 3, conv1x1.hip, conv_ws3.hip:
// [nrows][2][ld] f32 - sums and sums of squares of the stored values per pixel tile): fixed order, f64, MODE 0's epilogue.
__global__ __launch_bounds__(256) void combine_stats_rows_kernel(const float* rows, int nrows, int ld, const CombineParams p) {
  __shared__ double red[2][256];
  const int ch = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < nrows; b += 256) {
    const float* q = rows + (size_t)b * 2 * ld;
    s0 += (double)q[ch];
    s1 += (double)q[ld + ch];
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  s0 = red[0][0]; s1 = red[1][0];
  const double m = s0 / (double)p.npix;
  double v = s1 / (double)p.npix - m * m;
  if (v < 0) v = 0;
  p.mean[ch] = (float)m;
  p.var[ch] = (float)v;
  if (p.running_mean) {
    const double unb = p.npix > 1 ? v * (double)p.npix / (double)(p.npix - 1) : v;
    p.running_mean[ch] = (1.0f - p.momentum) * p.running_mean[ch] + p.momentum * (float)m;
    p.running_var[ch] = (1.0f - p.momentum) * p.running_var[ch] + p.momentum * (float)unb;
  }
}

struct BnApplyParams {
  const char* z; char* y; const char* aux;  // fwd: aux = residual; bwd: aux = dy
  long npix; int c, ldz, ldy, ldaux;
  const float* mean; const float* var; const float* gamma; const float* beta;
  const double* s0; const double* s1;  // bwd: sum(du), sum(du*xhat)
  float eps; int act;
};

// Thread = one 16-byte channel group walking a strided set of pixels: the per-channel constants (mean, rstd, gamma,
// beta, the two backward sums) are fetched once into registers, the loop is load - math - store.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyParams p) {
  constexpr int E = 16 / sizeof(T);
  const int cg = p.c / E;
  const int ppb = 256 / cg;
  const int tid = threadIdx.x;
  const int grp = tid % cg, sub = tid / cg;
  if (sub >= ppb) return;
  float mean[E], rstd[E], gam[E], bet[E], k0[E], k1[E];
  const float inv = 1.0f / (float)p.npix;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int ch = grp * E + e;
    mean[e] = p.mean[ch]; rstd[e] = 1.0f / sqrtf(p.var[ch] + p.eps); gam[e] = p.gamma[ch]; bet[e] = p.beta[ch];
    if (BWD) { k0[e] = (float)p.s0[ch]; k1[e] = (float)p.s1[ch]; }
  }
  const bool silu = p.act == UPA_ACT_SILU;
  for (long px = (long)blockIdx.x * ppb + sub; px < p.npix; px += (long)gridDim.x * ppb) {
    float v[E], o[E], a[E];
    load16<T>(p.z + ((size_t)px * p.ldz + grp * E) * sizeof(T), v);
    if (BWD || p.aux) load16<T>(p.aux + ((size_t)px * p.ldaux + grp * E) * sizeof(T), a);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float xh = (v[e] - mean[e]) * rstd[e];
      const float u = gam[e] * xh + bet[e];
      if (!BWD) {
        float r = u;
        if (silu) r = sizeof(T) == 2 ? u * sigmoid_t<true>(u) : u / (1.0f + expf(-u));
        if (p.aux) r += a[e];
        o[e] = r;
      } else {
        float du = a[e];
        if (silu) du *= silu_grad<sizeof(T) == 2>(u);
        o[e] = gam[e] * rstd[e] * (du - (k0[e] + xh * k1[e]) * inv);
      }
    }
    store16<T>(p.y + ((size_t)px * p.ldy + grp * E) * sizeof(T), o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient.  dW[co][ci][kh][kw] = sum_{n,oy,ox} dz[n,oy,ox,co] * x[n, oy*s+kh-p, ox*s+kw-p, ci].
// v_mfma_f32_16x16x4_f32: A[m][k] = dz[pixel k][co m], B[k][n] = x[pixel k shifted by the tap][ci n], lane = (m or n,
// k) - one f32 per lane straight out of an LDS image of the NHWC tile, rows padded to C+16 floats so the four pixel
// groups of a wave fall on disjoint banks.  A workgroup (2 x 2 waves, each MT x NT 16x16 tiles for every tap) owns a
// (co, ci) block and walks spatial tiles persistently; the taps x tiles accumulators live in registers and reach HBM as
// one f32 atomic per element per workgroup.
struct WgradParams {
  const char* x; const char* dz; float* dw;
  int N, H, W, Cin, ldx, OH, OW, Cout, lddz;
  int KS, stride, pad;
  int TH, TW, tilesX, tilesY, numTiles, IH, IW;
  float* partial;  // [co block][ci block][workgroup][tap][BCO][BCI] partial sums (plain stores; reduced by a second kernel)
};

template <typename T, int MT, int NT, int KK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  constexpr int BCO = 2 * MT * 16, BCI = 2 * NT * 16;
  constexpr int RZ = BCO + 16, RX = BCI + 16;  // LDS row strides (floats)
  constexpr int E = 16 / sizeof(T);
  float* zt = wsm;                               // [TH*TW][RZ]
  float* xt = wsm + (size_t)p.TH * p.TW * RZ;    // [IH*IW][RX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  f32x4 acc[KK * KK][MT][NT];
#pragma unroll
  for (int t = 0; t < KK * KK; ++t)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int npx = p.TH * p.TW;
  for (int tile = blockIdx.x; tile < p.numTiles; tile += gridDim.x) {
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    __syncthreads();  // previous tile fully consumed
    // ---- stage dz tile (zero outside the output map / past Cout)
    for (int i = tid; i < npx * (BCO / E); i += 256) {
      const int px = i / (BCO / E), g = i - px * (BCO / E);
      const int ty = px / p.TW, tx = px - ty * p.TW;
      const int oy = oy0 + ty, ox = ox0 + tx, co = co0 + g * E;
      float v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = 0.f;
      if (oy < p.OH && ox < p.OW && co < p.Cout)
        load16<T>(p.dz + ((((size_t)n * p.OH + oy) * p.OW + ox) * p.lddz + co) * sizeof(T), v);
#pragma unroll
      for (int e = 0; e < E; e += 4) *reinterpret_cast<f32x4*>(zt + px * RZ + g * E + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
    }
    // ---- stage x halo tile (zero padding / past Cin)
    const int nhx = p.IH * p.IW;
    for (int i = tid; i < nhx * (BCI / E); i += 256) {
      const int px = i / (BCI / E), g = i - px * (BCI / E);
      const int py = px / p.IW, pxx = px - py * p.IW;
      const int iy = iy0 + py, ix = ix0 + pxx, ci = ci0 + g * E;
      float v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = 0.f;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ci < p.Cin)
        load16<T>(p.x + ((((size_t)n * p.H + iy) * p.W + ix) * p.ldx + ci) * sizeof(T), v);
#pragma unroll
      for (int e = 0; e < E; e += 4) *reinterpret_cast<f32x4*>(xt + px * RX + g * E + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
    }
    __syncthreads();
    // ---- 4 pixels per MFMA step; lane group kq takes pixel q*4 + kq
    for (int q = 0; q < npx; q += 4) {
      const int px = q + kq;
      const int ty = px / p.TW, tx = px - ty * p.TW;
      float a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = zt[px * RZ + (wm * MT + i) * 16 + l16];
      const float* xb = xt + ((ty * p.stride) * p.IW + tx * p.stride) * RX + wn * NT * 16 + l16;
#pragma unroll
      for (int kh = 0; kh < KK; ++kh)
#pragma unroll
        for (int kw = 0; kw < KK; ++kw) {
          float b[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = xb[(kh * p.IW + kw) * RX + j * 16];
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[kh * KK + kw][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[kh * KK + kw][i][j], 0, 0, 0);
        }
    }
  }
  // ---- flush: D[row = co][col = ci]: lane holds rows 4*kq..+3 of column l16.  Every workgroup of a (co, ci) block
  // would hit the same dW addresses at the same moment with atomics (measured: ~270 us of serialised L2 atomics per
  // call); instead each workgroup stores its partial block and a small second kernel sums them in a fixed order.
  float* part = p.partial + ((((size_t)blockIdx.y * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * (KK * KK)) * (BCO * BCI);
#pragma unroll
  for (int t = 0; t < KK * KK; ++t)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = (wm * MT + i) * 16 + kq * 4 + r;
          const int cil = (wn * NT + j) * 16 + l16;
          part[((size_t)t * BCO + col) * BCI + cil] = acc[t][i][j][r];
        }
}

// ---- bf16 MFMA form of the weight gradient (3x3).  v_mfma_f32_16x16x32_bf16 wants 8 consecutive k (= pixels) per lane,
// but both operands are pixel-major in memory.  gfx950's transposing LDS read closes the gap: ds_read_b64_tr_b16 hands
// lane i of a 16-lane block element [row i/4 + 4j][column i%4] (j = 0..3) of the 16 x 4 block whose row r is the 8 bytes
// lane r pointed at (probed on hardware: tools/experiments/ds_read_tr.hip).  Pointing lane r at
// [pixel base + r/4][channels 4*(r%4)..+3] therefore leaves lane i with channel i of 4 consecutive pixels - half an MFMA
// operand - straight out of a plain [pixel][channel] LDS image, so the tile is staged with ordinary 16-byte copies, the
// tap shift is a pixel offset and stride 2 is a different per-lane address; nothing is transposed in software.
// Workgroup = 3 waves; wave w takes kernel row kh = w (3 taps) of the whole 64 x 64 (co, ci) block: 48 MFMAs per
// 32 ds_read_tr per 32-pixel step.  The next tile's global loads are in flight (in registers) while the current tile is
// multiplied.  Partial blocks go through the same two-stage reduction as the f32 kernel.
typedef short v4s16 __attribute__((ext_vector_type(4)));

// BCI = input channels per block.  Since round 4 only the BCI = 16 instantiation is launched, for 9 - 16 input channels: every
// other bf16 3x3 layer (and the pointwise layers, whose register-staged kernel is gone) takes the ring forms below.
template <int S, int BCI>
__global__ __launch_bounds__(192) void wgrad_bf16_k3_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char wsm_b[];
  constexpr int BCO = 64;
  constexpr int NT = BCI / 16;   // ci tiles
  constexpr int XG = BCI / 8;    // 16-byte chunks per input pixel
  constexpr int PZ = BCO * 2 + 32, PX = BCI * 2 + 32;       // LDS pixel pitch (bytes): +32 spreads 4 rows over the banks
  constexpr int TW = 16;
  constexpr int NTHR = 192;
  const int TH = p.TH, IH = p.IH, IW = p.IW;
  const int npx = TH * TW, nhx = IH * IW;
  char* zt = wsm_b;
  char* xt = wsm_b + (size_t)npx * PZ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int kh = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, r16 = lane & 15;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const int tilesPerImg = p.tilesX * p.tilesY;
  f32x4 acc[3][4][NT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging: chunk c (16 bytes = 8 channels of one pixel) of the dz tile then of the x halo; thread takes c = tid + q*192.
  // The decode of a chunk (operand, tile row / column, channel group, LDS address) does not depend on the tile: it is
  // done once (the per-tile form spent ~60 VALU instructions per chunk on divisions - more issue time than the MFMAs).
  constexpr int MAXQ = 16;  // (128 + 180) or (64 + 297) pixels * 8 chunks / 192 threads
  const int nzc = npx * 8;                 // dz chunks, then XG chunks per halo pixel
  const int nchunk = nzc + nhx * XG;
  int rel[MAXQ];      // element offset from the tile's dz / x origin pixel (channel block included)
  int ldsoff[MAXQ];   // byte offset in LDS, -1: no chunk
  int rc[MAXQ];       // (row << 16) | col inside the tile / halo, bit 31: x operand
  bool chok[MAXQ];    // channel group inside Cout / Cin
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int c = tid + q * NTHR;
    rel[q] = 0; ldsoff[q] = -1; rc[q] = 0; chok[q] = false;
    if (c < nzc) {
      const int px = c >> 3, g = c & 7;
      const int ty = px >> 4, tx = px & 15;
      rel[q] = (ty * p.OW + tx) * p.lddz + co0 + g * 8;
      ldsoff[q] = px * PZ + g * 16;
      rc[q] = (ty << 16) | tx;
      chok[q] = co0 + g * 8 < p.Cout;
    } else if (c < nchunk) {
      const int hp = (c - nzc) / XG, g = (c - nzc) - hp * XG;
      const int py = hp / IW, pxx = hp - py * IW;
      rel[q] = (py * p.W + pxx) * p.ldx + ci0 + g * 8;
      ldsoff[q] = npx * PZ + hp * PX + g * 16;
      rc[q] = (int)(0x80000000u | (py << 16) | pxx);
      chok[q] = ci0 + g * 8 < p.Cin;
    }
  }
  u32x4 stg[MAXQ];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const bf16_t* zb = (const bf16_t*)p.dz + ((size_t)(n * p.OH + oy0) * p.OW + ox0) * p.lddz;
    const bf16_t* xb = (const bf16_t*)p.x + ((long)(n * p.H + iy0) * p.W + ix0) * p.ldx;  // may point before the tensor: masked
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      const int r = (rc[q] >> 16) & 0x7fff, cc = rc[q] & 0xffff;
      if (rc[q] >= 0) {
        if (chok[q] && ldsoff[q] >= 0 && oy0 + r < p.OH && ox0 + cc < p.OW) v = *reinterpret_cast<const u32x4*>(zb + rel[q]);
      } else {
        const int iy = iy0 + r, ix = ix0 + cc;
        if (chok[q] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = *reinterpret_cast<const u32x4*>(xb + rel[q]);
      }
      stg[q] = v;
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
      if (ldsoff[q] >= 0) *reinterpret_cast<u32x4*>(wsm_b + ldsoff[q]) = stg[q];
  };
  int tile = blockIdx.x;
  if (tile < p.numTiles) fetch(tile);
  for (; tile < p.numTiles; tile += gridDim.x) {
    __syncthreads();  // everyone is done reading the previous tile
    commit();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < p.numTiles) fetch(next);  // in flight during the MFMAs below
    const int ksteps = npx >> 5;
    for (int ks = 0; ks < ksteps; ++ks) {
      // the two 4-pixel halves of this lane block's 8 pixels
      int zoff[2], xoff[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int px = ks * 32 + kg * 8 + h * 4 + (r16 >> 2);
        const int ty = px >> 4, tx = px & 15;
        zoff[h] = px * PZ + (r16 & 3) * 8;
        xoff[h] = ((ty * S + kh) * IW + tx * S) * PX + (r16 & 3) * 8;
      }
      u32x4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(zt + zoff[0] + i * 32));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(zt + zoff[1] + i * 32));
        a[i] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
      }
      // all three tap columns are fetched before the first MFMA: the reads of column kw+1 are in flight under the MFMAs
      // of column kw (with one shared fragment buffer the LDS latency sat in front of every 16 MFMAs)
      u32x4 b[3][NT];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s16*)(xt + xoff[0] + kw * PX + j * 32));
          const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4s16*)(xt + xoff[1] + kw * PX + j * 32));
          b[kw][j] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
        }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[kw][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a[i]), *reinterpret_cast<bf16x8*>(&b[kw][j]),
                                                                    acc[kw][i][j], 0, 0, 0);
    }
  }
  // flush this wave's three taps of the 64 x 64 block: D[row = co][col = ci], lane holds rows 4*kg..+3 of column r16
  float* part = p.partial + ((((size_t)blockIdx.y * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 9) * (BCO * BCI);
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[((size_t)(kh * 3 + kw) * BCO + i * 16 + kg * 4 + r) * BCI + j * 16 + r16] = acc[kw][i][j][r];
}

// ---- ring forms of the bf16 kernels (the pointwise one replaced a 128 x 128-block register-staged kernel of the same MFMA
// layout): the operands go global -> LDS by DMA (global_load_lds, 16 bytes per lane) into a
// ring of stages, RING - 1 of them in flight per workgroup, instead of through one register-staged tile.  The weight
// gradient is a streaming product (its MFMA floor is below its HBM floor on every yolov8s layer) and the register form
// kept ONE tile (41 - 64 KB) per workgroup in flight: 1.7 TB/s on the 160x160 layers, a tile every ~3 us for 0.4 us of
// MFMA (tools/experiments/r04_p2.sh: without the refetch the calls ran 20 - 45 % shorter, without the MFMAs 10 %).
// A DMA image is lane-linear (1 KB per wave instruction), so rows cannot be padded; the 32-byte units (16 channels) of a
// pixel row are XORed with pixel bits instead, which keeps the 8 pixels x 32 bytes of each 32-lane ds_read_b64_tr_b16
// group on 64 different banks.  Pixels / channels outside the tensors are fetched from a zero page.
UPA_STAMP_DEFINE(train)
#ifdef UPA_STAMP
#define WG_T(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#define WG_ACC(a, t0_, t1_) do { WG_T(t1_); a += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define WG_T(v) do {} while (0)
#define WG_ACC(a, t0_, t1_) do {} while (0)
#endif
typedef __attribute__((address_space(1))) const void* wg_gptr_t;
typedef __attribute__((address_space(3))) void* wg_lptr_t;
__device__ __attribute__((aligned(16))) unsigned g_wg_zero16[4] = {0u, 0u, 0u, 0u};

// The DMA is issued from inline asm: through the builtin the compiler knows that it writes LDS and puts s_waitcnt vmcnt(0)
// in front of the next LDS read - with stages in flight behind the one being read that wait is the whole pipeline.  Issued
// this way the loads are invisible to its counters; the counted waits below and the barrier order them against the reads.
__device__ __forceinline__ void wg_dma16(const char* src, const char* lds_dst) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory");
}

template <int N>
__device__ __forceinline__ void wg_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int R1_NPX = 64;                   // pixels per stage
constexpr int R1_IMG = R1_NPX * 256;         // one operand image: [px][128 channels], 16 KB
constexpr int R1_STAGE = 2 * R1_IMG;         // dz image, x image
constexpr int R1_RING = 4;                   // 128 KB of LDS, 96 KB in flight
constexpr int R1_NW = 8;                       // waves: 4 quadrants x 2 pixel halves of a stage
constexpr int R1_SLOTS = R1_STAGE / 1024 / R1_NW;  // DMA instructions per wave and stage (4)

// Pointwise (1x1, stride 1): 128 x 128 (co, ci) block, 8 waves = 4 quadrants of 4 x 4 MFMA tiles x the two 32-pixel halves of a
// 64-pixel stage (each half keeps its own accumulators and leaves as its own partial block: 2 * gridDim.x slices).  With 4
// waves (one per SIMD, 8 DMA instructions + 32 MFMAs per wave and stage) issuing the DMA took longer than the MFMAs.
__global__ __launch_bounds__(R1_NW * 64) void wgrad_k1_ring_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(1024))) char wsm_r1[];
  constexpr int BCO = 128, BCI = 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) & 1, wn = wave & 1, kh2 = wave >> 2;
  const int kg = lane >> 4, r16 = lane & 15;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const long P = (long)p.N * p.H * p.W;
  const long ntiles = (P + R1_NPX - 1) / R1_NPX;
  const int G = gridDim.x;
  const int nT = blockIdx.x < ntiles ? (int)((ntiles - 1 - blockIdx.x) / G) + 1 : 0;
  // DMA slot q of this wave = chunks ((q * 8 + wave) * 64 + lane) of the stage: operand q >> 1, pixel (q & 1) * 32 + wave * 4 + kg,
  // physical 16-byte chunk lane & 15 holding logical chunk (unit ^ (pixel & 7)) * 2 + half
  const int pxb = wave * 4 + kg;
  const int cl = ((((lane & 15) >> 1) ^ (pxb & 7)) << 1) | (lane & 1);
  const bool zok = co0 + cl * 8 < p.Cout, xok = ci0 + cl * 8 < p.Cin;
  const char* const zero = reinterpret_cast<const char*>(g_wg_zero16);
  const char* const zsrc = p.dz + ((size_t)pxb * p.lddz + co0 + cl * 8) * 2;
  const char* const xsrc = p.x + ((size_t)pxb * p.ldx + ci0 + cl * 8) * 2;
  auto stage = [&](int i) __attribute__((always_inline)) {   // tile i of this workgroup -> ring slot i % RING (past the end: zeros)
    const long p0 = ((long)blockIdx.x + (long)i * G) * R1_NPX;
    char* const dst = wsm_r1 + (size_t)(i % R1_RING) * R1_STAGE + wave * 1024;
#pragma unroll
    for (int q = 0; q < R1_SLOTS; ++q) {
      const long pp = p0 + (q & 1) * 32 + pxb;
      const char* src = zero;
      if (i < nT && pp < P) {
        if (q < 2) { if (zok) src = zsrc + (size_t)(p0 + (q & 1) * 32) * p.lddz * 2; }
        else       { if (xok) src = xsrc + (size_t)(p0 + (q & 1) * 32) * p.ldx * 2; }
      }
      wg_dma16(src, dst + q * (R1_NW * 1024));
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses inside an operand image: the lane's pixel of a 32-pixel step is h * 16 + kg * 4 + (r16 >> 2) (the K order
  // is free as long as both operands use it; this one puts 8 consecutive pixels in each 32-lane group)
  const int sw = (kg & 1) * 4 + (r16 >> 2);
  int offa[4], offb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    offa[i] = (kh2 * 32 + kg * 4 + (r16 >> 2)) * 256 + (((wm * 4 + i) ^ sw) << 5) + (r16 & 3) * 8;
    offb[i] = (kh2 * 32 + kg * 4 + (r16 >> 2)) * 256 + (((wn * 4 + i) ^ sw) << 5) + (r16 & 3) * 8 + R1_IMG;
  }
#pragma unroll
  for (int i = 0; i < R1_RING - 1; ++i) stage(i);
  for (int i = 0; i < nT; ++i) {
    wg_wait_vm<R1_SLOTS * (R1_RING - 2)>();   // this thread's share of stage i has landed ...
    __builtin_amdgcn_s_barrier();             // ... and everyone's; everyone is also done reading stage i - 1
    const char* const img = wsm_r1 + (size_t)(i % R1_RING) * R1_STAGE;
    u32x4 a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + offa[t]));
      const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + offa[t] + 16 * 256));
      a[t] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + offb[t]));
      const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + offb[t] + 16 * 256));
      b[t] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
    }
    stage(i + R1_RING - 1);                   // into the slot stage i - 1 used; its address arithmetic runs under the LDS latency
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a[ii]), *reinterpret_cast<bf16x8*>(&b[j]),
                                                             acc[ii][j], 0, 0, 0);
  }
  wg_wait_vm<0>();  // the zero stages issued past the last tile
  float* part = p.partial + (((size_t)blockIdx.y * gridDim.z + blockIdx.z) * (2 * gridDim.x) + blockIdx.x * 2 + kh2) * (BCO * BCI);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[(size_t)((wm * 4 + i) * 16 + kg * 4 + r) * BCI + (wn * 4 + j) * 16 + r16] = acc[i][j][r];
}

// 3x3 (stride S): 64 x 64 (co, ci) block, 12 waves = 3 kernel rows x 4 ci tiles (each: 3 taps x 4 co tiles = 12 MFMAs per
// 14 ds_read_tr per 32-pixel step; three waves per SIMD cover each other's LDS latency and DMA address arithmetic - at
// one wave per SIMD issuing a stage's DMA took a third to two thirds of the tile, tools/experiments/r04_t2.py).
// Stage = one TH x 16 output tile: the dz image [TH*16 px][64 co] then the x halo image [IH*IW px][64 ci], 128-byte rows
// whose four 32-byte units are XORed with (position >> 1) & 3.  At stride 2 the halo columns are stored de-interleaved
// (even columns, then odd) so that the 8 pixels of a 32-lane read group are consecutive positions again.
// Narrow inputs (the 3-channel stem, one 16-byte group per pixel): BCI = 16, 3 waves = the 3 kernel rows, the halo image is
// [IH*IW px][16 bytes]; MFMA columns 8 - 15 re-read columns 0 - 7 and are never stored.
template <int S, int BCI> struct R3Geo {
  static constexpr int NW = BCI == 64 ? 12 : 3;
  static constexpr int XCH = BCI == 64 ? 8 : 1;                  // 16-byte chunks per halo pixel
  static constexpr int TH = S == 1 ? 8 : 4, TW = 16;
  static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  static constexpr int NPX = TH * TW, NHX = IH * IW;
  static constexpr int NZC = NPX * 8, NCH = NZC + NHX * XCH;      // 16-byte chunks: dz, then x
  static constexpr int NSLOT = (NCH + 63) / 64;                  // DMA instructions per stage, dealt round-robin over the waves
  static constexpr int SPW = (NSLOT + NW - 1) / NW;        // ... per wave: SPW for waves < NFULL, SPW - 1 for the rest
  static constexpr int NFULL = NSLOT - (SPW - 1) * NW;
  static constexpr int STAGE = NSLOT * 1024;
  static constexpr int RING = (160 * 1024) / STAGE < 4 ? (160 * 1024) / STAGE : 4;
  static constexpr int NEV = (IW + 1) / 2;                       // even halo columns (stride 2)
  static_assert(RING >= 3, "ring of at least 3 stages");
  static_assert(NZC % 64 == 0, "a DMA instruction is all dz or all x");
  static_assert(SPW * (RING - 1) < 64, "vmcnt is a 6-bit counter");
};

template <int S, int BCI>
__global__ __launch_bounds__((R3Geo<S, BCI>::NW * 64)) void wgrad_k3_ring_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(1024))) char wsm_r3[];
  using G = R3Geo<S, BCI>;
  constexpr int BCO = 64, NW = G::NW;
  constexpr int KS = G::NPX / 32;       // 32-pixel steps per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = NW == 12 ? wave % 3 : wave, cq = NW == 12 ? wave / 3 : 0;
  const int kg = lane >> 4, r16 = lane & 15;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int Gx = gridDim.x;
  const int nT = (int)blockIdx.x < p.numTiles ? (p.numTiles - 1 - (int)blockIdx.x) / Gx + 1 : 0;
  // DMA slot q of this wave = chunks ((q * NW + wave) * 64 + lane), a whole slot is dz or x: decoded once into the byte offset
  // from the tile's dz / x origin pixel (channel block included) and the row / column inside the tile / halo (a lane with
  // nothing to fetch gets a row no tile reaches)
  int rel[G::SPW], rr[G::SPW], cc[G::SPW];
#pragma unroll
  for (int q = 0; q < G::SPW; ++q) {
    const int c = (q * NW + wave) * 64 + lane;
    const int cp = c & 7;
    rel[q] = 0; rr[q] = 0x40000000; cc[q] = 0;
    if (c < G::NZC) {
      const int px = c >> 3;
      const int cl = (((cp >> 1) ^ ((px >> 1) & 3)) << 1) | (cp & 1);
      const int ty = px >> 4, tx = px & 15;
      rel[q] = ((ty * p.OW + tx) * p.lddz + co0 + cl * 8) * 2;
      cc[q] = tx;
      if (co0 + cl * 8 < p.Cout) rr[q] = ty;
    } else if (c < G::NCH) {
      const int hp = BCI == 64 ? (c - G::NZC) >> 3 : c - G::NZC;
      const int cl = BCI == 64 ? (((cp >> 1) ^ ((hp >> 1) & 3)) << 1) | (cp & 1) : 0;
      const int py = hp / G::IW, pc = hp - py * G::IW;
      const int col = S == 1 ? pc : (pc < G::NEV ? 2 * pc : 2 * (pc - G::NEV) + 1);
      rel[q] = ((py * p.W + col) * p.ldx + ci0 + cl * 8) * 2;
      cc[q] = col;
      if (ci0 + cl * 8 < p.Cin) rr[q] = py;
    }
  }
  const char* const zero = reinterpret_cast<const char*>(g_wg_zero16);
  auto stage = [&](int i) __attribute__((always_inline)) {   // tile i of this workgroup -> ring slot i % RING (past the end: zeros)
    const int tile = (int)blockIdx.x + i * Gx;
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * G::TH, ox0 = txi * G::TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const char* const zb = p.dz + ((size_t)(n * p.OH + oy0) * p.OW + ox0) * p.lddz * 2;
    const char* const xb = p.x + ((long)(n * p.H + iy0) * p.W + ix0) * p.ldx * 2;  // may point before the tensor: masked
    const int zrl = i < nT ? p.OH : 0, xrl = i < nT ? p.H : 0;
    char* const sdst = wsm_r3 + (size_t)(i % G::RING) * G::STAGE + wave * 1024;
#pragma unroll
    for (int q = 0; q < G::SPW; ++q) {
      if (q == G::SPW - 1 && wave >= G::NFULL) break;   // wave-uniform
      const bool isx = q * NW + wave >= G::NZC / 64;  // wave-uniform
      const int rb = isx ? iy0 : oy0, rl = isx ? xrl : zrl, cb = isx ? ix0 : ox0, cl = isx ? p.W : p.OW;
      const char* const base = isx ? xb : zb;
      const bool ok = (unsigned)(rr[q] + rb) < (unsigned)rl && (unsigned)(cc[q] + cb) < (unsigned)cl;
      wg_dma16(ok ? base + rel[q] : zero, sdst + q * (NW * 1024));
    }
  };
  f32x4 acc[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the lane's pixel of a 32-pixel step: row ks * 2 + h, column tx = kg * 4 + (r16 >> 2) (8 consecutive columns per 32-lane group)
  const int tx = kg * 4 + (r16 >> 2);
  const int sub = (r16 & 3) * 8;
  int offa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) offa[i] = tx * 128 + ((i ^ ((tx >> 1) & 3)) << 5) + sub;
#pragma unroll
  for (int i = 0; i < G::RING - 1; ++i) stage(i);
#ifdef UPA_STAMP
  unsigned long long tq0 = 0, tq1 = 0, tStart = 0, aW = 0, aB = 0, aS = 0, aC = 0;
  WG_T(tStart); tq0 = tStart;
#endif
  for (int i = 0; i < nT; ++i) {
    if (wave < G::NFULL) wg_wait_vm<G::SPW * (G::RING - 2)>();
    else wg_wait_vm<(G::SPW - 1) * (G::RING - 2)>();
    WG_ACC(aW, tq0, tq1);
    __builtin_amdgcn_s_barrier();
    WG_ACC(aB, tq0, tq1);
    stage(i + G::RING - 1);   // into the ring slot stage i - 1 used (issued after the tile's products, or under the first step's LDS reads, instead: no faster)
    WG_ACC(aS, tq0, tq1);
    const char* const zt = wsm_r3 + (size_t)(i % G::RING) * G::STAGE;
    const char* const xt = zt + G::NZC * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 a[4], b[3];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(zt + offa[t] + (ks * 32) * 128));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(zt + offa[t] + (ks * 32 + 16) * 128));
        a[t] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int hoff[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = ((ks * 2 + h) * S + kh) * G::IW;
          const int hp = S == 1 ? row + tx + kw : row + ((kw & 1) ? G::NEV + tx : tx + (kw >> 1));
          hoff[h] = BCI == 64 ? hp * 128 + sub + ((cq ^ ((hp >> 1) & 3)) << 5) : hp * 16 + (r16 & 1) * 8;
        }
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(xt + hoff[0]));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(xt + hoff[1]));
        b[kw] = u32x4{((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]};
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
          acc[kw][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a[ii]), *reinterpret_cast<bf16x8*>(&b[kw]),
                                                                acc[kw][ii], 0, 0, 0);
    }
#ifdef UPA_STAMP
    asm volatile("s_nop 0" ::"v"(acc[2][3]));   // the tile's last MFMA has issued
    WG_ACC(aC, tq0, tq1);
#endif
  }
  wg_wait_vm<0>();
  float* part = p.partial + ((((size_t)blockIdx.y * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 9) * (BCO * BCI);
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[((size_t)(kh * 3 + kw) * BCO + i * 16 + kg * 4 + r) * BCI + cq * 16 + r16] = acc[kw][i][r];
#ifdef UPA_STAMP
  if (wave == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long tEnd; WG_T(tEnd);
    if (lane == 0) {   // synthetic boundaries: prologue | DMA waits | barriers | DMA issue | reads + MFMA | flush
      unsigned long long* o = g_upa_stamps + blockIdx.x * 16;
      const unsigned long long pro = tq0 - tStart - aW - aB - aS - aC;
      o[0] = tStart; o[1] = o[0] + pro; o[2] = o[1] + aW; o[3] = o[2] + aB; o[4] = o[3] + aS; o[5] = o[4] + aC; o[6] = tEnd;
    }
  }
  UPA_STAMP_HWID();
#endif
}

// dW[co][ci][t] (+)= sum over the workgroups of a block of their partial sums (fixed order: deterministic).
// Workgroup = one (block, co row): its kk2 * BCI partial values (kk2 segments of BCI floats per slice) as EG float4 groups x
// SG groups of slices; a thread adds its slices with 8 independent 16-byte loads in flight, the slice groups meet in LDS
// and the row leaves as one contiguous run of dW (the [t][ci] -> [ci][t] turn happens in LDS, not as 4-byte scatter).
// (The first form - 16 elements x 16 slice groups per 256 threads, two barriers per 16 elements - took 40 us for the
// 256 -> 256 3x3 layers and 27 us for a 5-slice pointwise layer: more than the products it summed.)
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* partial, int wgs, int bci, int BCO, int BCI, int kk2, int EG,
                                                            int SG, float* dw, int Cout, int Cin, int accumulate) {
  __shared__ float red[4096];
  const int R = kk2 * BCI;
  const int col = blockIdx.x % BCO;
  const int blk = blockIdx.x / BCO;   // by * bci + bz
  const int by = blk / bci, bz = blk - by * bci;
  const size_t per_wg = (size_t)kk2 * BCO * BCI;
  const int tid = threadIdx.x;
  const int eg = tid % EG, sg = tid / EG;
  if (sg < SG) {
    const int e0 = eg * 4;                 // element of the row: [t][cil]
    const int t = e0 / BCI, cil = e0 - t * BCI;
    const float* src = partial + (size_t)blk * wgs * per_wg + ((size_t)t * BCO + col) * BCI + cil;
    const int per = (wgs + SG - 1) / SG;
    const int w0 = sg * per, w1 = (w0 + per < wgs) ? w0 + per : wgs;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(w + u) * per_wg);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < w1; ++w) s += *reinterpret_cast<const f32x4*>(src + (size_t)w * per_wg);
    *reinterpret_cast<f32x4*>(&red[sg * R + e0]) = s;
  }
  __syncthreads();
  const int co = by * BCO + col;
  if (co >= Cout) return;
  for (int j = tid; j < R; j += blockDim.x) {
    const int cil = j / kk2, t = j - cil * kk2;
    const int ci = bz * BCI + cil;
    if (ci >= Cin) continue;
    float tot = 0.f;
    for (int q = 0; q < SG; ++q) tot += red[q * R + t * BCI + cil];
    float* d = dw + ((size_t)co * Cin + ci) * kk2 + t;
    *d = accumulate ? *d + tot : tot;
  }
}

static void launch_wgrad_reduce(const float* partial, int wgs, int bco, int bci, int BCO, int BCI, int kk2, float* dw, int Cout, int Cin,
                                int accumulate, hipStream_t s) {
  const int EG = kk2 * BCI / 4;
  int SG = 1024 / EG;
  if (SG > wgs) SG = wgs;
  if (SG > 16) SG = 16;
  int thr = (EG * SG + 63) / 64 * 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(bco * bci * BCO)), dim3(thr), 0, s, partial, wgs, bci, BCO, BCI, kk2, EG, SG, dw,
                     Cout, Cin, accumulate);
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void dilate2x_kernel(const char* src, int n, int oh, int ow, int c, int lds_, char* dst, int h, int w, int ldd) {
  // dst[n, y, x, :] = src[n, y/2, x/2, :] if y, x even and inside (oh, ow) else 0   (data gradient of stride-2 convs)
  constexpr int E = 16 / sizeof(T);
  const int cg = c / E;
  const long total = (long)n * h * w * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int g = (int)(t % cg); t /= cg;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h); t /= h;
    const int b = (int)t;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (!(y & 1) && !(x & 1) && (y >> 1) < oh && (x >> 1) < ow)
      v = *reinterpret_cast<const u32x4*>(src + ((((size_t)b * oh + (y >> 1)) * ow + (x >> 1)) * lds_ + g * E) * sizeof(T));
    *reinterpret_cast<u32x4*>(dst + ((((size_t)b * h + y) * w + x) * ldd + g * E) * sizeof(T)) = v;
  }
}

template <typename T>
__global__ void upsample2x_bwd_kernel(const char* dy, int n, int h, int w, int c, int lddy, char* dx, int lddx, int accumulate) {
  // dx[n, y, x, :] (+)= sum of the 2x2 block of dy (dy is (2h, 2w))
  constexpr int E = 16 / sizeof(T);
  const int cg = c / E;
  const long total = (long)n * h * w * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int g = (int)(t % cg); t /= cg;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h); t /= h;
    const int b = (int)t;
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    char* dst = dx + ((((size_t)b * h + y) * w + x) * lddx + g * E) * sizeof(T);
    if (accumulate) load16<T>(dst, s);
#pragma unroll
    for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
      for (int dx_ = 0; dx_ < 2; ++dx_) {
        float v[E];
        load16<T>(dy + ((((size_t)b * 2 * h + 2 * y + dy_) * 2 * w + 2 * x + dx_) * lddy + g * E) * sizeof(T), v);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += v[e];
      }
    store16<T>(dst, s);
  }
}

// MaxPool2d backward in two passes: (1) per output window the position (kh*k + kw) of its FIRST maximum in row-major
// scan order - the index torch's max_pool2d_with_indices keeps (k*k loads per output element); (2) every input pixel
// gathers dy from the windows that elected it (k*k one-byte index loads per element).  Deterministic, no atomics.
template <typename T>
__global__ void maxpool_argmax_kernel(const char* x, int n, int h, int w, int c, int ldx, int k, int stride, int pad, int oh, int ow,
                                      unsigned char* arg /* [n][oh][ow][c] */) {
  constexpr int E = 16 / sizeof(T);
  const int cg = c / E;
  const long total = (long)n * oh * ow * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int g = (int)(t % cg); t /= cg;
    const int ox = (int)(t % ow); t /= ow;
    const int oy = (int)(t % oh); t /= oh;
    const int b = (int)t;
    float best[E];
    unsigned char bi[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { best[e] = -INFINITY; bi[e] = 255; }
    for (int kh = 0; kh < k; ++kh) {
      const int yy = oy * stride - pad + kh;
      if (yy < 0 || yy >= h) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int xx = ox * stride - pad + kw;
        if (xx < 0 || xx >= w) continue;
        float v[E];
        load16<T>(x + ((((size_t)b * h + yy) * w + xx) * ldx + g * E) * sizeof(T), v);
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (v[e] > best[e] || bi[e] == 255) { best[e] = v[e]; bi[e] = (unsigned char)(kh * k + kw); }
      }
    }
    unsigned char* dst = arg + ((((size_t)b * oh + oy) * ow + ox) * c + g * E);
#pragma unroll
    for (int e = 0; e < E; ++e) dst[e] = bi[e];
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const unsigned char* arg, const char* dy, int n, int h, int w, int c, int lddy, int k, int stride,
                                   int pad, int oh, int ow, char* dx, int lddx, int accumulate) {
  constexpr int E = 16 / sizeof(T);
  const int cg = c / E;
  const long total = (long)n * h * w * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int g = (int)(t % cg); t /= cg;
    const int ix = (int)(t % w); t /= w;
    const int iy = (int)(t % h); t /= h;
    const int b = (int)t;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    char* dst = dx + ((((size_t)b * h + iy) * w + ix) * lddx + g * E) * sizeof(T);
    if (accumulate) load16<T>(dst, acc);
    for (int kh = 0; kh < k; ++kh) {
      const int ny = iy + pad - kh;  // oy * stride
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= oh) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int nx = ix + pad - kw;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= ow) continue;
        const size_t o = ((size_t)b * oh + oy) * ow + ox;
        const unsigned char* ai = arg + o * c + g * E;
        bool any = false;
        unsigned char a8[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { a8[e] = ai[e]; any |= a8[e] == kh * k + kw; }
        if (!any) continue;
        float d[E];
        load16<T>(dy + (o * lddy + g * E) * sizeof(T), d);
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (a8[e] == kh * k + kw) acc[e] += d[e];
      }
    }
    store16<T>(dst, acc);
  }
}

// k = 5, stride 1, pad 2 (SPPF's three pools, nn/modules/block.py:167-181), bf16: the two kernels above with the window unrolled -
// all 25 loads of a thread in flight, 32-bit indices, no divisions in the loop.  (The generic kernels walk the window with
// dependent branches: 35 + 31 us per pool for a 6.5 MB map; same scan order, same results.)
__global__ __launch_bounds__(256) void maxpool5_argmax_kernel(const char* x, int n, int h, int w, int cg, int ldx, unsigned char* arg) {
  const unsigned total = (unsigned)n * h * w * cg;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    unsigned t = idx;
    const int g = (int)(t % (unsigned)cg); t /= (unsigned)cg;
    const int ox = (int)(t % (unsigned)w); t /= (unsigned)w;
    const int oy = (int)(t % (unsigned)h);
    const int b = (int)(t / (unsigned)h);
    // clamped row / column byte offsets once (5 + 5 multiplies instead of 25 address computations)
    unsigned ro[5], co[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int yy = oy - 2 + q, xx = ox - 2 + q;
      const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy), xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      ro[q] = (unsigned)((b * h + yc) * w) * (unsigned)(ldx * 2);
      co[q] = (unsigned)xc * (unsigned)(ldx * 2) + (unsigned)g * 16u;
    }
    u32x4 v[25];
#pragma unroll
    for (int kk = 0; kk < 25; ++kk) v[kk] = *reinterpret_cast<const u32x4*>(x + ro[kk / 5] + co[kk % 5]);
    // strict > in scan order from -inf, the index starting at the first tap inside the image: what the generic kernel's
    // "first valid, then strictly greater" rule gives (32-bit index registers: byte-sized ones tripled the instruction count)
    float best[8];
    int bi[8];
    const int first = (oy < 2 ? 2 - oy : 0) * 5 + (ox < 2 ? 2 - ox : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = first; }
#pragma unroll
    for (int kk = 0; kk < 25; ++kk) {
      const int yy = oy - 2 + kk / 5, xx = ox - 2 + kk % 5;
      const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned wd = v[kk][e >> 1];
        const float f = __uint_as_float((e & 1) ? (wd & 0xFFFF0000u) : (wd << 16));
        const bool up = in && f > best[e];
        best[e] = up ? f : best[e];
        bi[e] = up ? kk : bi[e];
      }
    }
    const unsigned lo = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    const unsigned hi = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *reinterpret_cast<u32x2*>(arg + (size_t)idx * 8) = u32x2{lo, hi};   // [n][h][w][c] bytes, c = 8 cg
  }
}

__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const unsigned char* arg, const char* dy, int n, int h, int w, int cg, int lddy,
                                                            char* dx, int lddx, int accumulate) {
  const unsigned total = (unsigned)n * h * w * cg;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    unsigned t = idx;
    const int g = (int)(t % (unsigned)cg); t /= (unsigned)cg;
    const int ix = (int)(t % (unsigned)w); t /= (unsigned)w;
    const int iy = (int)(t % (unsigned)h);
    const int b = (int)(t / (unsigned)h);
    unsigned ra[5], ca[5], rd[5], cd[5];   // clamped row / column byte offsets into arg and dy
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int oy = iy + 2 - q, ox = ix + 2 - q;
      const int yc = oy < 0 ? 0 : (oy >= h ? h - 1 : oy), xc = ox < 0 ? 0 : (ox >= w ? w - 1 : ox);
      const unsigned row = (unsigned)((b * h + yc) * w);
      ra[q] = row * (unsigned)(cg * 8);
      ca[q] = (unsigned)xc * (unsigned)(cg * 8) + (unsigned)g * 8u;
      rd[q] = row * (unsigned)(lddy * 2);
      cd[q] = (unsigned)xc * (unsigned)(lddy * 2) + (unsigned)g * 16u;
    }
    u32x2 a8[25];
    u32x4 d[25];
#pragma unroll
    for (int kk = 0; kk < 25; ++kk) {   // the window (oy, ox) = (iy + 2 - kh, ix + 2 - kw) elected this pixel if its index is kk
      a8[kk] = *reinterpret_cast<const u32x2*>(arg + ra[kk / 5] + ca[kk % 5]);
      d[kk] = *reinterpret_cast<const u32x4*>(dy + rd[kk / 5] + cd[kk % 5]);
    }
    float acc[8];
    char* dst = dx + ((((size_t)b * h + iy) * w + ix) * lddx + g * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (accumulate) load16<bf16_t>(dst, acc);
#pragma unroll
    for (int kk = 0; kk < 25; ++kk) {
      const int oy = iy + 2 - kk / 5, ox = ix + 2 - kk % 5;
      if (oy < 0 || oy >= h || ox < 0 || ox >= w) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned ab = (a8[kk][e >> 2] >> (8 * (e & 3))) & 255u;
        const unsigned wd = d[kk][e >> 1];
        if (ab == (unsigned)kk) acc[e] += __uint_as_float((e & 1) ? (wd & 0xFFFF0000u) : (wd << 16));
      }
    }
    store16<bf16_t>(dst, acc);
  }
}

// ---- data gradient of a 3x3 stride-2 pad-1 convolution by output parity ------------------------------------------------
// dx[2i+py][2j+px] only receives the taps with kh = py+1 (mod 2), kw = px+1 (mod 2): four small stride-1 correlations
// over dz with 1, 2, 2 and 4 taps instead of a 9-tap correlation over a 4x zero-inserted dz.  Each phase is expressed as a
// 2x2 kernel V_p (zero taps where the parity has none) applied with pad 1 - out'[i'] = sum_a dz[i'+a-1] V[a], and the
// phase value of (i, j) is out'[i+1][j+1] - so the forward conv kernels run it unchanged; upa_interleave2x scatters the
// four phase maps into dx.   V_p[ci][co][a][b] = W[co][ci][kh][kw] with (py=0: a=0 -> kh=1; py=1: a=0 -> kh=2, a=1 -> kh=0).
__global__ void dgrad_s2_phase_weights_kernel(const float* w, int cout, int cin, float* v) {
  const long total = 4L * cin * cout * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int b = (int)(t & 1); t >>= 1;
    const int a = (int)(t & 1); t >>= 1;
    const int co = (int)(t % cout); t /= cout;
    const int ci = (int)(t % cin);
    const int ph = (int)(t / cin);
    const int py = ph >> 1, px = ph & 1;
    const int kh = py == 0 ? (a == 0 ? 1 : -1) : (a == 0 ? 2 : 0);
    const int kw = px == 0 ? (b == 0 ? 1 : -1) : (b == 0 ? 2 : 0);
    v[idx] = (kh < 0 || kw < 0) ? 0.f : w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
  }
}

template <typename T>
__global__ void interleave2x_kernel(const char* t0, const char* t1, const char* t2, const char* t3, int n, int oh1, int ow1, int c,
                                    int ldt, char* dx, int h, int w, int lddx, int accumulate) {
  constexpr int E = 16 / sizeof(T);
  const int cg = c / E;
  const long total = (long)n * h * w * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int g = (int)(t % cg); t /= cg;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h); t /= h;
    const int b = (int)t;
    const int ph = ((y & 1) << 1) | (x & 1);
    const char* src = ph == 0 ? t0 : (ph == 1 ? t1 : (ph == 2 ? t2 : t3));
    const int i = (y >> 1) + 1, j = (x >> 1) + 1;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = 0.f;
    if (i < oh1 && j < ow1) load16<T>(src + ((((size_t)b * oh1 + i) * ow1 + j) * ldt + g * E) * sizeof(T), v);
    char* dst = dx + ((((size_t)b * h + y) * w + x) * lddx + g * E) * sizeof(T);
    if (accumulate) {
      float o[E];
      load16<T>(dst, o);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] += o[e];
    }
    store16<T>(dst, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two stages, both in a fixed order (no atomics: the norm - and with it the clip coefficient of every weight - is the same
// bit pattern run after run): per-block partial sums, then one workgroup folds the partials.
__global__ void sumsq_kernel(const float* g, long n, double* partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)g[i] * (double)g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void sumsq_fold_kernel(const double* partial, int nblocks, double* out, int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = accumulate ? *out + red[0] : red[0];
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.SGD(nesterov) + ModelEMA.update over one flat parameter segment
// (engine/trainer.py:674-682, utils/torch_utils.py:632-646).  sumsq = squared gradient norm of ALL parameters.
// scaler (AMP GradScaler state on the device, or nullptr): [0] = the loss scale the gradients carry, see grad_scaler_update_kernel.
// GradScaler.unscale_ + clip + GradScaler.step (trainer.py:676-678): the norm is that of the UNSCALED gradients, the gradients are
// divided by the scale on the way into the update, and a step whose gradients hold an inf / NaN leaves parameters and momentum alone
// (optimizer.zero_grad() and the EMA update still happen, as in the reference's optimizer_step).
__global__ void sgd_nesterov_ema_kernel(float* p, float* g, float* buf, float* ema, long n, const double* sumsq, float max_norm,
                                        float lr, float momentum, float wd, int first_step, float ema_d, const float* ema_d_dev,
                                        int zero_grad, const float* scaler) {
  if (ema_d_dev) ema_d = *ema_d_dev;  // replayed graphs read the per-step decay from device memory
  const double ss = *sumsq;
  const float inv_scale = scaler ? 1.0f / scaler[0] : 1.0f;
  const bool skip = scaler != nullptr && !isfinite(ss);
  const float total = (float)sqrt(ss) * inv_scale;
  float coef = max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;
  coef *= inv_scale;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    float pn = pi;
    if (!skip) {
      float gi = g[i] * coef;
      if (wd != 0.f) gi = gi + wd * pi;
      float b = first_step ? gi : momentum * buf[i] + gi;
      buf[i] = b;
      gi = gi + momentum * b;
      pn = pi + (-lr) * gi;
      p[i] = pn;
    }
    if (ema) {
      float e = ema[i] * ema_d;
      e = e + (1.0f - ema_d) * pn;
      ema[i] = e;
    }
    if (zero_grad) g[i] = 0.f;
  }
}

// GradScaler.update() (torch/amp/grad_scaler.py): state = {scale, growth tracker, found_inf of the last step, the scale before this update}.  An overflowing step
// multiplies the scale by backoff and clears the tracker; `interval` clean steps in a row multiply it by growth.
__global__ void grad_scaler_update_kernel(float* state, const double* sumsq, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool inf = !isfinite(*sumsq);
  float scale = state[0];
  int tracker = (int)state[1];
  if (inf) { scale *= backoff; tracker = 0; }
  else if (++tracker == interval) { scale *= growth; tracker = 0; }
  state[3] = state[0];  // the scale the gradients of the step just taken carried (GradScaler.get_scale() after update() returns the NEW one)
  state[0] = scale; state[1] = (float)tracker; state[2] = inf ? 1.f : 0.f;
}

__global__ void ema_only_kernel(float* ema, const float* v, long n, float d, const float* d_dev) {
  if (d_dev) d = *d_dev;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float e = ema[i] * d;
    ema[i] = e + (1.0f - d) * v[i];
  }
}

template <typename TS, typename TD>
__global__ void cast_view_kernel(const char* src, int lds_, char* dst, int ldd, long npix, int c) {
  // 8 channels per thread (32 B of f32 / 16 B of bf16)
  const int cg = c / 8;
  const long total = npix * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long px = idx / cg;
    const int g = (int)(idx - px * cg);
    float v[8];
    const char* sp = src + ((size_t)px * lds_ + g * 8) * sizeof(TS);
    if constexpr (sizeof(TS) == 4) { load16<float>(sp, v); load16<float>(sp + 16, v + 4); }
    else load16<bf16_t>(sp, v);
    char* dp = dst + ((size_t)px * ldd + g * 8) * sizeof(TD);
    if constexpr (sizeof(TD) == 4) { store16<float>(dp, v); store16<float>(dp + 16, v + 4); }
    else store16<bf16_t>(dp, v);
  }
}

int grid_for(long total, int per_block, int cap);
int reduce_grid(long npix) {
  constexpr int ppb = 64;
  // measured on MI355X (tools/bench_bn.py, stats / whole backward in us): 512 blocks beat 2048 on every mid-size layer
  // (204800 px x 128 ch: 14.0 / 61 vs 22.8 / 79; 819200 x 32: 14.6 / 62 vs 22.7 / 74 - fewer per-block prologues, LDS
  // folds and partial rows); only the 3.3 M-pixel stem output prefers 1024 (35.5 vs 39.7)
  const int cap = npix > 1500000 ? 1024 : 512;
  return grid_for(npix, ppb, cap < 2048 ? cap : 2048);  // the workspace holds 2048 block partials
}

int grid_for(long total, int per_block = 256, int cap = 256 * 16) {
  long g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int upa_pack_conv_weight_dev(const float* w_oihw, int cout, int cin, int k, int dtype, int transpose_flip,
                                        void* out, void* stream) {
  UPA_CHECK_ARG(w_oihw && out && cout > 0 && cin > 0 && k >= 1 && k <= 7, "pack_conv_weight_dev: bad args");
  const int E = dtype == UPA_BF16 ? 8 : 4;
  const int lc_out = transpose_flip ? cin : cout, lc_in = transpose_flip ? cout : cin;
  const long total = (long)k * k * ((lc_in + 4 * E - 1) / (4 * E)) * ((lc_out + 15) / 16) * 64 * E;
  hipLaunchKernelGGL(pack_weight_dev_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, k, E,
                     transpose_flip, out, total);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_pack_conv_weights_batched(const UpaPackDesc* descs_dev, int n, void* stream) {
  UPA_CHECK_ARG(descs_dev && n > 0 && n <= 65535, "pack_conv_weights_batched: bad args");
  hipLaunchKernelGGL(pack_weight_batched_kernel, dim3(128, (unsigned)n), dim3(256), 0, (hipStream_t)stream, descs_dev);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

static int check_view(long npix, int c, int ld, int dtype, const char* what) {
  const int E = 16 / upa_elem_size(dtype);
  UPA_CHECK_ARG(npix > 0 && c > 0 && c % E == 0 && ld % E == 0 && c <= 256 * E, "%s: channels / stride must be multiples of %d (<= %d)",
                what, E, 256 * E);
  return UPA_OK;
}

extern "C" size_t upa_channel_reduce_workspace_bytes(int c) {
  return (size_t)(2048 + 1) * 2 * c * sizeof(double);  // per-block partials + the combined sums
}

static int run_channel_reduce(ReduceParams& r, int mode, int dtype, double* ws, hipStream_t s) {
  const int grid = reduce_grid(r.npix);
  r.out0 = ws; r.out1 = nullptr;
  const size_t lds = 2 * r.c * sizeof(double);
#define UPA_RED(MODE_)                                                                                               \
  do {                                                                                                               \
    if (dtype == UPA_BF16) hipLaunchKernelGGL((channel_reduce_kernel<bf16_t, MODE_>), dim3(grid), dim3(256), lds, s, r); \
    else hipLaunchKernelGGL((channel_reduce_kernel<float, MODE_>), dim3(grid), dim3(256), lds, s, r);               \
  } while (0)
  if (mode == 0) UPA_RED(0);
  else if (mode == 1) UPA_RED(1);
  else UPA_RED(2);
#undef UPA_RED
  return grid;  // block partials are in ws[0 .. grid*2c)
}

extern "C" int upa_bn_stats(const void* z, long npix, int c, int ldz, double* ws, int dtype, void* stream) {
  UPA_CHECK_ARG(z && ws, "bn_stats: null pointer");
  if (int rc = check_view(npix, c, ldz, dtype, "bn_stats")) return rc;
  ReduceParams p{};
  p.z = (const char*)z; p.npix = npix; p.c = c; p.ldz = ldz;
  run_channel_reduce(p, 0, dtype, ws, (hipStream_t)stream);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_bn_finalize(const double* ws, long npix, int c, float momentum, float* mean, float* var,
                               float* running_mean, float* running_var, void* stream) {
  UPA_CHECK_ARG(ws && mean && var && npix > 0, "bn_finalize: bad args");
  CombineParams q{};
  q.part = ws; q.nblocks = reduce_grid(npix); q.c = c; q.npix = npix; q.momentum = momentum; q.mean = mean; q.var = var;
  q.running_mean = running_mean; q.running_var = running_var;
  hipLaunchKernelGGL((combine_finalize_kernel<0>), dim3(c), dim3(256), 0, (hipStream_t)stream, q);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// (internal, conv.hip: upa_conv2d_bn_stats) batch statistics from the rows of a convolution's statistics epilogue
int upa_bn_finalize_rows(const float* rows, int nrows, int ld, long npix, int c, float momentum, float* mean, float* var,
                         float* running_mean, float* running_var, void* stream) {
  CombineParams q{};
  q.c = c; q.npix = npix; q.momentum = momentum; q.mean = mean; q.var = var; q.running_mean = running_mean; q.running_var = running_var;
  hipLaunchKernelGGL(combine_stats_rows_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, rows, nrows, ld, q);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_bn_act_fwd(const void* z, long npix, int c, int ldz, const float* mean, const float* var, const float* gamma,
                              const float* beta, float eps, int act, void* y, int ldy, const void* residual, int ldr, int dtype,
                              void* stream) {
  UPA_CHECK_ARG(z && y && mean && var && gamma && beta, "bn_act_fwd: null pointer");
  UPA_CHECK_ARG(act == UPA_ACT_SILU || act == UPA_ACT_NONE, "bn_act_fwd: activation must be SiLU or none");
  if (int rc = check_view(npix, c, ldz, dtype, "bn_act_fwd")) return rc;
  BnApplyParams p{};
  p.z = (const char*)z; p.y = (char*)y; p.aux = (const char*)residual; p.npix = npix; p.c = c; p.ldz = ldz; p.ldy = ldy; p.ldaux = ldr;
  p.mean = mean; p.var = var; p.gamma = gamma; p.beta = beta; p.eps = eps; p.act = act;
  const int E = 16 / upa_elem_size(dtype);
  const int grid = grid_for(npix, 256 / (c / E) * 4, 256 * 16);
  if (dtype == UPA_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((bn_apply_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_bn_act_bwd(const void* z, const void* dy, long npix, int c, int ldz, int lddy, const float* mean,
                              const float* var, const float* gamma, const float* beta, float eps, int act, void* dz, int lddz,
                              float* dgamma, float* dbeta, int accumulate, double* ws /* 2*c doubles */, int dtype, void* stream) {
  UPA_CHECK_ARG(z && dy && dz && mean && var && gamma && beta && dgamma && dbeta && ws, "bn_act_bwd: null pointer");
  UPA_CHECK_ARG(act == UPA_ACT_SILU || act == UPA_ACT_NONE, "bn_act_bwd: activation must be SiLU or none");
  if (int rc = check_view(npix, c, ldz, dtype, "bn_act_bwd")) return rc;
  hipStream_t s = (hipStream_t)stream;
  ReduceParams r{};
  r.z = (const char*)z; r.dy = (const char*)dy; r.npix = npix; r.c = c; r.ldz = ldz; r.lddy = lddy;
  r.mean = mean; r.var = var; r.gamma = gamma; r.beta = beta; r.eps = eps; r.act = act;
  const int nb = run_channel_reduce(r, 1, dtype, ws, s);
  double* fin = ws + (size_t)2048 * 2 * c;
  CombineParams q{};
  q.part = ws; q.nblocks = nb; q.c = c; q.fin = fin; q.dgamma = dgamma; q.dbeta = dbeta; q.accumulate = accumulate;
  hipLaunchKernelGGL((combine_finalize_kernel<1>), dim3(c), dim3(256), 0, s, q);
  BnApplyParams p{};
  p.z = (const char*)z; p.y = (char*)dz; p.aux = (const char*)dy; p.npix = npix; p.c = c; p.ldz = ldz; p.ldy = lddz; p.ldaux = lddy;
  p.mean = mean; p.var = var; p.gamma = gamma; p.beta = beta; p.s0 = fin; p.s1 = fin + c; p.eps = eps; p.act = act;
  const int E = 16 / upa_elem_size(dtype);
  const int grid2 = grid_for(npix, 256 / (c / E) * 4, 256 * 16);
  if (dtype == UPA_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, true>), dim3(grid2), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((bn_apply_kernel<float, true>), dim3(grid2), dim3(256), 0, s, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// The whole backward of a training-mode Conv (conv.py:177-186) in ONE call, for the common form (stride 1): upa_bn_act_bwd (dgamma,
// dbeta, dz), the weight gradient - on `side_stream` if given, ordered behind dz by an event, so that it overlaps the data-gradient chain
// the caller continues with - and the data gradient dx = conv(dz, W^T flipped) (+= with accumulate_dx; skipped when w_packed_t is NULL).
// The same launches as the separate calls; what it saves is host time: three library calls and an event record / wait from Python per
// layer (the eager step issues ~600 launches, its small-map phases are bound by the launch loop).
extern "C" int upa_conv2d_bias_act(const void*, int, int, int, int, int, const void*, const float*, void*, int, int, const void*, int, int,
                                   int, int, int, int, const upa_opts*, void*);
extern "C" int upa_conv2d_wgrad(const void* x, int n, int h, int w, int cin, int ldx, const void* dz, int cout, int lddz, float* dw_oihw,
                                int k, int stride, int pad, int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int upa_conv_bn_act_bwd(const void* x, int n, int h, int w, int cin, int ldx, const void* z, const void* dy, int cout, int ldz,
                                   int lddy, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act,
                                   void* dz, int lddz, float* dgamma, float* dbeta, double* ws, float* dw_oihw, void* wgrad_ws,
                                   size_t wgrad_ws_bytes, void* side_stream, const void* w_packed_t, void* dx, int lddx, int accumulate_dx,
                                   int k, int pad, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && z && dy && dz && dw_oihw && wgrad_ws, "conv_bn_act_bwd: null pointer");
  const int oh = h + 2 * pad - k + 1, ow = w + 2 * pad - k + 1;  // stride 1
  const long npix = (long)n * oh * ow;
  if (const int rc = upa_bn_act_bwd(z, dy, npix, cout, ldz, lddy, mean, var, gamma, beta, eps, act, dz, lddz, dgamma, dbeta, 1, ws, dtype,
                                    stream); rc != UPA_OK)
    return rc;
  hipStream_t main_s = (hipStream_t)stream, side = (hipStream_t)side_stream;
  if (side && side != main_s) {
    // one event per thread is enough: a wait enqueued on the side stream refers to the record that precedes it, re-recording later does
    // not move it
    // (... per DEVICE: an event belongs to the device that was current when it was created; a thread that trains on cuda:1 after cuda:0
    // must not record cuda:0's event on a cuda:1 stream)
    static thread_local hipEvent_t evs[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
      upa_set_error("conv_bn_act_bwd: hipGetDevice failed or device index >= 16");
      return UPA_ELAUNCH;
    }
    hipEvent_t& ev = evs[dev];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      upa_set_error("conv_bn_act_bwd: cannot create an event");
      return UPA_ELAUNCH;
    }
    if (hipEventRecord(ev, main_s) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) {
      upa_set_error("conv_bn_act_bwd: event record / wait failed");
      return UPA_ELAUNCH;
    }
  } else {
    side = main_s;
  }
  if (const int rc = upa_conv2d_wgrad(x, n, h, w, cin, ldx, dz, cout, lddz, dw_oihw, k, 1, pad, 1, dtype, wgrad_ws, wgrad_ws_bytes, side);
      rc != UPA_OK)
    return rc;
  if (!w_packed_t) return UPA_OK;
  UPA_CHECK_ARG(dx, "conv_bn_act_bwd: dx missing");
  return upa_conv2d_bias_act(dz, n, oh, ow, cout, lddz, w_packed_t, nullptr, dx, cin, lddx, accumulate_dx ? dx : nullptr,
                             accumulate_dx ? lddx : 0, k, 1, k - 1 - pad, UPA_ACT_NONE, dtype, opts, stream);
}

extern "C" int upa_channel_sum(const void* z, long npix, int c, int ldz, float* out, int accumulate, double* ws, int dtype,
                               void* stream) {
  UPA_CHECK_ARG(z && out && ws, "channel_sum: null pointer");
  if (int rc = check_view(npix, c, ldz, dtype, "channel_sum")) return rc;
  hipStream_t s = (hipStream_t)stream;
  ReduceParams r{};
  r.z = (const char*)z; r.npix = npix; r.c = c; r.ldz = ldz;
  const int nb = run_channel_reduce(r, 2, dtype, ws, s);
  CombineParams q{};
  q.part = ws; q.nblocks = nb; q.c = c; q.dbeta = out; q.accumulate = accumulate;
  hipLaunchKernelGGL((combine_finalize_kernel<2>), dim3(c), dim3(256), 0, s, q);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

static size_t wgrad_partial_bytes(int bco_n, int bci_n, int wgs, int BCO, int BCI, int k) {
  return (size_t)bco_n * bci_n * wgs * k * k * BCO * BCI * sizeof(float);
}

template <typename T, int MT, int NT>
static int launch_wgrad(WgradParams& p, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  constexpr int BCO = 2 * MT * 16, BCI = 2 * NT * 16;
  // tile: LDS = (TH*TW*(BCO+16) + IH*IW*(BCI+16)) * 4 bytes <= ~150 KB
  int TH = 8, TW = 16;
  auto lds_bytes = [&](int th, int tw) {
    const long ih = (th - 1) * p.stride + p.KS, iw = (tw - 1) * p.stride + p.KS;
    return (size_t)((long)th * tw * (BCO + 16) + ih * iw * (BCI + 16)) * 4;
  };
  while (lds_bytes(TH, TW) > 150 * 1024 && TH > 1) TH >>= 1;
  while (lds_bytes(TH, TW) > 150 * 1024 && TW > 4) TW >>= 1;
  UPA_CHECK_ARG(lds_bytes(TH, TW) <= 150 * 1024, "wgrad: tile does not fit LDS (k=%d s=%d)", p.KS, p.stride);
  p.TH = TH; p.TW = TW;
  p.tilesX = cdiv(p.OW, TW); p.tilesY = cdiv(p.OH, TH);
  p.numTiles = p.tilesX * p.tilesY * p.N;
  p.IH = (TH - 1) * p.stride + p.KS; p.IW = (TW - 1) * p.stride + p.KS;
  const int bco = cdiv(p.Cout, BCO), bci = cdiv(p.Cin, BCI);
  int wgs = 256 / (bco * bci);
  if (wgs < 1) wgs = 1;
  if (wgs > p.numTiles) wgs = p.numTiles;
  const size_t need = wgrad_partial_bytes(bco, bci, wgs, BCO, BCI, p.KS);
  UPA_CHECK_ARG(ws && ws_bytes >= need, "wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
  p.partial = (float*)ws;
  const size_t lds = lds_bytes(TH, TW);
  dim3 grid(wgs, bco, bci);
#define UPA_WG_LAUNCH(KK_)                                                                                        \
  do {                                                                                                            \
    auto kern = wgrad_kernel<T, MT, NT, KK_>;                                                                     \
    (void)upa_full_lds<wgrad_kernel<T, MT, NT, KK_>>();                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);                                                         \
  } while (0)
  if (p.KS == 1) UPA_WG_LAUNCH(1);
  else if (p.KS == 3) {
    if constexpr (MT * NT <= 4) UPA_WG_LAUNCH(3);
    else { upa_set_error("wgrad: 128x128 blocks are built for k = 1 only"); return UPA_EUNSUPPORTED; }
  } else { upa_set_error("wgrad: kernel size %d not built (1 and 3 are)", p.KS); return UPA_EUNSUPPORTED; }
#undef UPA_WG_LAUNCH
  launch_wgrad_reduce(p.partial, wgs, bco, bci, BCO, BCI, p.KS * p.KS, p.dw, p.Cout, p.Cin, accumulate, s);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

static bool wgrad_small(int cin, int cout) { return cin <= 32 || cout <= 32; }

static int launch_wgrad_bf16_k1(WgradParams& p, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  constexpr int BCO = 128, BCI = 128;
  const long P = (long)p.N * p.H * p.W;
  const int bco = cdiv(p.Cout, BCO), bci = cdiv(p.Cin, BCI);
  // 128 workgroups (half the CUs): in the yolov8s step, where these run on the side stream beside the main chain, 64 / 128 / 256
  // are within 0.1 ms of each other (tools/experiments/r04_z2.sh); alone 256 is faster on the 160x160 layers (47 vs 88 us)
#ifndef UPA_WGRAD_K1_BUDGET
#define UPA_WGRAD_K1_BUDGET 128
#endif
  constexpr int k1_budget = UPA_WGRAD_K1_BUDGET;
  long wgs = k1_budget / (bco * bci);
  if (wgs < 1) wgs = 1;
  const long nst = (P + R1_NPX - 1) / R1_NPX;
  if (wgs > nst) wgs = nst;
  const size_t need = wgrad_partial_bytes(bco, bci, 2 * (int)wgs, BCO, BCI, 1);   // two pixel halves per workgroup
  UPA_CHECK_ARG(ws && ws_bytes >= need, "wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
  p.partial = (float*)ws;
  auto kern = wgrad_k1_ring_kernel;
  (void)upa_full_lds<wgrad_k1_ring_kernel>();
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs, bco, bci), dim3(R1_NW * 64), (size_t)R1_RING * R1_STAGE, s, p);
  launch_wgrad_reduce(p.partial, 2 * (int)wgs, bco, bci, BCO, BCI, 1, p.dw, p.Cout, p.Cin, accumulate, s);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int BCI>
static int launch_wgrad_bf16_k3_t(WgradParams& p, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  constexpr int BCO = 64, PZ = BCO * 2 + 32, PX = BCI * 2 + 32;
  p.TW = 16;
  p.TH = p.stride == 1 ? 8 : 4;
  p.tilesX = cdiv(p.OW, p.TW); p.tilesY = cdiv(p.OH, p.TH);
  p.numTiles = p.tilesX * p.tilesY * p.N;
  p.IH = (p.TH - 1) * p.stride + 3; p.IW = (p.TW - 1) * p.stride + 3;
  const int bco = cdiv(p.Cout, BCO), bci = cdiv(p.Cin, BCI);
  // workgroups per launch (= slices of the pixel axis x channel blocks): 128 twelve-wave ring workgroups = half the CUs.  A ring
  // workgroup fills its CU (156 KB of LDS, 504 of 512 registers per lane), so in the training step - where these run on the side
  // stream beside the main chain - every CU a launch takes is one the main chain waits for, and each slice costs a 147 KB partial
  // block (PMC: 93 MB of HBM traffic per launch at 256 slices, 41 MB of it operands).  Same-box A/B of the yolov8s step, two
  // interleaved rounds on two boxes (tools/experiments/r04_e3.sh, compile-time variants): 256: 12.23 / 11.80 ms, 192: 12.00,
  // 160: 11.48, 128: 11.85 / 11.45, 96: 11.56, 64: 11.91; pointwise kernel at 64 / 128 / 192: 11.91 / 11.85 / 11.58 (second box: 128 best)
#ifndef UPA_WGRAD_K3_BUDGET
#define UPA_WGRAD_K3_BUDGET 128
#endif
  constexpr int wg_budget = UPA_WGRAD_K3_BUDGET;
  // ... except the stem (3.3 M output pixels): its weight gradient is the last kernel of the backward pass, nothing is
  // left to overlap it with: 768 three-wave workgroups of the narrow ring form (121 us; 130 at 512 and at 1024)
  constexpr long big_px = 3000000;
  const bool ring = BCI == 64 || p.Cin <= 8;   // the narrow ring form holds one 16-byte group (8 channels) per halo pixel
  int budget = wg_budget;
  if ((long)p.N * p.OH * p.OW >= big_px) budget = ring && BCI == 16 ? 768 : 512;  // the first layers run last in the backward pass
  int wgs = budget / (bco * bci);
  if (wgs < 1) wgs = 1;
  if (wgs > p.numTiles) wgs = p.numTiles;
  const size_t need = wgrad_partial_bytes(bco, bci, wgs, BCO, BCI, 3);
  UPA_CHECK_ARG(ws && ws_bytes >= need, "wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
  p.partial = (float*)ws;
  dim3 grid(wgs, bco, bci);
  if (ring) {
    if (p.stride == 1) {
      auto kern = wgrad_k3_ring_kernel<1, BCI>;
      (void)upa_full_lds<(wgrad_k3_ring_kernel<1, BCI>)>();
      using G1 = R3Geo<1, BCI>;
      hipLaunchKernelGGL(kern, grid, dim3(G1::NW * 64), (size_t)G1::RING * G1::STAGE, s, p);
    } else {
      auto kern = wgrad_k3_ring_kernel<2, BCI>;
      (void)upa_full_lds<(wgrad_k3_ring_kernel<2, BCI>)>();
      using G2 = R3Geo<2, BCI>;
      hipLaunchKernelGGL(kern, grid, dim3(G2::NW * 64), (size_t)G2::RING * G2::STAGE, s, p);
    }
  } else if constexpr (BCI == 16) {   // 9 - 16 input channels: the register-staged narrow form (two 16-byte groups per halo pixel)
    const size_t lds = (size_t)p.TH * p.TW * PZ + (size_t)p.IH * p.IW * PX;
    UPA_CHECK_ARG(p.TH * p.TW * 8 + p.IH * p.IW * (BCI / 8) <= 16 * 192, "wgrad: staging registers too few for this tile");
    if (p.stride == 1) {
      auto kern = wgrad_bf16_k3_kernel<1, 16>;
      (void)upa_full_lds<wgrad_bf16_k3_kernel<1, 16>>();
      hipLaunchKernelGGL(kern, grid, dim3(192), lds, s, p);
    } else {
      auto kern = wgrad_bf16_k3_kernel<2, 16>;
      (void)upa_full_lds<wgrad_bf16_k3_kernel<2, 16>>();
      hipLaunchKernelGGL(kern, grid, dim3(192), lds, s, p);
    }
  }
  launch_wgrad_reduce(p.partial, wgs, bco, bci, BCO, BCI, 9, p.dw, p.Cout, p.Cin, accumulate, s);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

static int launch_wgrad_bf16_k3(WgradParams& p, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  return p.Cin <= 16 ? launch_wgrad_bf16_k3_t<16>(p, accumulate, ws, ws_bytes, s)
                     : launch_wgrad_bf16_k3_t<64>(p, accumulate, ws, ws_bytes, s);
}

extern "C" size_t upa_conv2d_wgrad_workspace_bytes(int cin, int cout, int k) {
  // upper bound over the kernels the dispatcher may pick (f32 32/64/128 blocks with <= 256 workgroups, bf16 3x3 64-blocks
  // with <= 512, bf16 pointwise 128-blocks with <= 256)
  size_t best = 0;
  const int Bs[3] = {32, 64, 128};
  for (int b = 0; b < 3; ++b) {
    const int B = Bs[b];
    const int bco = cdiv(cout, B), bci = cdiv(cin, B);
    int wgs = (k == 3 && B == 64 ? 1024 : 256) / (bco * bci);
    if (wgs < 2) wgs = 2;   // the pointwise ring kernel leaves two partial blocks per workgroup
    const size_t n = wgrad_partial_bytes(bco, bci, wgs, B, B, k);
    if (n > best) best = n;
  }
  return best;
}

extern "C" int upa_conv2d_wgrad(const void* x, int n, int h, int w, int cin, int ldx, const void* dz, int cout, int lddz,
                                float* dw_oihw, int k, int stride, int pad, int accumulate, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
  UPA_CHECK_ARG(x && dz && dw_oihw, "wgrad: null pointer");
  UPA_CHECK_ARG(dtype == UPA_F32 || dtype == UPA_BF16, "wgrad: bad dtype");
  const int E = 16 / upa_elem_size(dtype);
  UPA_CHECK_ARG(ldx % E == 0 && lddz % E == 0 && cout % E == 0, "wgrad: strides / cout must be multiples of %d", E);
  UPA_CHECK_ARG(cin % E == 0 || cin < E, "wgrad: cin must be a multiple of %d (or a padded narrow input)", E);
  hipStream_t s = (hipStream_t)stream;
  WgradParams p{};
  p.x = (const char*)x; p.dz = (const char*)dz; p.dw = dw_oihw;
  p.N = n; p.H = h; p.W = w; p.Cin = cin; p.ldx = ldx; p.Cout = cout; p.lddz = lddz;
  p.OH = (h + 2 * pad - k) / stride + 1; p.OW = (w + 2 * pad - k) / stride + 1;
  p.KS = k; p.stride = stride; p.pad = pad;
  const bool small = wgrad_small(cin, cout);
  constexpr bool no_bf16_mfma = false;
  if (dtype == UPA_BF16 && k == 3 && cout >= 16 && !no_bf16_mfma && (cin % 8 == 0 || (cin < 8 && ldx >= 8)))
    return launch_wgrad_bf16_k3(p, accumulate, workspace, workspace_bytes, s);
  if (dtype == UPA_BF16 && k == 1 && stride == 1 && pad == 0 && cin >= 32 && cout >= 32 && !no_bf16_mfma && cin % 8 == 0)
    return launch_wgrad_bf16_k1(p, accumulate, workspace, workspace_bytes, s);
  if (k == 1 && cin >= 128 && cout >= 128) {  // pointwise = plain GEMM over the pixels: 128 x 128 blocks
    return dtype == UPA_BF16 ? launch_wgrad<bf16_t, 4, 4>(p, accumulate, workspace, workspace_bytes, s)
                             : launch_wgrad<float, 4, 4>(p, accumulate, workspace, workspace_bytes, s);
  }
  if (dtype == UPA_BF16)
    return small ? launch_wgrad<bf16_t, 1, 1>(p, accumulate, workspace, workspace_bytes, s)
                 : launch_wgrad<bf16_t, 2, 2>(p, accumulate, workspace, workspace_bytes, s);
  return small ? launch_wgrad<float, 1, 1>(p, accumulate, workspace, workspace_bytes, s)
               : launch_wgrad<float, 2, 2>(p, accumulate, workspace, workspace_bytes, s);
}

extern "C" int upa_dilate2x(const void* src, int n, int oh, int ow, int c, int lds_, void* dst, int h, int w, int ldd, int dtype,
                            void* stream) {
  UPA_CHECK_ARG(src && dst, "dilate2x: null pointer");
  if (int rc = check_view((long)n * h * w, c, lds_, dtype, "dilate2x")) return rc;
  const int E = 16 / upa_elem_size(dtype);
  const long total = (long)n * h * w * (c / E);
  if (dtype == UPA_BF16) hipLaunchKernelGGL((dilate2x_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                           (const char*)src, n, oh, ow, c, lds_, (char*)dst, h, w, ldd);
  else hipLaunchKernelGGL((dilate2x_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)src, n, oh,
                          ow, c, lds_, (char*)dst, h, w, ldd);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_upsample2x_bwd(const void* dy, int n, int h, int w, int c, int lddy, void* dx, int lddx, int accumulate, int dtype,
                                  void* stream) {
  UPA_CHECK_ARG(dy && dx, "upsample2x_bwd: null pointer");
  if (int rc = check_view((long)n * h * w, c, lddy, dtype, "upsample2x_bwd")) return rc;
  const int E = 16 / upa_elem_size(dtype);
  const long total = (long)n * h * w * (c / E);
  if (dtype == UPA_BF16) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                           (const char*)dy, n, h, w, c, lddy, (char*)dx, lddx, accumulate);
  else hipLaunchKernelGGL((upsample2x_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)dy, n,
                          h, w, c, lddy, (char*)dx, lddx, accumulate);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" size_t upa_maxpool2d_bwd_workspace_bytes(int n, int h, int w, int c, int k, int stride, int pad) {
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  return (size_t)n * oh * ow * c;
}

extern "C" int upa_maxpool2d_bwd(const void* x, const void* dy, int n, int h, int w, int c, int ldx, int lddy, int k, int stride,
                                 int pad, void* dx, int lddx, int accumulate, int dtype, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  UPA_CHECK_ARG(x && dy && dx && k >= 1 && k <= 15 && stride >= 1, "maxpool2d_bwd: bad args");
  if (int rc = check_view((long)n * h * w, c, ldx, dtype, "maxpool2d_bwd")) return rc;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  UPA_CHECK_ARG(workspace && workspace_bytes >= (size_t)n * oh * ow * c, "maxpool2d_bwd: workspace too small");
  const int E = 16 / upa_elem_size(dtype);
  const long tot_o = (long)n * oh * ow * (c / E), tot_i = (long)n * h * w * (c / E);
  unsigned char* arg = (unsigned char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UPA_BF16 && k == 5 && stride == 1 && pad == 2 &&
      (long)n * h * w * (ldx > lddy ? ldx : lddy) * 2 < (1L << 32) && (long)n * h * w * c < (1L << 32)) {   // 32-bit byte offsets
    hipLaunchKernelGGL(maxpool5_argmax_kernel, dim3(grid_for(tot_o)), dim3(256), 0, s, (const char*)x, n, h, w, c / 8, ldx, arg);
    hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(grid_for(tot_i)), dim3(256), 0, s, arg, (const char*)dy, n, h, w, c / 8, lddy, (char*)dx,
                       lddx, accumulate);
  } else if (dtype == UPA_BF16) {
    hipLaunchKernelGGL((maxpool_argmax_kernel<bf16_t>), dim3(grid_for(tot_o)), dim3(256), 0, s, (const char*)x, n, h, w, c, ldx, k, stride,
                       pad, oh, ow, arg);
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(grid_for(tot_i)), dim3(256), 0, s, arg, (const char*)dy, n, h, w, c, lddy, k,
                       stride, pad, oh, ow, (char*)dx, lddx, accumulate);
  } else {
    hipLaunchKernelGGL((maxpool_argmax_kernel<float>), dim3(grid_for(tot_o)), dim3(256), 0, s, (const char*)x, n, h, w, c, ldx, k, stride,
                       pad, oh, ow, arg);
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(grid_for(tot_i)), dim3(256), 0, s, arg, (const char*)dy, n, h, w, c, lddy, k,
                       stride, pad, oh, ow, (char*)dx, lddx, accumulate);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" size_t upa_sumsq_workspace_bytes(void) { return 1024 * sizeof(double); }

extern "C" int upa_sumsq(const float* g, long n, double* out, int accumulate, void* workspace, void* stream) {
  UPA_CHECK_ARG(g && out && workspace && n > 0, "sumsq: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(n, 1024, 1024);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, g, n, (double*)workspace);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, grid, out, accumulate);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_sgd_nesterov_ema(float* p, float* g, float* momentum_buf, float* ema, long n, const double* grad_sumsq,
                                    float max_norm, float lr, float momentum, float weight_decay, int first_step, float ema_d,
                                    const float* ema_d_dev, int zero_grad, void* stream) {
  UPA_CHECK_ARG(p && g && momentum_buf && grad_sumsq && n > 0, "sgd: bad args");
  hipLaunchKernelGGL(sgd_nesterov_ema_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf,
                     ema, n, grad_sumsq, max_norm, lr, momentum, weight_decay, first_step, ema_d, ema_d_dev, zero_grad,
                     (const float*)nullptr);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// The same step under an AMP GradScaler: g holds gradients of loss * scaler_state[0]; grad_sumsq is the squared norm of those SCALED
// gradients (inf / NaN = an overflowing step, which leaves p and the momentum buffer untouched).  scaler_state: 4 floats on the device.
extern "C" int upa_sgd_nesterov_ema_scaled(float* p, float* g, float* momentum_buf, float* ema, long n, const double* grad_sumsq,
                                           float max_norm, float lr, float momentum, float weight_decay, int first_step, float ema_d,
                                           const float* ema_d_dev, int zero_grad, const float* scaler_state, void* stream) {
  UPA_CHECK_ARG(p && g && momentum_buf && grad_sumsq && scaler_state && n > 0, "sgd_scaled: bad args");
  hipLaunchKernelGGL(sgd_nesterov_ema_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf,
                     ema, n, grad_sumsq, max_norm, lr, momentum, weight_decay, first_step, ema_d, ema_d_dev, zero_grad, scaler_state);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_grad_scaler_update(float* scaler_state, const double* grad_sumsq, float growth_factor, float backoff_factor,
                                      int growth_interval, void* stream) {
  UPA_CHECK_ARG(scaler_state && grad_sumsq && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
                "grad_scaler_update: bad args");
  hipLaunchKernelGGL(grad_scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler_state, grad_sumsq, growth_factor,
                     backoff_factor, growth_interval);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_ema_update(float* ema, const float* v, long n, float d, const float* d_dev, void* stream) {
  UPA_CHECK_ARG(ema && v && n > 0, "ema_update: bad args");
  hipLaunchKernelGGL(ema_only_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, ema, v, n, d, d_dev);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_cast_view(const void* src, int src_dtype, int lds_, void* dst, int dst_dtype, int ldd, long npix, int c,
                             void* stream) {
  UPA_CHECK_ARG(src && dst && npix > 0 && c % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "cast_view: bad args (c, strides multiples of 8)");
  const int grid = grid_for(npix * (c / 8));
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == UPA_BF16 && dst_dtype == UPA_F32)
    hipLaunchKernelGGL((cast_view_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const char*)src, lds_, (char*)dst, ldd, npix, c);
  else if (src_dtype == UPA_F32 && dst_dtype == UPA_BF16)
    hipLaunchKernelGGL((cast_view_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const char*)src, lds_, (char*)dst, ldd, npix, c);
  else if (src_dtype == UPA_F32 && dst_dtype == UPA_F32)
    hipLaunchKernelGGL((cast_view_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const char*)src, lds_, (char*)dst, ldd, npix, c);
  else
    hipLaunchKernelGGL((cast_view_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const char*)src, lds_, (char*)dst, ldd, npix, c);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_dgrad_s2_phase_weights(const float* w_oihw, int cout, int cin, float* v, void* stream) {
  UPA_CHECK_ARG(w_oihw && v && cout > 0 && cin > 0, "dgrad_s2_phase_weights: bad args");
  hipLaunchKernelGGL(dgrad_s2_phase_weights_kernel, dim3(grid_for(16L * cin * cout)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, v);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_interleave2x(const void* t00, const void* t01, const void* t10, const void* t11, int n, int oh1, int ow1, int c,
                                int ldt, void* dx, int h, int w, int lddx, int accumulate, int dtype, void* stream) {
  UPA_CHECK_ARG(t00 && t01 && t10 && t11 && dx, "interleave2x: null pointer");
  if (int rc = check_view((long)n * h * w, c, lddx, dtype, "interleave2x")) return rc;
  const int E = 16 / upa_elem_size(dtype);
  const long total = (long)n * h * w * (c / E);
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL((interleave2x_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)t00,
                       (const char*)t01, (const char*)t10, (const char*)t11, n, oh1, ow1, c, ldt, (char*)dx, h, w, lddx, accumulate);
  else
    hipLaunchKernelGGL((interleave2x_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)t00,
                       (const char*)t01, (const char*)t10, (const char*)t11, n, oh1, ow1, c, ldt, (char*)dx, h, w, lddx, accumulate);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
