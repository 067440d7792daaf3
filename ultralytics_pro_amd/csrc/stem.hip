// First-layer convolution: reads the model input in the reference's NCHW layout (f32 or bf16, <= 4 channels) and writes
// NHWC, so no separate layout-conversion pass touches the largest tensor of the network.
// Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 (k=3 s=2 for yolov8, k=6 s=2 p=2 for yolov5, k=3 s=1 v3).
//
// HBM-bound (K = cin*k*k <= 108 is too shallow for MFMA tiles): one thread = one output pixel x CO_T output channels,
// weights ([tap][ci][co] f32) broadcast from LDS, f32 FMA chain in (kh, kw, ci) order, coalesced reads along x,
// one contiguous CO_T*esize store per thread.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

struct StemParams {
  const void* x;
  const float* w;     // repacked [tap][ci][co16 group][16] f32 (device)
  const float* bias;
  char* y;
  int N, Cin, H, W, OH, OW, Cout, ldy, KS, stride, pad, act, x_bf16;
  int TH, TW, tilesX, tilesY, PR, PC, PCS;  // output tile, patch rows/cols, padded LDS row stride
  int wgs;  // stem_mfma_kernel: persistent workgroup cap
  int pool; // stem_mfma_kernel: 1 = y holds MaxPool2d(2, 2)(act(conv)) at (OH / 2, OW / 2)
  int no_xcd;  // 1 = tiles walked in slot order (upa_opts.no_xcd: A/B of the XCD-aware walk)
#ifdef UPA_ABLATE
  int ablate;  // debug build only (upa_opts.ablate_stem)
#endif
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
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && !UPA_ABL(p, 1)) {
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
        if (colOK && line + q * LSTEP < nlines && iy >= 0 && iy < p.H && !UPA_ABL(p, 1)) {
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
  for (int kh = 0; kh < (UPA_ABL(p, 2) ? 0 : p.KS); ++kh) {
    for (int kw = 0; kw < p.KS; ++kw) {
      for (int ci = 0; ci < p.Cin; ++ci) {
        const float xv = patch[(ci * p.PR + ty * p.stride + kh) * p.PCS + tx * p.stride + kw];
        const float* wt = wg + (size_t)((kh * p.KS + kw) * p.Cin + ci) * wstride;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = fmaf(xv, wt[c], acc[c]);
      }
    }
  }
  if (oy >= p.OH || ox >= p.OW || UPA_ABL(p, 4)) return;
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

// ---- MFMA form (bf16 compute mode, cin = 3).  The scalar-FMA kernel above spends cin*k*k*cout = 432 VALU FMAs per
// output pixel (36 us of pure FMA issue for the yolov8n stem) - more than the HBM time of the layer.  Here the im2col row
// of a pixel (K = 3*k*k, zero padded to 32*KSTEPS) is gathered from a bf16 LDS patch straight into the B operand of
// v_mfma_f32_16x16x32_bf16 (8 two-byte LDS reads per lane per k-step), the weights sit in registers as A fragments for the
// whole kernel, and a lane ends up with 4 consecutive output channels of one pixel: bias, SiLU, one 8-byte NHWC store.
// One workgroup = 8 x 64 output pixels x all couts (NT tiles of 16); a wave owns two output rows.
// All geometry is compile time (no integer divisions in the kernel); the patch origin is rounded down to 8 pixels so a
// bf16 NCHW input whose width is a multiple of 8 is staged by LDS-DMA in 16-byte chunks (no VGPR round trip).
typedef __attribute__((ext_vector_type(4))) short stem_s16x4;
typedef __attribute__((address_space(1))) const void* sgptr_t;
typedef __attribute__((address_space(3))) void* slptr_t;
__device__ __attribute__((aligned(16))) unsigned g_stem_zero16[4] = {0u, 0u, 0u, 0u};

template <int KS, int S>
struct StemGeo {
  static constexpr int TH = 8, TW = 64;
  static constexpr int PR = (TH - 1) * S + KS;          // patch rows
  static constexpr int PC = (TW - 1) * S + KS;          // patch columns actually used
  static constexpr int NCH = (PC + 7 + 7) / 8;          // 8-pixel chunks per line (origin shift <= 7)
  static constexpr int LS = NCH * 8;                    // line stride (elements)
  static constexpr int ITEMS = 3 * PR * NCH;            // 16-byte chunks in the patch
  static constexpr int ITEMS_PAD = (ITEMS + 255) / 256 * 256;
  static constexpr int KTOT = 3 * KS * KS;
  static constexpr int KSTEPS = (KTOT + 31) / 32;
};

// Stage the patch of output tile `tile` into `buf` (bf16 [3][PR][LS], column 0 = the tile's first input column rounded
// down to 8).  bf16 NCHW inputs with W % 8 == 0 go by LDS-DMA (asynchronous: caller waits vmcnt); the rest through VGPRs.
template <int KS, int S>
__device__ __forceinline__ void stem_stage(const StemParams& p, int tile, unsigned short* buf, int tid, int wave) {
  using G = StemGeo<KS, S>;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = tile / tilesPerImg;
  const int t2 = tile - n * tilesPerImg;
  const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
  const int iy0 = tyi * G::TH * S - p.pad, ix0 = txi * G::TW * S - p.pad;
  const int ixa = ix0 & ~7;                     // two's complement: floors negatives too
  const int plane = p.H * p.W;                  // 3 * plane < 2^31 (checked by the launcher)
  if (p.x_bf16 == 1 && (p.W & 7) == 0 && !UPA_ABL(p, 16)) {
    // lane = one 16-byte chunk (8 pixels of one line); chunks are entirely inside or outside the image
    const bf16_t* xb = (const bf16_t*)p.x + (size_t)n * 3 * plane;
#pragma unroll
    for (int it = 0; it < G::ITEMS_PAD / 256; ++it) {
      const int item = it * 256 + tid;
      const int line = item / G::NCH, ch = item - line * G::NCH;
      const int ci = line / G::PR, row = line - ci * G::PR;
      const int iy = iy0 + row, ix = ixa + ch * 8;
      const bool in = item < G::ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && !UPA_ABL(p, 1);
      const char* src = in ? reinterpret_cast<const char*>(xb + ci * plane + iy * p.W + ix)
                           : reinterpret_cast<const char*>(g_stem_zero16);
      __builtin_amdgcn_global_load_lds((sgptr_t)src, (slptr_t)((char*)buf + (it * 256 + wave * 64) * 16), 16, 0, 0);
    }
  } else if (p.x_bf16 == 2) {
    // uint8 HWC BGR frames (engine/predictor.py:151-173): BGR->RGB, HWC->CHW, /255 fused into the load
    const unsigned char* xu = (const unsigned char*)p.x;
    for (int i = tid; i < G::PR * G::LS; i += 256) {
      const int row = i / G::LS, col = i - row * G::LS;
      const int iy = iy0 + row, ix = ixa + col;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
        const unsigned char* px = xu + (((size_t)n * p.H + iy) * p.W + ix) * 3;
        v0 = (float)px[2] / 255.0f; v1 = (float)px[1] / 255.0f; v2 = (float)px[0] / 255.0f;
      }
      buf[(0 * G::PR + row) * G::LS + col] = f32_to_bf16(v0);
      buf[(1 * G::PR + row) * G::LS + col] = f32_to_bf16(v1);
      buf[(2 * G::PR + row) * G::LS + col] = f32_to_bf16(v2);
    }
  } else {
    // f32 input or odd widths: one thread = one column pair of one line
    const bool bf = p.x_bf16 != 0;
    const bf16_t* xb = (const bf16_t*)p.x;
    const float* xf = (const float*)p.x;
    unsigned* bufw = reinterpret_cast<unsigned*>(buf);
    constexpr int NP = G::LS / 2;
    for (int i = tid; i < 3 * G::PR * NP; i += 256) {
      const int line = i / NP, cp = i - line * NP;
      const int ci = line / G::PR, row = line - ci * G::PR;
      const int iy = iy0 + row, ix = ixa + 2 * cp;
      float lo = 0.f, hi = 0.f;
      if (iy >= 0 && iy < p.H) {
        const size_t rb = ((size_t)n * 3 + ci) * plane + (size_t)iy * p.W;
        if (ix >= 0 && ix < p.W) lo = bf ? bf16_to_f32(xb[rb + ix]) : xf[rb + ix];
        if (ix + 1 >= 0 && ix + 1 < p.W) hi = bf ? bf16_to_f32(xb[rb + ix + 1]) : xf[rb + ix + 1];
      }
      bufw[i] = pack_bf16x2(lo, hi);
    }
  }
}

// Persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; while tile i is computed from one LDS buffer the
// DMA of tile i+1 is already in flight into the other, so neither the weight fetch (once per workgroup) nor the memory
// round trip of a patch is exposed per tile.
// max of two packed bf16 pairs (as the floats they are; the result is one of the inputs, so re-packing is a shift)
__device__ __forceinline__ unsigned stem_max_bf16x2(unsigned a, unsigned b) {
  const float lo = fmaxf(__uint_as_float(a << 16), __uint_as_float(b << 16));
  const float hi = fmaxf(__uint_as_float(a & 0xFFFF0000u), __uint_as_float(b & 0xFFFF0000u));
  return (__float_as_uint(hi) & 0xFFFF0000u) | (__float_as_uint(lo) >> 16);
}

// POOL: the layer is followed by nn.MaxPool2d(2, 2, 0) (yolov3-tiny.yaml rows 0-1: 419 MB written and read back at bs 32) - the pool
// runs on the bf16-rounded activations in the epilogue and only the pooled tensor is written: a wave owns the row PAIR (2 wave,
// 2 wave + 1) of the tile, the second row is max-ed into the first in the wave's LDS strip, neighbouring pixels when the strip is
// read back for the 16-byte stores.  Bit-identical to conv -> store -> maxpool (the max of bf16 values is exact).
// G16 (k = 3, stride 1, pad 1: yolov3-tiny / darknet53 first layers; round 6): the im2col row as THREE 16-deep k-steps whose lane groups take one patch
// LINE (ci, kh) each and read four consecutive elements of it with ONE 4-byte-aligned ds_read2_b32 (as the fused kernels below; the form above
// costs eight ds_read_u16 + four packs per segment).  At stride 1 the first tap of pixel c sits at element c + 7 of the line - odd for even c - so a
// segment here is 16 pixels of the SAME parity (c = 32 h + 2 j + par): its lanes read consecutive dwords, and the two parities use two weight
// sets (taps at elements 1-3 of the four for even pixels, 0-2 for odd ones).  Lines are dealt so that the two lane groups sharing an LDS half-wave
// read 16-bank windows 16 banks apart: (0,0) | (0,2), (1,2) | (1,0); (2,0) | (2,2), (0,1) | (1,1); (2,1) alone (the other groups re-read it: broadcast).
template <int NT, int KS, int S, bool SILU, bool POOL = false, bool G16 = false>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemParams p) {
  static_assert(!G16 || (KS == 3 && S == 1), "the 16-deep gather form is built for k = 3, stride 1 (pad 1: checked by the launcher)");
  using G = StemGeo<KS, S>;
  constexpr int KSTEPS = G::KSTEPS;
  constexpr int PATCH_BYTES = G::ITEMS_PAD * 16;
  constexpr int OUT_ROW_BYTES = G::TW * NT * 32;      // one output row of the tile: 64 pixels x NT*16 bf16
  extern __shared__ __attribute__((aligned(16))) char stem_sm[];  // [2][PATCH_BYTES] + [4 waves][OUT_ROW_BYTES]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = p.tilesX * p.tilesY * p.N;
  const bool xcd = !p.no_xcd && ((gridDim.x & 7) == 0 || (int)gridDim.x >= ntiles);  // XCD-aware walk, as the fused kernel below
  int slot = blockIdx.x;
  if (slot >= ntiles) return;
  int tile = xcd ? upa_xcd_tile(slot, ntiles) : slot;
  stem_stage<KS, S>(p, tile, reinterpret_cast<unsigned short*>(stem_sm), tid, wave);
  // ---- weights -> A fragments (lane: cout row lane%16, k = ks*32 + (lane/16)*8 + j), kept for the whole kernel;
  // k = (kh*KS + kw)*3 + ci matches the packed [tap][ci][co] order
  const int kg = lane >> 4, l16 = lane & 15;
  const int wstride = (p.Cout + 15) / 16 * 16;
  u32x4 afrag[G16 ? 1 : NT][G16 ? 1 : KSTEPS];
  if constexpr (!G16) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = ks * 32 + kg * 8 + j;
          wv[j] = k < G::KTOT ? p.w[k * wstride + nt * 16 + l16] : 0.f;
        }
        afrag[nt][ks] = u32x4{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3]), pack_bf16x2(wv[4], wv[5]),
                              pack_bf16x2(wv[6], wv[7])};
      }
  }
  // G16: weight set q = first tap's element among the four (0: odd pixels, 1: even pixels), step m, and the byte offset of the lane's line
  u32x2 a16[G16 ? 2 : 1][G16 ? NT : 1][3];
  int g16[3] = {0, 0, 0};
  if constexpr (G16) {
    constexpr int LCI[3][4] = {{0, 0, 1, 1}, {2, 2, 0, 1}, {2, 2, 2, 2}}, LKH[3][4] = {{0, 2, 2, 0}, {0, 2, 1, 1}, {1, 1, 1, 1}};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int ci = LCI[m][kg], kh = LKH[m][kg];
      const bool live = m < 2 || kg == 0;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kw = e - q;
            wv[e] = (live && kw >= 0 && kw < 3) ? p.w[((kh * 3 + kw) * 3 + ci) * wstride + nt * 16 + l16] : 0.f;
          }
          a16[q][nt][m] = u32x2{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3])};
        }
      g16[m] = ((ci * G::PR + kh + (POOL ? 2 * wave : wave)) * G::LS) * 2 + l16 * 4;
    }
  }
  f32x4 bias4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[nt][r] = p.bias ? p.bias[nt * 16 + kg * 4 + r] : 0.f;
  // per-lane gather offsets (bytes, inside a patch buffer) of the im2col row for output row `wave`, segment 0
  const int shift = (-p.pad) & 7;  // tile origins are multiples of 64*S columns: the same for every tile
  int gat[KSTEPS][8];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ks * 32 + kg * 8 + j;
      int o = 0;
      if (k < G::KTOT) {
        const int tap = k / 3, ci = k - tap * 3;
        const int kh = tap / KS, kw = tap - kh * KS;
        o = (ci * G::PR + kh) * G::LS + kw;
      }
      gat[ks][j] = (o + shift + ((POOL ? 2 * wave : wave) * S) * G::LS + l16 * S) * 2;
    }
  char* ostage = stem_sm + 2 * PATCH_BYTES + wave * OUT_ROW_BYTES;
  const int tilesPerImg = p.tilesX * p.tilesY;
  constexpr int SEGS = G::TW / 16;
  for (int cur = 0;; cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // patch `cur` landed for every wave; every wave is done reading the other buffer
    slot += gridDim.x;
    const int next = slot < ntiles ? (xcd ? upa_xcd_tile(slot, ntiles) : slot) : ntiles;
    if (next < ntiles) stem_stage<KS, S>(p, next, reinterpret_cast<unsigned short*>(stem_sm + (cur ^ 1) * PATCH_BYTES), tid, wave);
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * G::TH, ox0 = txi * G::TW;
    const char* pb = stem_sm + cur * PATCH_BYTES;
    [[maybe_unused]] f32x4 vkeep[(POOL && G16) ? SEGS : 1][(POOL && G16) ? NT : 1];  // POOL + G16: SiLU'd row 2 wave, kept in f32 for the pool
#pragma unroll
    for (int rr = 0; rr < G::TH / 4; ++rr) {
      if UPA_ABL(p, 2) break;
      f32x4 acc[SEGS][NT];
      if constexpr (G16) {
        // segment sx = (h, par): pixels 32 h + 2 j + par (j = lane & 15); first tap at element c + 7 of the line (origin shift 7 for pad 1):
        // par 1 -> element 32 h + 2 j + 8 = dword 16 h + j + 4, taps at elements 0-2 (set 0); par 0 -> dword 16 h + j + 3, taps at 1-3 (set 1)
        u32x2 b[SEGS][3];
#pragma unroll
        for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const unsigned* src = reinterpret_cast<const unsigned*>(pb + g16[m] + (rr * (POOL ? 1 : 4) * G::LS) * 2 + (16 * (sx >> 1) + 3 + (sx & 1)) * 4);
            b[sx][m] = u32x2{src[0], src[1]};
          }
#pragma unroll
        for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[sx][nt] = bias4[nt];
#pragma unroll
            for (int m = 0; m < 3; ++m)
              acc[sx][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<stem_s16x4*>(&a16[1 - (sx & 1)][nt][m]),
                                                                     *reinterpret_cast<stem_s16x4*>(&b[sx][m]), acc[sx][nt], 0, 0, 0);
          }
      } else {
        u32x4 b[SEGS][KSTEPS];
#pragma unroll
        for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            unsigned e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              e[j] = *reinterpret_cast<const unsigned short*>(pb + gat[ks][j] + (rr * (POOL ? 1 : 4) * S * G::LS + sx * 16 * S) * 2);
            b[sx][ks] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
          }
#pragma unroll
        for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[sx][nt] = bias4[nt];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
              acc[sx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&afrag[nt][ks]),
                                                                    *reinterpret_cast<bf16x8*>(&b[sx][ks]), acc[sx][nt], 0, 0, 0);
          }
      }
      if constexpr (POOL && G16) {
        // the 2 x 2 pool entirely in registers: a lane holds pixels 2 j (segment (h, 0)) and 2 j + 1 (segment (h, 1)) of BOTH rows of the pair, so
        // the pooled pixel 16 h + j is the max of four of its own f32 values - taken before the bf16 rounding, which is monotone: bit-identical
        // to rounding first (-0 / +0 aside) - and leaves as one 8-byte store per lane; no LDS strip, no unpack / pack around the max
#pragma unroll
        for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = acc[sx][nt][r];
              if (!UPA_ABL(p, 4)) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
              vkeep[sx][nt][r] = rr == 0 ? t : fmaxf(vkeep[sx][nt][r], t);
            }
        if (rr == 0) continue;
        const int oyp = (oy0 >> 1) + wave, oxp0 = ox0 >> 1;
        if (oyp < (p.OH >> 1) && !UPA_ABL(p, 8)) {
          const unsigned rowpix = ((unsigned)n * (p.OH >> 1) + oyp) * (p.OW >> 1) + oxp0;
#pragma unroll
          for (int h = 0; h < SEGS / 2; ++h) {
            const int px = 16 * h + l16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(vkeep[2 * h][nt][r], vkeep[2 * h + 1][nt][r]);
              if (oxp0 + px < (p.OW >> 1))
                *reinterpret_cast<u32x2*>(p.y + ((size_t)(rowpix + px) * p.ldy + nt * 16 + kg * 4) * sizeof(bf16_t)) =
                    u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
          }
        }
        continue;
      }
      // epilogue: bias is in, SiLU, bf16; the row goes through a wave-private LDS strip so that the global stores are
      // whole contiguous 16-byte chunks (1 KiB per wave instruction) instead of 8-byte pieces of four lanes per pixel
#pragma unroll
      for (int sx = 0; sx < SEGS; ++sx)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = acc[sx][nt][r];
            if constexpr (SILU) if (!UPA_ABL(p, 4)) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
            v[r] = t;
          }
          u32x2 pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          const int pix = G16 ? 32 * (sx >> 1) + 2 * l16 + (sx & 1) : sx * 16 + l16;   // pixel of the tile row this lane holds
          u32x2* slot = reinterpret_cast<u32x2*>(ostage + pix * (NT * 32) + nt * 32 + kg * 8);
          if constexpr (POOL) {
            if (rr == 1) {  // the pair's second row: max with what this lane stored for the first
              const u32x2 up = *slot;
              pk = u32x2{stem_max_bf16x2(pk[0], up[0]), stem_max_bf16x2(pk[1], up[1])};
            }
          }
          *slot = pk;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (POOL) {
        if (rr == 0) continue;
        const int oyp = (oy0 >> 1) + wave, oxp0 = ox0 >> 1;
        if (oyp < (p.OH >> 1) && !UPA_ABL(p, 8)) {
          const unsigned rowpix = ((unsigned)n * (p.OH >> 1) + oyp) * (p.OW >> 1) + oxp0;
#pragma unroll
          for (int c = 0; c < NT; ++c) {  // 32 pooled pixels x NT * 2 chunks
            const int chunk = c * 64 + lane;
            const int px = chunk / (NT * 2), part = chunk - px * (NT * 2);
            const u32x4 a = *reinterpret_cast<const u32x4*>(ostage + (2 * px) * (NT * 32) + part * 16);
            const u32x4 b = *reinterpret_cast<const u32x4*>(ostage + (2 * px + 1) * (NT * 32) + part * 16);
            const u32x4 val = u32x4{stem_max_bf16x2(a[0], b[0]), stem_max_bf16x2(a[1], b[1]), stem_max_bf16x2(a[2], b[2]),
                                    stem_max_bf16x2(a[3], b[3])};
            if (oxp0 + px < (p.OW >> 1))
              *reinterpret_cast<u32x4*>(p.y + ((size_t)(rowpix + px) * p.ldy + part * 8) * sizeof(bf16_t)) = val;
          }
        }
        continue;
      }
      const int oy = oy0 + wave + rr * 4;
      if (oy < p.OH && !UPA_ABL(p, 8)) {
        const unsigned rowpix = ((unsigned)n * p.OH + oy) * p.OW + ox0;  // < 2^31 pixels per tensor
#pragma unroll
        for (int c = 0; c < NT * 2; ++c) {
          const int chunk = c * 64 + lane;
          const int px = chunk / (NT * 2), part = chunk - px * (NT * 2);
          const u32x4 val = *reinterpret_cast<const u32x4*>(ostage + chunk * 16);
          if (ox0 + px < p.OW)
            *reinterpret_cast<u32x4*>(p.y + ((size_t)(rowpix + px) * p.ldy + part * 8) * sizeof(bf16_t)) = val;
        }
      }
    }
    tile = next;
    if (tile >= ntiles) break;
  }
}

template <int NT, int KS, int S>
static void launch_stem_mfma(const StemParams& p, int n, hipStream_t st) {
  using G = StemGeo<KS, S>;
  StemParams q = p;
  q.TW = G::TW; q.TH = G::TH;
  q.tilesX = cdiv(p.OW, G::TW);
  q.tilesY = cdiv(p.OH, G::TH);
  const long ntiles = (long)q.tilesX * q.tilesY * n;
  const int wgs = p.wgs > 0 ? p.wgs : 1024;
  dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs));
  const size_t lds = (size_t)2 * G::ITEMS_PAD * 16 + 4 * G::TW * NT * 32;
  if constexpr (S == 1 && KS == 3) {
    if (p.pool && p.pad == 1) {  // (SiLU only: the launcher's caller checked)
      auto kern = stem_mfma_kernel<NT, KS, S, true, true, true>;
      (void)upa_full_lds<stem_mfma_kernel<NT, KS, S, true, true, true>>();
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
      return;
    }
    if (p.pool) {
      auto kern = stem_mfma_kernel<NT, KS, S, true, true>;
      (void)upa_full_lds<stem_mfma_kernel<NT, KS, S, true, true>>();
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
      return;
    }
    if (p.pad == 1 && p.act == UPA_ACT_SILU) {
      auto kern = stem_mfma_kernel<NT, KS, S, true, false, true>;
      (void)upa_full_lds<stem_mfma_kernel<NT, KS, S, true, false, true>>();
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
      return;
    }
  }
  if (p.act == UPA_ACT_SILU) {
    auto kern = stem_mfma_kernel<NT, KS, S, true>;
    (void)upa_full_lds<stem_mfma_kernel<NT, KS, S, true>>();
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
  } else {
    auto kern = stem_mfma_kernel<NT, KS, S, false>;
    (void)upa_full_lds<stem_mfma_kernel<NT, KS, S, false>>();
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
  }
}
template <int KS, int S>
static void launch_stem_mfma_nt(const StemParams& p, int n, int nt, hipStream_t st) {
  if (nt == 1) launch_stem_mfma<1, KS, S>(p, n, st);
  else if (nt == 2) launch_stem_mfma<2, KS, S>(p, n, st);
  else launch_stem_mfma<4, KS, S>(p, n, st);
}

// ---- Stem + second conv fused (yolov8n: Conv(3,16,3,2) -> Conv(16,32,3,2), both SiLU; yolov8.yaml rows 0-1).
// The stem output is the largest activation of the network (bs 32: 105 MB written and read back = more HBM time than
// either conv); here it only ever exists as a 17 x 33 pixel LDS tile.  Per 8 x 16 output tile of the second conv:
//   1. input patch (35 x 67 pixels x 3) by LDS-DMA, double buffered across tiles (persistent workgroup);
//   2. stem: im2col gather (16-deep k-steps, one ds_read2_b32 per lane group and patch line: see the kernel) + MFMAs per 16 stem pixels, SiLU,
//      bf16, written to an LDS tile laid out PLANAR [8-channel plane][row][column parity][column/2][16 B] (sf::SPLANE: the stride-2 reads of the
//      next stage step through consecutive 16-byte slots of one parity; conflict-free ds_read_b128 and ds_write_b64); stem pixels outside the
//      stem map are ZERO (the second conv's padding);
//   3. second conv as implicit GEMM from that tile: 16 input channels = half a 32-wide MFMA k-step, so k-steps pair two
//      taps (lane groups 0-1 take tap 2s, groups 2-3 tap 2s+1): 5 k-steps instead of 9; the A fragments are read
//      straight from the standard packed weights (only the address of a lane's 16 bytes changes);
//   4. bias, SiLU, bf16, v_permlane16_swap pairs the two 16-channel tiles into 16-byte stores.
struct StemFusedParams {
  const void* x; const float* w0; const float* b0; const char* w1; const float* b1; char* y;
  int N, H, W, H0, W0, OH, OW, ldy, x_bf16;
  int tilesX, tilesY;
  int no_xcd;
};

namespace sf {
constexpr int T1H = 8, T1W = 16;               // output tile of the second conv
constexpr int S0H = 2 * T1H + 1, S0W = 2 * T1W + 1;  // stem tile 17 x 33
constexpr int S0WH = (S0W + 1) / 2;            // 17 columns per parity
// stem tile in LDS, PLANAR: [8-channel plane (2)][row][column parity][column / 2][16 B]; the planes a multiple of 256 B apart, so the two lane groups
// that share a ds_read_b128 group (the hardware's groups are lanes {0-3, 12-15, 20-27}, ...: kg 0 with kg 1, kg 2 with kg 3) read the same slots of
// different planes = sixteen distinct 16-byte bank columns.  (Rounds 3-5 kept a pixel's 32 bytes together at a 48-byte pitch: with those lane
// groups 5 of a group's 16 lanes met an occupied bank row on every read of the second conv.)
constexpr int SPLANE = (S0H * 2 * S0WH * 16 + 255) / 256 * 256;  // 9472
constexpr int STILE = 2 * SPLANE;                                // 18944
// first conv k = KS0 (3: yolov8, pad 1 | 6: yolov5, pad 2), stride 2: input patch (2 (S0H - 1) + KS0) x (2 (S0W - 1) + KS0) = 35 x 67 | 38 x 70
template <int KS0>
struct Geo {
  static constexpr int PAD0 = KS0 == 3 ? 1 : 2;
  static constexpr int PR = 2 * (S0H - 1) + KS0, PC = 2 * (S0W - 1) + KS0;
  static constexpr int NCH = (PC + 7 + 7) / 8;   // 16-byte chunks per patch line (origin rounded down to 8 pixels)
  static constexpr int LS = NCH * 8;
  static constexpr int ITEMS = 3 * PR * NCH;
  static constexpr int KTOT = 3 * KS0 * KS0, KSTEPS = (KTOT + 31) / 32;   // im2col depth of a stem pixel: 27 -> 1 k-step, 108 -> 4
  static constexpr int SHIFT = (-(2 + PAD0)) & 7;  // input origin 4 * (16 txi) - 2 - PAD0, rounded down to 8 pixels
  static constexpr int items_pad(int nw) { return (ITEMS + nw * 64 - 1) / (nw * 64) * (nw * 64); }
  static constexpr int patch_bytes(int nw) { return items_pad(nw) * 16; }
};
}  // namespace sf

// -DUPA_STEM_PROF (tools/experiments/r05_stem_phases.sh builds it into a separate library): wave 0 adds the shader cycles it spends per
// phase of the fused stem kernel to g_stem_prof (read and cleared by upa_debug_stem_prof); the product build carries none of it.
#ifdef UPA_STEM_PROF
__device__ unsigned long long g_stem_prof[8];  // wait for the patch, issue the next patch, stage 2, barrier, stage 3, tiles, workgroups
#define SP_DECL unsigned long long sp_t = __builtin_amdgcn_s_memtime(), sp_acc[6] = {0, 0, 0, 0, 0, 0}
#define SP_AT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sp_acc[i] += t_ - sp_t; sp_t = t_; } while (0)
#define SP_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_stem_prof[i_], sp_acc[i_]); atomicAdd(&g_stem_prof[6], 1ull); } } while (0)
extern "C" int upa_debug_stem_prof(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stem_prof), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stem_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
#define SP_DECL
#define SP_AT(i)
#define SP_FLUSH
#endif

// NW waves per workgroup: 4 (two workgroups = two waves per SIMD) or 8 (four per SIMD: the segments of stage 2 and the rows of
// stage 3 are spread over twice the waves, the same LDS).
template <int NW, int KS0 = 3>
__global__ __launch_bounds__(NW * 64, NW / 2) void stem_conv_fused_kernel(const StemFusedParams p) {
  using namespace sf;
  using G0 = Geo<KS0>;
  constexpr int PR = G0::PR, NCH = G0::NCH, LS = G0::LS, ITEMS = G0::ITEMS;
  constexpr int ITEMS_PAD = G0::items_pad(NW);
  constexpr int PATCH = G0::patch_bytes(NW);
  constexpr int NTH = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char fsm[];  // [2][PATCH] input patches, [STILE] stem tile
  char* stile = fsm + 2 * PATCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, l16 = lane & 15;
  const int ntiles = p.tilesX * p.tilesY * p.N;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int plane = p.H * p.W;
  // this thread's items of a patch (channel plane, row, 8-pixel column group) do not depend on the tile: decomposed once
  constexpr int NIT = ITEMS_PAD / NTH;
  int it_row[NIT], it_col[NIT], it_off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = it * NTH + tid;
    const int line = item / NCH, ch = item - line * NCH;
    const int ci = line / PR, row = line - ci * PR;
    it_row[it] = item < ITEMS ? row : (1 << 28);  // past the patch: never inside the image
    it_col[it] = ch * 8;
    it_off[it] = ci * plane + row * p.W + ch * 8;
  }
  auto stage = [&](int tile, char* buf) __attribute__((always_inline)) {
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    // output tile origin (oy0, ox0) -> stem origin (2*oy0 - 1, 2*ox0 - 1) -> input origin (2*sy0 - 1, 2*sx0 - 1)
    const int iy0 = 2 * (2 * tyi * T1H - 1) - G0::PAD0, ix0 = 2 * (2 * txi * T1W - 1) - G0::PAD0;
    const int ixa = ix0 & ~7;
    const bf16_t* xb = (const bf16_t*)p.x + (size_t)n * 3 * plane;
    const bf16_t* xt = xb + iy0 * p.W + ixa;  // patch origin (may lie outside the image: only in-image items are read)
    // the whole staged patch (PR rows x LS columns) inside the image (76 % of the tiles at 640 x 640; workgroup-uniform): no per-item checks
    const bool inside = iy0 >= 0 && iy0 + PR <= p.H && ixa >= 0 && ixa + LS <= p.W;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int first = it * NTH + wave * 64;   // this wave's 64 items of the pass (wave-uniform)
      if (first >= ITEMS) continue;             // all padding: nothing reads those LDS bytes (7 of 8 waves in the last of the k = 3 form's 3 passes)
      if (inside && first + 64 <= ITEMS) {
        __builtin_amdgcn_global_load_lds((sgptr_t) reinterpret_cast<const char*>(xt + it_off[it]), (slptr_t)(buf + first * 16), 16, 0, 0);
        continue;
      }
      const int iy = iy0 + it_row[it], ix = ixa + it_col[it];
      const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const char* src = in ? reinterpret_cast<const char*>(xt + it_off[it]) : reinterpret_cast<const char*>(g_stem_zero16);
      __builtin_amdgcn_global_load_lds((sgptr_t)src, (slptr_t)(buf + first * 16), 16, 0, 0);
    }
  };
  // XCD-aware walk (common.h: upa_xcd_tile): slot = blockIdx.x + i * gridDim.x names the XCD by slot & 7 when the grid is a multiple
  // of 8 (or one round), and XCD x then owns a contiguous range of tiles - neighbouring patches share their 128-byte lines in ONE L2
  const bool xcd = !p.no_xcd && ((gridDim.x & 7) == 0 || (int)gridDim.x >= ntiles);
  auto tile_of = [&](int slot) __attribute__((always_inline)) { return xcd ? upa_xcd_tile(slot, ntiles) : slot; };
  int slot = blockIdx.x;
  if (slot >= ntiles) return;
  int tile = tile_of(slot);
  stage(tile, fsm);
  f32x4 bias0;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias0[r] = p.b0 ? p.b0[kg * 4 + r] : 0.f;
  // ---- stem weights.  The im2col row of a stem pixel as NM 16-deep k-steps (v_mfma_f32_16x16x16_bf16): a lane group takes one patch LINE
  // (ci, kh) and FOUR consecutive bf16 elements of it, so a lane's B fragment is ONE 4-byte-aligned ds_read2_b32 - no 2-byte gathers, no packs
  // (rounds 1-5: eight ds_read_u16 + four packs per 32-deep k-step, 5.3 M LDS bank-conflict cycles per launch of the yolov8n pair).
  //   k = 3 (yolov8; a pixel's taps are elements 2c + 5 .. 2c + 7 of the line): elements 2c + 4 .. 2c + 7 = one element of zero weight, then
  //       kw = 0, 1, 2; nine lines = 3 steps.
  //   k = 6 (yolov5; taps at 2c + 4 .. 2c + 9): two HALF-lines, elements 2c + 4 .. + 7 (kw 0-3) and 2c + 8 .. + 11 (kw 4, 5 and two elements of
  //       zero weight, still inside the 80-element line); 18 lines x 2 = 9 steps.
  // Lines are dealt so that the two lane groups sharing an LDS half-wave (kg 0 | 1 and kg 2 | 3; ds_read2_b32 banks = dword mod 32, line pitch
  // 40 dwords: window start 8 (L mod 4) + 2 half) read 16-bank windows 16 banks apart - lines whose indices differ by 2 mod 4, same half - or
  // overlapping windows of ONE line (same dwords: a broadcast).  stem_line(): (ci, kh, half, live) of lane group kg in step m.
  constexpr int NM = KS0 == 3 ? 3 : 9;
  u32x2 a3[NM];
  int goff3[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    int ci, kh, half = 0;
    bool live = true;
    if constexpr (KS0 == 3) {
      constexpr int LCI[3][4] = {{0, 0, 0, 1}, {1, 2, 1, 2}, {2, 2, 2, 2}}, LKH[3][4] = {{0, 2, 1, 0}, {1, 0, 2, 1}, {2, 2, 2, 2}};
      ci = LCI[m][kg]; kh = LKH[m][kg];
      live = m < 2 || kg == 0;   // the ninth line (2, 2) in lane group 0 of the third step; the other groups re-read it with zero weights
    } else {
      // line index L = 38 ci + kh; L mod 4: class 0 {(0,0) (0,4) (1,2) (2,0) (2,4)}, 2 {(0,2) (1,0) (1,4) (2,2)}, 1 {(0,1) (0,5) (1,3) (2,1) (2,5)},
      // 3 {(0,3) (1,1) (1,5) (2,3)}: eight pairs (class 0 | 2, class 1 | 3), each once per half, + the two left-over lines as (half 0 | half 1)
      constexpr int PA[10][2] = {{0, 0}, {0, 4}, {1, 2}, {2, 0}, {0, 1}, {0, 5}, {1, 3}, {2, 1}, {2, 4}, {2, 5}};   // first line of pair slot / 2
      constexpr int PB[10][2] = {{0, 2}, {1, 0}, {1, 4}, {2, 2}, {0, 3}, {1, 1}, {1, 5}, {2, 3}, {2, 4}, {2, 5}};   // second line
      const int slot = 2 * m + (kg >> 1);          // 18 pair slots: 0-15 = pair (slot >> 1), half (slot & 1); 16, 17 = the left-over lines
      const int pr = slot < 16 ? slot >> 1 : 8 + (slot - 16);
      ci = (kg & 1) ? PB[pr][0] : PA[pr][0];
      kh = (kg & 1) ? PB[pr][1] : PA[pr][1];
      half = slot < 16 ? (slot & 1) : (kg & 1);
    }
    float wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kw = KS0 == 3 ? e - 1 : 4 * half + e;
      wv[e] = (live && kw >= 0 && kw < KS0) ? p.w0[(((kh * KS0 + kw) * 3) + ci) * 16 + l16] : 0.f;
    }
    a3[m] = u32x2{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3])};
    goff3[m] = ((ci * PR + kh) * LS + (KS0 == 3 ? G0::SHIFT - 1 : G0::SHIFT + 4 * half)) * 2;
  }
  // second conv: A fragments [k-step][n-tile] from the standard packed layout [tap][1 k-tile][2 n-tiles][lane][16 B]:
  // lane (kg, r) of k-step s needs W[co = nt*16 + r][ci = (kg&1)*8 ..+7][tap = 2s + (kg>>1)] = the 16 bytes of packed lane
  // ((kg&1)*16 + r) of that tap
  u32x4 a1[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int tap = 2 * s + (kg >> 1);
      a1[s][nt] = tap < 9 ? *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(tap * 2 + nt) * 64 + (kg & 1) * 16 + l16) * 16)
                          : u32x4{0u, 0u, 0u, 0u};
      // ODD stem columns keep the two 4-channel halves of each 8-channel group swapped in the stem tile (below: conflict-free ds_write_b64);
      // the taps that read odd columns (2 l16 + kw with kw = 1) take their k order from the same swap
      if (tap < 9 && tap % 3 == 1) a1[s][nt] = u32x4{a1[s][nt][2], a1[s][nt][3], a1[s][nt][0], a1[s][nt][1]};
    }
  f32x4 bias1[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias1[nt][r] = p.b1 ? p.b1[nt * 16 + kg * 4 + r] : 0.f;
  // this wave's segments of the 17 x 33 stem tile (segment = 16 consecutive pixels of the linearised tile; wave + i NW): byte offset of the lane's
  // pixel inside a patch and of its record in the stem tile (-1: past the tile, never stored)
  constexpr int NSEG_ = (S0H * S0W + 15) / 16, SEGW = (NSEG_ + NW - 1) / NW;
  int sg_in[SEGW], sg_out[SEGW];
#pragma unroll
  for (int i = 0; i < SEGW; ++i) {
    const int q = (wave + i * NW) * 16 + l16;
    const bool qin = q < S0H * S0W;
    const int qq = qin ? q : S0H * S0W - 1;
    const int r = qq / S0W, c = qq - r * S0W;
    sg_in[i] = ((2 * r) * LS + 2 * c) * 2;
    // a lane group (one kg) writes sixteen pixels' 8-byte pieces at a 16-byte pitch: even columns at dwords 4 (c / 2) + 2 (kg & 1), odd ones 68
    // further - the same banks mod 32 (a 2-way conflict on every ds_write_b64 of stage 2) unless the odd columns store the two 4-channel halves
    // of a slot swapped: then the two sets differ by two dwords and interleave
    sg_out[i] = qin ? (kg >> 1) * SPLANE + ((r * 2 + (c & 1)) * S0WH + (c >> 1)) * 16 + ((kg & 1) ^ (c & 1)) * 8 : -1;
  }
  SP_DECL;
  // (Two restructurings measured with the phase profile, tools/experiments/r05_stem_phases.py, and dropped - profiles/r05_stem_phases.txt:
  // LDS-only barriers + the wait for the next patch moved behind the second conv's matrix work moved the waits, not the 11.2 k cycles a
  // tile takes - the two workgroups of a CU fill each other's waits; requesting segment s + NW's gathers before segment s is multiplied
  // made stage 2 slower, 3454 -> 3705 cycles per tile, the kernel 52.9 -> 56.8 us.)
  for (int cur = 0;; cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // patch `cur` landed; everyone is done with the stem tile and the other patch buffer
    SP_AT(0);
    slot += gridDim.x;
    const int next = slot < ntiles ? tile_of(slot) : ntiles;
    if (next < ntiles) stage(next, fsm + (cur ^ 1) * PATCH);
    SP_AT(1);
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * T1H, ox0 = txi * T1W;
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;  // stem coordinates of the stem tile origin
    // the 17 x 33 stem tile inside the stem map (72 % of the tiles at 640 x 640): no padding mask (workgroup-uniform)
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + S0H <= p.H0 && sx0 + S0W <= p.W0;
    const char* pb = fsm + cur * PATCH;
    // ---- stage 2: the 17 x 33 stem tile, 16 stem pixels per MFMA; segment = 16 consecutive pixels of the linearised tile.
    // Two copies of the loop: interior tiles (above) carry no padding mask and no tail-segment selects on the gather.
    constexpr int NSEG = (S0H * S0W + 15) / 16;  // 36
    // Per-lane segment geometry is hoisted out of the tile loop (sg_in / sg_out / sg_rc); the gathers of CH segments are issued before the first
    // is multiplied (k = 3: three - 126 registers, four would spill at the 128 that two workgroups per CU allow; 50.6 -> 48.6 us, same-box A/B;
    // k = 6: one segment = nine reads in flight)
    constexpr int CH = KS0 == 3 ? 3 : 1;
    auto stage2x = [&](auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
      for (int i0 = 0; i0 < SEGW; i0 += CH) {
        u32x2 b[CH][NM];
#pragma unroll
        for (int i = i0; i < i0 + CH && i < SEGW; ++i)
          if (wave + i * NW < NSEG) {   // (kept a run-time branch although it is always taken for i < 4: folded, the scheduler hoists more gathers and spills 15 registers - 54.1 vs 46.8 us)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              const unsigned* src = reinterpret_cast<const unsigned*>(pb + sg_in[i] + goff3[m]);
              b[i - i0][m] = u32x2{src[0], src[1]};
            }
          }
#pragma unroll
        for (int i = i0; i < i0 + CH && i < SEGW; ++i)
          if (wave + i * NW < NSEG) {   // (kept a run-time branch although it is always taken for i < 4: folded, the scheduler hoists more gathers and spills 15 registers - 54.1 vs 46.8 us)
            f32x4 acc = bias0;
#pragma unroll
            for (int m = 0; m < NM; ++m)
              acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<stem_s16x4*>(&a3[m]), *reinterpret_cast<stem_s16x4*>(&b[i - i0][m]), acc, 0, 0, 0);
            bool inmap = true;
            if constexpr (MASKED) {
              // (row, column) back out of sg_in = 4 (LS r + c): r = floor(x 205 / 2^16) is exact for x = 320 r + 4 c, r <= 16, c <= 32 (LS = 80)
              static_assert(LS == 80, "the multiply-shift below divides by 4 LS = 320");
              const int rr_ = (sg_in[i] * 205) >> 16, cc_ = (sg_in[i] - rr_ * (4 * LS)) >> 2;
              const int sy = sy0 + rr_, sx = sx0 + cc_;
              inmap = sy >= 0 && sy < p.H0 && sx >= 0 && sx < p.W0;
            }
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float u = acc[t];
              v[t] = inmap ? u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)) : 0.f;
            }
            // (only the tile's last segment has lanes past its end: the store of every other segment needs no exec mask)
            if ((i + 1) * NW * 16 <= S0H * S0W || sg_out[i] >= 0) *reinterpret_cast<u32x2*>(stile + sg_out[i]) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
      }
    };
    if (interior) stage2x(std::false_type{});
    else stage2x(std::true_type{});
    SP_AT(2);
    __syncthreads();
    SP_AT(3);
    // ---- stage 3: second conv, a wave owns T1H / NW output rows (16 pixels each) x 32 channels
#pragma unroll
    for (int rr = 0; rr < T1H / NW; ++rr) {
      const int i = wave * (T1H / NW) + rr;
      f32x4 acc[2] = {bias1[0], bias1[1]};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        int tap = 2 * s + (kg >> 1);
        if (tap > 8) tap = 8;  // zero weights there
        const int kh = tap / 3, kw = tap - kh * 3;
        const int sr = 2 * i + kh, sc = 2 * l16 + kw;
        const u32x4 b = *reinterpret_cast<const u32x4*>(stile + (kg & 1) * SPLANE + ((sr * 2 + (sc & 1)) * S0WH + (sc >> 1)) * 16);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a1[s][nt]), *reinterpret_cast<const bf16x8*>(&b),
                                                            acc[nt], 0, 0, 0);
      }
      const int oy = oy0 + i, ox = ox0 + l16;
      float v0[4], v1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v0[t] = acc[0][t] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[0][t]));
        v1[t] = acc[1][t] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[1][t]));
      }
      auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
      const int cb = 16 * (kg & 1) + 8 * (kg >> 1);
      if (oy < p.OH && ox < p.OW)
        *reinterpret_cast<u32x4*>(p.y + ((size_t)((n * p.OH + oy) * p.OW + ox) * p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    }
    SP_AT(4);
#ifdef UPA_STEM_PROF
    sp_acc[5] += 1;
#endif
    tile = next;
    if (tile >= ntiles) break;
  }
  SP_FLUSH;
}

// ---- The same fusion for yolov8s' stem pair: Conv(3, 32, 3, 2) -> Conv(32, 64, 3, 2), both SiLU (yolov8.yaml rows 0-1 at width 0.5).
// The 32-channel intermediate (210 MB at bs 32: written by one launch, read back by the next at 83 + 106 us) stays in LDS.  Same patch,
// same 8 x 16 output tile and the same stem-tile layout [row][column parity][column / 2] as above, with 64-byte pixel records at an
// 80-byte pitch; what changes is the shape of the two GEMMs:
//   * stage 2: the gathered im2col fragment of 16 stem pixels feeds TWO MFMAs (32 stem channels = 2 n-tiles);
//   * stage 3: 32 input channels are exactly one k-step per tap (9 k-steps, no tap pairing) and there are 4 n-tiles of output channels.
//     16 waves: wave = (n-tile, row pair), so a wave keeps only ITS n-tile's nine A fragments (36 registers) for the whole kernel and
//     reads 9 pixel fragments per output row; lane (kg, l16) stores channels 16 nt + 4 kg .. + 3 of its pixel as 8 bytes.
namespace sf32 {
using namespace sf;
constexpr int NW = 8, NTH = NW * 64;
constexpr int SP = 80;                               // bytes per stem pixel in LDS (64 used)
constexpr int STILE32 = S0H * 2 * S0WH * SP;         // 46240
// first conv k = 3, pad 1, stride ST0 (2: yolov8s; 1: darknet53's Conv(3, 32, 3, 1) -> Conv(32, 64, 3, 2), yolov3-rtdetr rows 0-1): the
// input patch of the 17 x 33 stem tile is (ST0 * 16 + 3) x (ST0 * 32 + 3) pixels
template <int ST0>
struct GeoS {
  static constexpr int PAD0 = 1;
  static constexpr int PR = ST0 * (S0H - 1) + 3, PC = ST0 * (S0W - 1) + 3;
  static constexpr int NCH = (PC + 7 + 7) / 8;
  static constexpr int LS = NCH * 8;
  static constexpr int ITEMS = 3 * PR * NCH;
  static constexpr int KTOT = 27;
  static constexpr int SHIFT = (-(ST0 + PAD0)) & 7;  // input origin ST0 * (32 txi - 1) - PAD0, rounded down to 8 pixels
  static constexpr int items_pad(int nw) { return (ITEMS + nw * 64 - 1) / (nw * 64) * (nw * 64); }
  static constexpr int patch_bytes(int nw) { return items_pad(nw) * 16; }
};
}  // namespace sf32

template <int ST0>
__global__ __launch_bounds__(512, 4) void stem_conv_fused32_kernel(const StemFusedParams p) {
  using namespace sf32;
  using G0 = GeoS<ST0>;
  constexpr int PR = G0::PR, NCH = G0::NCH, LS = G0::LS, ITEMS = G0::ITEMS;
  constexpr int ITEMS_PAD = G0::items_pad(NW);
  constexpr int PATCH = G0::patch_bytes(NW);
  extern __shared__ __attribute__((aligned(16))) char fsm[];  // [PATCH] input patch, [STILE32] stem tile
  char* stile = fsm + PATCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, l16 = lane & 15;
  const int ntiles = p.tilesX * p.tilesY * p.N;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int plane = p.H * p.W;
  constexpr int NIT = ITEMS_PAD / NTH;
  int it_row[NIT], it_col[NIT], it_off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = it * NTH + tid;
    const int line = item / NCH, ch = item - line * NCH;
    const int ci = line / PR, row = line - ci * PR;
    it_row[it] = item < ITEMS ? row : (1 << 28);
    it_col[it] = ch * 8;
    it_off[it] = ci * plane + row * p.W + ch * 8;
  }
  auto stage = [&](int tile, char* buf) __attribute__((always_inline)) {
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int iy0 = ST0 * (2 * tyi * T1H - 1) - G0::PAD0, ix0 = ST0 * (2 * txi * T1W - 1) - G0::PAD0;
    const int ixa = ix0 & ~7;
    const bf16_t* xb = (const bf16_t*)p.x + (size_t)n * 3 * plane;
    const bf16_t* xt = xb + iy0 * p.W + ixa;
    const bool inside = iy0 >= 0 && iy0 + PR <= p.H && ixa >= 0 && ixa + LS <= p.W;   // (as stem_conv_fused_kernel: no checks, no all-padding passes)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int first = it * NTH + wave * 64;
      if (first >= ITEMS) continue;
      if (inside && first + 64 <= ITEMS) {
        __builtin_amdgcn_global_load_lds((sgptr_t) reinterpret_cast<const char*>(xt + it_off[it]), (slptr_t)(buf + first * 16), 16, 0, 0);
        continue;
      }
      const int iy = iy0 + it_row[it], ix = ixa + it_col[it];
      const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const char* src = in ? reinterpret_cast<const char*>(xt + it_off[it]) : reinterpret_cast<const char*>(g_stem_zero16);
      __builtin_amdgcn_global_load_lds((sgptr_t)src, (slptr_t)(buf + first * 16), 16, 0, 0);
    }
  };
  const bool xcd = !p.no_xcd && ((gridDim.x & 7) == 0 || (int)gridDim.x >= ntiles);
  auto tile_of = [&](int slot) __attribute__((always_inline)) { return xcd ? upa_xcd_tile(slot, ntiles) : slot; };
  int slot = blockIdx.x;
  if (slot >= ntiles) return;
  int tile = tile_of(slot);
  stage(tile, fsm);
  // ---- stem weights: A fragments of the two 16-channel n-tiles; k = (kh*3 + kw)*3 + ci (packed [tap][ci][co32] f32)
  u32x4 a0[2];
  f32x4 bias0[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kg * 8 + j;
      wv[j] = k < G0::KTOT ? p.w0[k * 32 + nt * 16 + l16] : 0.f;
    }
    a0[nt] = u32x4{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3]), pack_bf16x2(wv[4], wv[5]), pack_bf16x2(wv[6], wv[7])};
#pragma unroll
    for (int r = 0; r < 4; ++r) bias0[nt][r] = p.b0 ? p.b0[nt * 16 + kg * 4 + r] : 0.f;
  }
  // ST0 = 2 (yolov8s; the patch geometry of sf::Geo<3>): the im2col row as three 16-deep k-steps, one ds_read2_b32 per lane group and patch line
  // (stem_conv_fused_kernel above: same line table, same bank argument); ST0 = 1 keeps the eight 2-byte gathers (odd / even first taps)
  u32x2 a3[3][2];
  int goff3[3] = {0, 0, 0};
  if constexpr (ST0 == 2) {
    static_assert(ST0 != 2 || (LS == 80 && PR == 35 && G0::SHIFT == 5), "line table and bank windows assume the 35 x 80 patch");
    constexpr int LCI[3][4] = {{0, 0, 0, 1}, {1, 2, 1, 2}, {2, 2, 2, 2}}, LKH[3][4] = {{0, 2, 1, 0}, {1, 0, 2, 1}, {2, 2, 2, 2}};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int ci = LCI[m][kg], kh = LKH[m][kg];
      const bool live = m < 2 || kg == 0;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float wv[4];
        wv[0] = 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) wv[1 + kw] = live ? p.w0[(((kh * 3 + kw) * 3) + ci) * 32 + nt * 16 + l16] : 0.f;
        a3[m][nt] = u32x2{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3])};
      }
      goff3[m] = ((ci * PR + kh) * LS + G0::SHIFT - 1) * 2;
    }
  }
  // second conv: this wave's n-tile, one A fragment per tap from the standard packed layout [tap][1 k-tile][4 n-tiles][lane][16 B]
  const int nt1 = wave & 3, rp = wave >> 2;
  u32x4 a1[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) a1[tap] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(tap * 4 + nt1) * 64 + lane) * 16);
  f32x4 bias1;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias1[r] = p.b1 ? p.b1[nt1 * 16 + kg * 4 + r] : 0.f;
  constexpr int shift = G0::SHIFT;
  int goff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kg * 8 + j;
    int o = 0;
    if (k < G0::KTOT) {
      const int tap = k / 3, ci = k - tap * 3;
      const int kh = tap / 3, kw = tap - kh * 3;
      o = (ci * PR + kh) * LS + kw;
    }
    goff[j] = (o + shift) * 2;
  }
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the patch landed; everyone is done with the stem tile
    slot += gridDim.x;
    const int next = slot < ntiles ? tile_of(slot) : ntiles;
    const int n = tile / tilesPerImg;
    const int t2 = tile - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int oy0 = tyi * T1H, ox0 = txi * T1W;
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + S0H <= p.H0 && sx0 + S0W <= p.W0;
    const char* pb = fsm;
    // ---- stage 2: the 17 x 33 stem tile x 32 channels, 16 stem pixels per MFMA pair
    constexpr int NSEG = (S0H * S0W + 15) / 16;  // 36
    auto stage2 = [&](auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      for (int sg = wave; sg < NSEG; sg += NW) {
        const int q = sg * 16 + l16;
        const bool qin = q < S0H * S0W;
        const int qq = qin ? q : S0H * S0W - 1;
        const int r = qq / S0W, c = qq - r * S0W;
        const char* base = pb + ((ST0 * r) * LS + ST0 * c) * 2;
        [[maybe_unused]] u32x4 b;
        [[maybe_unused]] u32x2 b3[3];
        if constexpr (ST0 == 2) {
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const unsigned* src = reinterpret_cast<const unsigned*>(base + goff3[m]);
            b3[m] = u32x2{src[0], src[1]};
          }
        } else {
          unsigned e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = *reinterpret_cast<const unsigned short*>(base + goff[j]);
          b = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        }
        bool inmap = true;
        if constexpr (MASKED) {
          const int sy = sy0 + r, sx = sx0 + c;
          inmap = sy >= 0 && sy < p.H0 && sx >= 0 && sx < p.W0;
        }
        char* dst = stile + ((r * 2 + (c & 1)) * S0WH + (c >> 1)) * SP + kg * 8;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4 acc = bias0[nt];
          if constexpr (ST0 == 2) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
              acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<stem_s16x4*>(&a3[m][nt]), *reinterpret_cast<stem_s16x4*>(&b3[m]), acc, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a0[nt]), *reinterpret_cast<bf16x8*>(&b), acc, 0, 0, 0);
          }
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float u = acc[t];
            v[t] = inmap ? u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)) : 0.f;
          }
          if (qin) *reinterpret_cast<u32x2*>(dst + nt * 32) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      }
    };
    if (interior) stage2(std::false_type{});
    else stage2(std::true_type{});
    __syncthreads();
    // the patch is dead from here (stage 3 reads the stem tile only): the next tile's patch is fetched under stage 3 into the SAME buffer
    if (next < ntiles) stage(next, fsm);
    // ---- stage 3: second conv; this wave: output rows 4 rp .. 4 rp + 3 (16 pixels each) x its 16 channels
#pragma unroll 2
    for (int rr = 0; rr < 4; ++rr) {
      const int i = 4 * rp + rr;
      f32x4 acc = bias1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int sr = 2 * i + kh, sc = 2 * l16 + kw;
        const u32x4 b = *reinterpret_cast<const u32x4*>(stile + ((sr * 2 + (sc & 1)) * S0WH + (sc >> 1)) * SP + kg * 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a1[tap]), *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
      }
      const int oy = oy0 + i, ox = ox0 + l16;
      float v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = acc[t] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[t]));
      if (oy < p.OH && ox < p.OW)
        *reinterpret_cast<u32x2*>(p.y + ((size_t)((n * p.OH + oy) * p.OW + ox) * p.ldy + nt1 * 16 + kg * 4) * 2) =
            u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    tile = next;
    if (tile >= ntiles) break;
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

static int stem_nchw_impl(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt, const float* bias, void* y,
                          int cout, int ldy, int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream, int pool);

extern "C" int upa_conv2d_stem_nchw(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt,
                                    const float* bias, void* y, int cout, int ldy, int k, int stride, int pad, int act,
                                    int dtype, const upa_opts* opts, void* stream) {
  return stem_nchw_impl(x, x_dtype, n, cin, h, w, wt, bias, y, cout, ldy, k, stride, pad, act, dtype, opts, stream, 0);
}

/* The same first layer followed by nn.MaxPool2d(2, 2, 0) (yolov3-tiny.yaml rows 0-1) as ONE kernel: y = the pooled (n, oh / 2,
 * ow / 2, cout) tensor, the full-resolution activation is never written.  Bit-identical to the two layers run separately.
 * UPA_EUNSUPPORTED outside the fused form (bf16 output, 3 input channels, k = 3, stride 1, SiLU, cout 16 | 32 | 64, even oh / ow):
 * the caller then runs upa_conv2d_stem_nchw and upa_maxpool2d. */
extern "C" int upa_conv2d_stem_nchw_pool2(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt,
                                          const float* bias, void* y, int cout, int ldy, int k, int stride, int pad, int act,
                                          int dtype, const upa_opts* opts, void* stream) {
  return stem_nchw_impl(x, x_dtype, n, cin, h, w, wt, bias, y, cout, ldy, k, stride, pad, act, dtype, opts, stream, 1);
}

static int stem_nchw_impl(const void* x, int x_dtype, int n, int cin, int h, int w, const float* wt, const float* bias, void* y,
                          int cout, int ldy, int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream, int pool) {
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
  p.wgs = UPA_OPT(opts, stem_wgs);
  p.pool = pool;
  p.no_xcd = UPA_OPT(opts, no_xcd);
#ifdef UPA_ABLATE
  p.ablate = UPA_OPT(opts, ablate_stem);
#endif
  const bool no_mfma = UPA_OPT(opts, stem_no_mfma) != 0;
  const int nt16 = cout / 16;
  if (pool && !(dtype == UPA_BF16 && !no_mfma && cin == 3 && cout % 16 == 0 && (nt16 == 1 || nt16 == 2 || nt16 == 4) && k == 3 &&
                stride == 1 && act == UPA_ACT_SILU && p.OH % 2 == 0 && p.OW % 2 == 0 && (long)cin * h * w < (1L << 31) &&
                (long)n * p.OH * p.OW < (1L << 31))) {
    upa_set_error("stem + maxpool: outside the fused form (bf16, 3 -> 16 | 32 | 64 channels, k 3, stride 1, SiLU, even output size)");
    return UPA_EUNSUPPORTED;
  }
  if (dtype == UPA_BF16 && !no_mfma && cin == 3 && cout % 16 == 0 && (nt16 == 1 || nt16 == 2 || nt16 == 4) &&
      ((k == 3 && (stride == 1 || stride == 2)) || (k == 6 && stride == 2)) && (long)cin * h * w < (1L << 31) &&
      (long)n * p.OH * p.OW < (1L << 31)) {
    UPA_CHECK_ARG(act == UPA_ACT_SILU || act == UPA_ACT_NONE, "stem: activation must be SiLU or none");
    hipStream_t stm = (hipStream_t)stream;
    if (k == 3 && stride == 2) launch_stem_mfma_nt<3, 2>(p, n, nt16, stm);
    else if (k == 3) launch_stem_mfma_nt<3, 1>(p, n, nt16, stm);
    else launch_stem_mfma_nt<6, 2>(p, n, nt16, stm);
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
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

/* Fused Conv(3,16,k0,2)+SiLU -> Conv(16,32,3,2,1)+SiLU on a bf16 NCHW input (W % 8 == 0): the 16-channel intermediate never
 * reaches HBM.  k0 = 3 (pad 1: yolov8.yaml rows 0-1) or 6 (pad 2: yolov5 rows 0-1, cfg/models/v5/Detect/yolov5-BoT3.yaml:15-16).
 * w0 / b0: stem weights packed by upa_pack_stem_weight (+ folded bias); w1 / b1: second conv packed by
 * upa_pack_conv_weight(bf16) (+ folded bias).  y: NHWC bf16 view (n, h/4, w/4, 32). */
static int stem_conv_fused_impl(const void* x, int n, int h, int w, int k0, int c0, const float* w0, const float* b0, const void* w1,
                                const float* b1, void* y, int ldy, const upa_opts* opts, void* stream, int s0 = 2) {
  UPA_CHECK_ARG(x && w0 && w1 && y, "stem_conv_fused: null pointer");
  UPA_CHECK_ARG(k0 == 3 || k0 == 6, "stem_conv_fused: first conv k = 3 (pad 1) or 6 (pad 2)");
  UPA_CHECK_ARG(c0 == 16 || (c0 == 32 && k0 == 3), "stem_conv_fused: 3 -> 16 -> 32 channels (k 3 | 6) or 3 -> 32 -> 64 (k 3)");
  UPA_CHECK_ARG(s0 == 2 || (s0 == 1 && c0 == 32), "stem_conv_fused: first conv stride 2, or 1 for the 3 -> 32 -> 64 form");
  UPA_CHECK_ARG(w % 8 == 0 && h % (2 * s0) == 0 && w % (2 * s0) == 0 && (long)3 * h * w < (1L << 31), "stem_conv_fused: w %% 8, h %% (2 s0) == 0 required");
  UPA_CHECK_ARG(ldy % 8 == 0 && (uintptr_t)y % 16 == 0, "stem_conv_fused: output view must be 16-byte aligned");
  StemFusedParams p{};
  p.x = x; p.w0 = w0; p.b0 = b0; p.w1 = (const char*)w1; p.b1 = b1; p.y = (char*)y;
  p.N = n; p.H = h; p.W = w; p.H0 = h / s0; p.W0 = w / s0; p.OH = h / (2 * s0); p.OW = w / (2 * s0); p.ldy = ldy;
  p.tilesX = cdiv(p.OW, sf::T1W); p.tilesY = cdiv(p.OH, sf::T1H);
  p.no_xcd = UPA_OPT(opts, no_xcd);
  const long ntiles = (long)p.tilesX * p.tilesY * n;
  const int wgs = UPA_OPT(opts, stemf_wgs) > 0 ? UPA_OPT(opts, stemf_wgs) : 512;
  const int nw = UPA_OPT(opts, stemf_waves) == 4 ? 4 : 8;
  const dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs));
  hipStream_t st = (hipStream_t)stream;
  if (c0 == 32 && s0 == 1) {  // (patch 8 KB + stem tile 46 KB: two to three 8-wave workgroups per CU)
    (void)upa_full_lds<stem_conv_fused32_kernel<1>>();
    hipLaunchKernelGGL(stem_conv_fused32_kernel<1>, grid, dim3(512), (size_t)sf32::GeoS<1>::patch_bytes(8) + sf32::STILE32, st, p);
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
  if (c0 == 32) {  // two 8-wave workgroups per CU (71 KB of LDS each)
    (void)upa_full_lds<stem_conv_fused32_kernel<2>>();
    hipLaunchKernelGGL(stem_conv_fused32_kernel<2>, grid, dim3(512), (size_t)sf32::GeoS<2>::patch_bytes(8) + sf32::STILE32, st, p);
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
  if (k0 == 6) {  // 4 waves: the 8-wave form of the 108-deep im2col row needs more than the 128 registers two 8-wave workgroups per CU allow (spills)
    (void)upa_full_lds<stem_conv_fused_kernel<4, 6>>();
    hipLaunchKernelGGL((stem_conv_fused_kernel<4, 6>), grid, dim3(256), (size_t)2 * sf::Geo<6>::patch_bytes(4) + sf::STILE, st, p);
  } else if (nw == 4) {
    (void)upa_full_lds<stem_conv_fused_kernel<4, 3>>();
    hipLaunchKernelGGL((stem_conv_fused_kernel<4, 3>), grid, dim3(256), (size_t)2 * sf::Geo<3>::patch_bytes(4) + sf::STILE, st, p);
  } else {
    (void)upa_full_lds<stem_conv_fused_kernel<8, 3>>();
    hipLaunchKernelGGL((stem_conv_fused_kernel<8, 3>), grid, dim3(512), (size_t)2 * sf::Geo<3>::patch_bytes(8) + sf::STILE, st, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
extern "C" int upa_stem_conv_fused(const void* x, int n, int h, int w, const float* w0, const float* b0, const void* w1,
                                   const float* b1, void* y, int ldy, const upa_opts* opts, void* stream) {
  return stem_conv_fused_impl(x, n, h, w, 3, 16, w0, b0, w1, b1, y, ldy, opts, stream);
}
/* The same with the first conv's kernel size given: k0 = 3 | 6 (yolov5's Conv(3, 16, 6, 2, 2)). */
extern "C" int upa_stem_conv_fused_k(const void* x, int n, int h, int w, int k0, const float* w0, const float* b0, const void* w1,
                                     const float* b1, void* y, int ldy, const upa_opts* opts, void* stream) {
  return stem_conv_fused_impl(x, n, h, w, k0, 16, w0, b0, w1, b1, y, ldy, opts, stream);
}
/* The same with the channel count of the first conv given too: c0 = 16 (-> 32 output channels; k0 = 3 | 6) or 32 (-> 64; k0 = 3:
 * yolov8s' Conv(3, 32, 3, 2) -> Conv(32, 64, 3, 2)).  y: NHWC bf16 view (n, h/4, w/4, 2 * c0). */
extern "C" int upa_stem_conv_fused_c(const void* x, int n, int h, int w, int k0, int c0, const float* w0, const float* b0, const void* w1,
                                     const float* b1, void* y, int ldy, const upa_opts* opts, void* stream) {
  return stem_conv_fused_impl(x, n, h, w, k0, c0, w0, b0, w1, b1, y, ldy, opts, stream);
}
/* ... and the first conv's stride: s0 = 2, or 1 for the 3 -> 32 -> 64 form (darknet53's Conv(3, 32, 3, 1) -> Conv(32, 64, 3, 2),
 * cfg/models/v3/Detect/yolov3-rtdetr.yaml rows 0-1).  y: NHWC bf16 view (n, h / (2 s0), w / (2 s0), 2 * c0). */
extern "C" int upa_stem_conv_fused_s(const void* x, int n, int h, int w, int k0, int s0, int c0, const float* w0, const float* b0,
                                     const void* w1, const float* b1, void* y, int ldy, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(s0 == 1 || s0 == 2, "stem_conv_fused: first conv stride 1 | 2");
  return stem_conv_fused_impl(x, n, h, w, k0, c0, w0, b0, w1, b1, y, ldy, opts, stream, s0);
}
