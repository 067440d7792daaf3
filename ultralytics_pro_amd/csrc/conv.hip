// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4).
//
//   y[n,oy,ox,co] = act( sum_{kh,kw,ci} x[n, oy*s+kh-p, ox*s+kw-p, ci] * W[co,ci,kh,kw] + bias[co] ) (+ residual)
//
// Replaces the torch dispatch of Conv.forward_fuse (ultralytics/nn/modules/conv.py:188-197) with BN folded
// (utils/torch_utils.py:236-266) and the Bottleneck residual add (nn/modules/block.py:668) fused in the epilogue.
//
// Mapping (one workgroup = WM x WN wavefronts of 64 lanes):
//   * the workgroup owns a TH x TW tile of output pixels of one image (BM = TH*TW = WM*MTW*16 pixels) and
//     BN = WN*NTW*16 output channels;
//   * the input halo tile ((TH-1)s+k) x ((TW-1)s+k) pixels x (<= CKT*64 bytes of channels) is staged ONCE per channel
//     chunk in LDS (zero-filled outside the image and past Cin); every tap (kh,kw) then reads it at a shifted offset,
//     so each input byte is fetched from HBM/L2 once per workgroup instead of k*k times;
//   * GEMM orientation: A = weights (rows = output channels), B = pixels (columns); D[row=co][col=pixel] leaves each
//     lane with 4 consecutive output channels of one pixel -> one 8/16-byte NHWC store per accumulator tile;
//   * weights are pre-packed in exact A-fragment order ([tap][ktile][ntile][lane][16 B]) so a wave's fragment load is
//     one fully coalesced 1 KiB global_load_dwordx4 (served by L2/L1; the whole yolov8n weight set is 6 MB);
//   * bf16: v_mfma_f32_16x16x32_bf16 (one per 64-byte k-tile); f32: 4 x v_mfma_f32_16x16x4_f32 per k-tile - exact f32
//     (k-ordered fmaf chain), used for the <=1e-3 parity mode.  k order inside a tile is permuted identically for A
//     and B (lane group g supplies bytes [16g,16g+16) of the tile), which leaves the sum unchanged.
//   * the LDS halo image is unpadded (the DMA writes contiguous kilobytes); the 16-byte group cg of pixel pl sits in slot
//     cg ^ swz(pl), which keeps the ds_read_b128 fragment reads conflict-free.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

struct ConvParams {
  const char* x;
  char* y;
  const char* res;
  const char* w;
  const float* bias;
  int N, H, W, Cin, ldx;
  int OH, OW, Cout, ldy, ldr;
  int KS, stride, pad;
  int TH, TW, tilesX, tilesY;
  int KTT;  // k-tiles per tap  (Cin padded to the k-tile / channels per k-tile)
  int CKT;  // k-tiles per LDS chunk (1, 2 or 4)
  int NTn;  // n-tiles in the packed weights (Cout padded to 16 / 16)
  int act;
  int IH, IW, PS;  // halo tile dims, LDS pixel stride in bytes
  unsigned magicIW;  // ceil(2^32 / IW)
  int tw_shift;      // log2(TW): tile widths are powers of two
  int stageRows;     // halo rows covered per staging pass = NTHREADS / (IW * G16) (>= 1)
  int numTiles;      // ws kernel: spatial tiles in total (over all images)
  int wsNTB;         // ws kernel: n-tiles per workgroup
#ifdef UPA_ABLATE
  int ablate;        // debug build only: 1 = no input loads, 2 = no weight loads, 4 = no stores, 8 = no MFMA, 32 = return at once
#endif
  const upa_opts* opts;  // HOST side only (dispatch overrides of this call); never read by a kernel
};

// 16 zero bytes in device memory: source of every out-of-image / padded-channel 16-byte group of the halo DMA
__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool PRECISE, int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if constexpr (ACT == UPA_ACT_SILU) {
    if constexpr (PRECISE) return v / (1.0f + expf(-v));  // ocml expf (<= 1 ulp) + IEEE divide: f32 parity mode
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));  // bf16 perf mode: v_exp_f32 + v_rcp_f32 (1 ulp), no IEEE divide
  } else if constexpr (ACT == UPA_ACT_RELU) {
    return fmaxf(v, 0.0f);
  } else {
    return v;
  }
}

// CKT = k-tiles (64 B of channels each) staged per LDS chunk and held per tap in registers; the host picks a CKT that
// divides KTT, so every chunk is full.
template <typename T, int WM, int WN, int MTW, int NTW, int CKT>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if UPA_ABL(p, 32) return;  // debug: launch + workgroup dispatch only
  constexpr int ES = sizeof(T);
  constexpr int E = 16 / ES;      // elements per 16 bytes
  constexpr int KT_CH = 64 / ES;  // channels per k-tile
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int G16 = CKT * 4;    // 16-byte groups per pixel per chunk
  constexpr int G16SHIFT = CKT == 1 ? 2 : (CKT == 2 ? 3 : 4);
  static_assert(CKT == 1 || CKT == 2 || CKT == 4, "CKT must be 1, 2 or 4");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: weight pointers and n-tile tests stay on the SALU
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile decode
  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX;
  const int txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
  const int nt0 = (blockIdx.y * WN + wn) * NTW;  // first n-tile of this wave

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane halo-tile pixel index of the top-left tap of its pixel, for each m-tile
  int pixbase[MTW];
  int pty[MTW], ptx[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int pp = (wm * MTW + i) * 16 + r;
    const int ty = pp >> p.tw_shift;
    const int tx = pp & (p.TW - 1);
    pty[i] = ty;
    ptx[i] = tx;
    pixbase[i] = (ty * p.stride) * p.IW + tx * p.stride;
  }

  const int nChunks = p.KTT / CKT;
  const size_t wTileStride = (size_t)p.NTn * 1024;  // bytes per (tap, ktile)
  const char* wuni = p.w + (size_t)nt0 * 1024;  // wave-uniform part of the weight address (SGPRs)
  const unsigned lane16 = lane * 16;              // per-lane part: global_load ... v_off, s[base] addressing
  const int taps = p.KS * p.KS;
  // Weight fragments of one tap (CKT k-tiles x NTW n-tiles) live in registers, ping-pong buffered: while the MFMAs of
  // tap t run from one buffer, the 1 KiB-per-wave coalesced loads of tap t+1 (or of the next chunk's first tap - they
  // do not depend on the LDS tile) land in the other.
  u32x4 A0[CKT][NTW], A1[CKT][NTW];
  // n-tiles past the packed weights (Cout 80 run as 96) contribute zeros; the common case - every n-tile of the wave
  // exists - is decided once, so a tap's fragment loads are straight-line code (the per-fragment test compiled into a
  // uniform branch and a block of register moves per load)
  const bool wfull = nt0 + NTW <= p.NTn && !UPA_ABL(p, 2);
  auto fetch_tap = [&](u32x4(&dst)[CKT][NTW], int c, int tap) {
    const char* wb = wuni + (size_t)(tap * p.KTT + c * CKT) * wTileStride;
    if (wfull) {
#pragma unroll
      for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) dst[kt][j] = *reinterpret_cast<const u32x4*>(wb + kt * wTileStride + j * 1024 + lane16);
    } else {
      if UPA_ABL(p, 2) return;
#pragma unroll
      for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          dst[kt][j] = (nt0 + j < p.NTn) ? *reinterpret_cast<const u32x4*>(wb + kt * wTileStride + j * 1024 + lane16)
                                         : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // LDS halo image: pixel-major, G16 16-byte slots per pixel, NO padding (the DMA writes 1 KiB contiguous per wave);
  // the 16-byte group cg of pixel pl sits in slot cg ^ swz(pl): conflict-free ds_read_b128 for CKT 1 / 2 (2-way CKT 4)
  auto compute_tap = [&](u32x4(&A)[CKT][NTW], int tapshift) {
    if UPA_ABL(p, 8) return;
    int paddr[MTW], pswz[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int pl = pixbase[i] + tapshift;
      paddr[i] = pl * (G16 * 16);
      pswz[i] = (CKT == 1 ? (pl >> 1) : pl) & (G16 - 1);
    }
#pragma unroll
    for (int kt = 0; kt < CKT; ++kt) {
      u32x4 b[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i)
        b[i] = *reinterpret_cast<const u32x4*>(smem + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if constexpr (ES == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&A[kt][j]),
                                                                *reinterpret_cast<bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
          } else {
            const float* af = reinterpret_cast<const float*>(&A[kt][j]);
            const float* bf = reinterpret_cast<const float*>(&b[i]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  fetch_tap(A0, 0, 0);

  // bias of this wave's output channels (4 consecutive per accumulator tile), fetched once up front
  f32x4 biasv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int co = (nt0 + j) * 16 + g * 4;
    biasv[j] = (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  [[maybe_unused]] f32x4 accT[MTW][NTW];  // f32 parity mode: per-chunk partial sums folded into a running total
  if constexpr (ES == 4) {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) accT[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int c = 0; c < nChunks; ++c) {
    const int c0 = c * CKT * KT_CH;
    if (c > 0) __syncthreads();
    // ---- stage the halo tile of this channel chunk by LDS-DMA (global_load_lds_dwordx4): no VGPR data, no ds_write,
    // every 16-byte item of the tile is in flight at once and the workgroup pays ONE memory round trip (the register
    // staged form paid five in sequence: 12k of the 29k cycles a wave lived).  Each wave-instruction fills 1 KiB of
    // contiguous LDS; out-of-image pixels and channels past Cin are sourced from a 16-byte zero page.
    {
      const int haloItems = p.IH * p.IW * G16;
      const int haloPadded = (haloItems + 63) & ~63;
      const int waveBase = __builtin_amdgcn_readfirstlane(wave * 64);
      for (int base = 0; base < haloPadded; base += NTHREADS) {
        const int wbase = base + waveBase;  // first item of this wave-instruction (uniform)
        if (wbase >= haloPadded) break;
        const int idx = wbase + lane;
        const int pix = idx >> G16SHIFT;
        const int slot = idx & (G16 - 1);
        const int cg = slot ^ ((CKT == 1 ? (pix >> 1) : pix) & (G16 - 1));
        const int py = __umulhi((unsigned)pix, p.magicIW);
        const int px = pix - py * p.IW;
        const int iy = iy0 + py, ix = ix0 + px;
        const int ch = c0 + cg * E;
        const char* src = reinterpret_cast<const char*>(g_zero16);
        if (idx < haloItems && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin && !UPA_ABL(p, 1))
          src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * ES;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + wbase * 16), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // ---- taps, two per trip (ping-pong)
    int kh = 0, kw = 0;
    for (int tap = 0; tap < taps; tap += 2) {
      const bool has1 = tap + 1 < taps;
      if (has1) fetch_tap(A1, c, tap + 1);
      else if (c + 1 < nChunks) fetch_tap(A1, c + 1, 0);
      compute_tap(A0, kh * p.IW + kw);
      if (++kw == p.KS) { kw = 0; ++kh; }
      if (has1) {
        if (tap + 2 < taps) fetch_tap(A0, c, tap + 2);
        else if (c + 1 < nChunks) fetch_tap(A0, c + 1, 0);
        compute_tap(A1, kh * p.IW + kw);
        if (++kw == p.KS) { kw = 0; ++kh; }
      } else if (c + 1 < nChunks) {
        // odd tap count: the next chunk's first tap was fetched into A1
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
          for (int j = 0; j < NTW; ++j) A0[kt][j] = A1[kt][j];
      }
    }
    if constexpr (ES == 4) {
      if (nChunks > 1) {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            accT[i][j] += acc[i][j];
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  }
  if constexpr (ES == 4) {
    if (nChunks > 1) {
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = accT[i][j];
    }
  }

  // ---- bf16 epilogue straight from the accumulators (as conv_pipe.hip): bias, activation, bf16 pack, a
  // v_permlane16_swap pairs neighbouring channel quads so every lane stores 16 contiguous bytes (64 B per pixel per
  // instruction); the residual is read with the same shape.  No LDS round trip, no barrier, and the workgroup needs LDS
  // for its halo only (one more co-resident workgroup per CU on most layers).  The f32 parity mode uses the two-phase
  // LDS form below.
  if constexpr (ES == 2) {
    if (true) {
      auto direct = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        const int cw = (blockIdx.y * WN + wn) * NTW * 16;  // first channel of this wave
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
          const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW && !UPA_ABL(p, 4);
          const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
          char* yrow = p.y + (pixoff * p.ldy + cw) * 2;
          const char* rrow = p.res ? p.res + (pixoff * p.ldr + cw) * 2 : nullptr;
#pragma unroll
          for (int j = 0; j + 1 < NTW; j += 2) {
            const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
            float v0[4], v1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v0[q] = act_fn<false, ACT>(acc[i][j][q] + biasv[j][q]);
              v1[q] = act_fn<false, ACT>(acc[i][j + 1][q] + biasv[j + 1][q]);
            }
            const bool ok = pok && cw + cb < p.Cout;
            if (p.res) {
              float x8[8];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[q]), __float_as_uint(v1[q]), false, false);
                x8[q] = __uint_as_float(sw[0]);
                x8[4 + q] = __uint_as_float(sw[1]);
              }
              if (ok) {
                const u32x4 rv = *reinterpret_cast<const u32x4*>(rrow + cb * 2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  x8[2 * q] += __uint_as_float(rv[q] << 16);
                  x8[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
                }
                *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{pack_bf16x2(x8[0], x8[1]), pack_bf16x2(x8[2], x8[3]),
                                                                pack_bf16x2(x8[4], x8[5]), pack_bf16x2(x8[6], x8[7])};
              }
            } else {
              auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
              auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
              if (ok) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
            }
          }
          if constexpr (NTW & 1) {
            constexpr int j = NTW - 1;
            const int cb = 16 * j + 4 * g;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_fn<false, ACT>(acc[i][j][q] + biasv[j][q]);
            if (pok && cw + cb < p.Cout) {
              if (p.res) {
                const u32x2 rv = *reinterpret_cast<const u32x2*>(rrow + cb * 2);
                v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
                v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
              }
              *reinterpret_cast<u32x2*>(yrow + cb * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
          }
        }
      };
      if (p.act == UPA_ACT_SILU) direct(std::integral_constant<int, UPA_ACT_SILU>{});
      else if (p.act == UPA_ACT_RELU) direct(std::integral_constant<int, UPA_ACT_RELU>{});
      else direct(std::integral_constant<int, UPA_ACT_NONE>{});
      return;
    }
  }

  // ---- epilogue.  Phase 1: every lane writes act(acc + bias) as f32 into an LDS [pixel][channel] tile (the halo
  // buffer is dead by now).  Phase 2: the workgroup streams the tile out as whole NHWC rows - 16 bytes per lane,
  // BN*ES contiguous bytes per pixel (128 B for 64 bf16 channels) - adding the residual from equally coalesced loads.
  // (The direct form stored 8 B per lane in 32-B runs and cost 25 % of the conv time.)
  constexpr int BNB = WN * NTW * 16;        // channels per workgroup
  constexpr int OROW = BNB + 4;             // f32 row stride in LDS (pad 16 B)
  float* otile = reinterpret_cast<float*>(smem);
  __syncthreads();
  auto phase1 = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int pp = (wm * MTW + i) * 16 + r;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int cb = (wn * NTW + j) * 16 + g * 4;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_fn<ES == 4, ACT>(acc[i][j][q] + biasv[j][q]);
        *reinterpret_cast<f32x4*>(otile + pp * OROW + cb) = v;
      }
    }
  };
  if (p.act == UPA_ACT_SILU) phase1(std::integral_constant<int, UPA_ACT_SILU>{});
  else if (p.act == UPA_ACT_RELU) phase1(std::integral_constant<int, UPA_ACT_RELU>{});
  else phase1(std::integral_constant<int, UPA_ACT_NONE>{});
  __syncthreads();
  {
    constexpr int GPP = BNB / E;            // 16-byte output groups per pixel
    constexpr int BMP = WM * MTW * 16;      // pixels per workgroup
    const int cbase = blockIdx.y * BNB;
#pragma unroll 2
    for (int idx = tid; idx < BMP * GPP; idx += NTHREADS) {
      const int pp = idx / GPP, gq = idx - pp * GPP;   // GPP is a compile-time constant
      const int ty = pp >> p.tw_shift, tx = pp & (p.TW - 1);
      const int oy = oy0 + ty, ox = ox0 + tx;
      const int co = cbase + gq * E;
      if (ty >= p.TH || oy >= p.OH || ox >= p.OW || co >= p.Cout || UPA_ABL(p, 4)) continue;
      const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
      const float* src = otile + pp * OROW + gq * E;
      if constexpr (ES == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        if (p.res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + (pixoff * p.ldr + co) * 4);
          v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
        }
        *reinterpret_cast<f32x4*>(p.y + (pixoff * p.ldy + co) * 4) = v;
      } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (p.res) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + (pixoff * p.ldr + co) * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[2 * q] += __uint_as_float(rv[q] << 16);
            v[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
          }
        }
        *reinterpret_cast<u32x4*>(p.y + (pixoff * p.ldy + co) * 2) =
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Weights-stationary persistent variant ("ws"): used when the folded weights of the workgroup's output-channel slice fit
// LDS next to two halo buffers (K*BN*ES <= ~90 KB: every 3x3 conv of yolov8n up to 64->64, every small 1x1).
//   * grid = (min(#tiles, #CUs), Cout slices); a workgroup DMAs its weight slab into LDS ONCE, then walks its spatial
//     tiles; the halo of tile t+1 is DMA'd into the other buffer while tile t is computed, and the compute loop touches
//     LDS only (A and B fragments by ds_read_b128), so there is no vmcnt wait inside it;
//   * one wave per SIMD (WM x WN = 4 waves), big per-wave tiles (MTW x NTW 16x16 tiles, up to 64 px x 64 ch) keep the
//     LDS read traffic at <= 50 % of the LDS rate at full MFMA issue;
//   * KTT (k-tiles per tap = whole Cin) is a template parameter: the halo row holds every input channel.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int MTW, int NTW, int KTT>
__global__ __launch_bounds__(WM* WN * 64) void conv_ws_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if UPA_ABL(p, 32) return;  // debug: launch + workgroup dispatch only
  constexpr int ES = sizeof(T);
  constexpr int E = 16 / ES;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int G16 = KTT * 4;
  constexpr int G16SHIFT = KTT == 1 ? 2 : (KTT == 2 ? 3 : 4);
  constexpr int NTB = WN * NTW;  // n-tiles per workgroup
  static_assert(KTT == 1 || KTT == 2 || KTT == 4, "KTT must be 1, 2 or 4");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int taps = p.KS * p.KS;
  const int ntb0 = blockIdx.y * NTB;          // first n-tile of this workgroup
  const int nt0 = ntb0 + wn * NTW;            // first n-tile of this wave

  const int wBytes = taps * KTT * NTB * 1024;
  const int haloItems = p.IH * p.IW * G16;
  const int haloPadded = (haloItems + 63) & ~63;
  char* wsm = smem;
  char* hbuf0 = smem + wBytes;
  char* hbuf1 = hbuf0 + haloPadded * 16;

  // ---- weights: one DMA pass, [tap][kt][NTB][lane][16 B] (a contiguous copy when the slice covers every n-tile)
  {
    const int items = wBytes >> 4;
    for (int base = wave * 64; base < items; base += NTHREADS) {
      const int idx = base + lane;
      const int seg = idx / (NTB * 64);          // (tap, kt)
      const int within = idx - seg * (NTB * 64);
      const int nt = ntb0 + (within >> 6);
      const char* src = (nt < p.NTn) ? p.w + (((size_t)seg * p.NTn + ntb0) * 64 + within) * 16
                                     : reinterpret_cast<const char*>(g_zero16);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wsm + base * 16), 16, 0, 0);
    }
  }
  auto stage = [&](int tile, char* dst) {
    const int tilesPerImg = p.tilesX * p.tilesY;
    const int n = tile / tilesPerImg;
    const int rem = tile - n * tilesPerImg;
    const int tyi = rem / p.tilesX, txi = rem - tyi * p.tilesX;
    const int iy0 = tyi * p.TH * p.stride - p.pad, ix0 = txi * p.TW * p.stride - p.pad;
    for (int base = wave * 64; base < haloPadded; base += NTHREADS) {
      const int idx = base + lane;
      const int pix = idx >> G16SHIFT;
      const int slot = idx & (G16 - 1);
      const int cg = slot ^ ((KTT == 1 ? (pix >> 1) : pix) & (G16 - 1));
      const int py = __umulhi((unsigned)pix, p.magicIW);
      const int px = pix - py * p.IW;
      const int iy = iy0 + py, ix = ix0 + px;
      const int ch = cg * E;
      const char* src = reinterpret_cast<const char*>(g_zero16);
      if (idx < haloItems && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin)
        src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * ES;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + base * 16), 16, 0, 0);
    }
  };

  int pixbase[MTW], pty[MTW], ptx[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int pp = (wm * MTW + i) * 16 + r;
    pty[i] = pp >> p.tw_shift;
    ptx[i] = pp & (p.TW - 1);
    pixbase[i] = (pty[i] * p.stride) * p.IW + ptx[i] * p.stride;
  }
  f32x4 biasv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int co = (nt0 + j) * 16 + g * 4;
    biasv[j] = (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  int tile = blockIdx.x;
  if (tile < p.numTiles) stage(tile, hbuf0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const char* wlds = wsm + (wn * NTW) * 1024 + lane * 16;
  int buf = 0;
  for (; tile < p.numTiles; tile += gridDim.x, buf ^= 1) {
    const char* hb = buf ? hbuf1 : hbuf0;
    const int next = tile + gridDim.x;
    if (next < p.numTiles) stage(next, buf ? hbuf0 : hbuf1);  // lands while this tile is computed

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kh = 0, kw = 0;
    for (int tap = 0; tap < taps; ++tap) {
      const int tapshift = kh * p.IW + kw;
      int paddr[MTW], pswz[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int pl = pixbase[i] + tapshift;
        paddr[i] = pl * (G16 * 16);
        pswz[i] = (KTT == 1 ? (pl >> 1) : pl) & (G16 - 1);
      }
      const char* wt = wlds + (size_t)tap * KTT * NTB * 1024;
#pragma unroll
      for (int kt = 0; kt < KTT; ++kt) {
        u32x4 a[NTW], b[MTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) a[j] = *reinterpret_cast<const u32x4*>(wt + (kt * NTB + j) * 1024);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
          b[i] = *reinterpret_cast<const u32x4*>(hb + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            if constexpr (ES == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a[j]),
                                                                  *reinterpret_cast<bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
            } else {
              const float* af = reinterpret_cast<const float*>(&a[j]);
              const float* bf = reinterpret_cast<const float*>(&b[i]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
      if (++kw == p.KS) { kw = 0; ++kh; }
    }
    // next tile's halo must have landed before anyone reads it; wait BEFORE this tile's stores are issued (vmcnt also
    // counts stores), so the wait covers only the DMA, which had the whole compute phase to complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue (direct): lane holds 4 consecutive couts of one pixel per accumulator tile
    const int tilesPerImg = p.tilesX * p.tilesY;
    const int n = tile / tilesPerImg;
    const int rem = tile - n * tilesPerImg;
    const int tyi = rem / p.tilesX, txi = rem - tyi * p.tilesX;
    const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
    auto epilogue = [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
        const bool pvalid = pty[i] < p.TH && oy < p.OH && ox < p.OW;
        const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int co = (nt0 + j) * 16 + g * 4;
          if (!pvalid || co >= p.Cout) continue;
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = act_fn<ES == 4, ACT>(acc[i][j][q] + biasv[j][q]);
          if constexpr (ES == 4) {
            if (p.res) {
              const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + (pixoff * p.ldr + co) * 4);
              v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
            }
            *reinterpret_cast<f32x4*>(p.y + (pixoff * p.ldy + co) * 4) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            if (p.res) {
              const u32x2 rv = *reinterpret_cast<const u32x2*>(p.res + (pixoff * p.ldr + co) * 2);
              v[0] += __uint_as_float(rv[0] << 16);
              v[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
              v[2] += __uint_as_float(rv[1] << 16);
              v[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
            }
            *reinterpret_cast<u32x2*>(p.y + (pixoff * p.ldy + co) * 2) =
                u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
        }
      }
    };
    if (p.act == UPA_ACT_SILU) epilogue(std::integral_constant<int, UPA_ACT_SILU>{});
    else if (p.act == UPA_ACT_RELU) epilogue(std::integral_constant<int, UPA_ACT_RELU>{});
    else epilogue(std::integral_constant<int, UPA_ACT_NONE>{});
    __syncthreads();  // every wave is done with hb (and sees the landed next halo) before it is overwritten / read
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned short host_f32_to_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

extern "C" size_t upa_conv_packed_weight_bytes(int cout, int cin, int k, int dtype) {
  const int ktch = dtype == UPA_BF16 ? 32 : 16;
  const int ktt = cdiv(cin, ktch);
  const int ntn = cdiv(cout, 16);
  return (size_t)k * k * ktt * ntn * 1024;
}

// Packed layout: [tap = kh*k+kw][ktile][ntile][lane = g*16 + r][E elements], element j of lane (g, r) =
// W[co = ntile*16 + r][ci = ktile*KT_CH + g*E + j][kh][kw]  (zero beyond cout / cin).
extern "C" int upa_pack_conv_weight(const float* w, int cout, int cin, int k, int dtype, void* out) {
  UPA_CHECK_ARG(w && out && cout > 0 && cin > 0 && k >= 1 && k <= 7, "pack_conv_weight: bad args");
  const int E = dtype == UPA_BF16 ? 8 : 4;
  const int ktch = 4 * E;
  const int ktt = cdiv(cin, ktch), ntn = cdiv(cout, 16);
  size_t idx = 0;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw)
      for (int kt = 0; kt < ktt; ++kt)
        for (int nt = 0; nt < ntn; ++nt)
          for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, r = lane & 15;
            const int co = nt * 16 + r;
            for (int j = 0; j < E; ++j, ++idx) {
              const int ci = kt * ktch + g * E + j;
              float v = 0.f;
              if (co < cout && ci < cin) v = w[(((size_t)co * cin + ci) * k + kh) * k + kw];
              if (dtype == UPA_BF16)
                ((unsigned short*)out)[idx] = host_f32_to_bf16(v);
              else
                ((float*)out)[idx] = v;
            }
          }
  return UPA_OK;
}

namespace {

struct TileCfg {
  int TH, TW;
};

thread_local int g_query_only = 0;   // upa_conv_variant: run the dispatch logic without launching
thread_local int g_last_variant = 0;

template <typename T, int WM, int WN, int MTW, int NTW, int CKT>
int launch_conv_ckt(ConvParams& p, hipStream_t stream) {
  constexpr int BM = WM * MTW * 16;
  constexpr int BN = WN * NTW * 16;
  g_last_variant = (CKT << 16) | (WM << 12) | (WN << 8) | (MTW << 4) | NTW;
  if (g_query_only) return UPA_OK;
  // tile shape: BM pixels as TH x TW
  int TW, TH;
  if (p.KS == 1 && p.stride == 1 && p.pad == 0) {
    // pointwise: flatten (n,h,w) into one pixel row - views have a uniform pixel stride
    const long P = (long)p.N * p.H * p.W;
    p.N = 1; p.H = 1; p.W = (int)P; p.OH = 1; p.OW = (int)P;
    TH = 1; TW = BM;
  } else {
    TW = p.OW >= 16 ? 16 : 8;
    TH = BM / TW;
  }
  p.TH = TH; p.TW = TW;
  p.tw_shift = 0;
  while ((1 << p.tw_shift) < TW) ++p.tw_shift;
  p.tilesX = cdiv(p.OW, TW);
  p.tilesY = cdiv(p.OH, TH);
  p.IH = (TH - 1) * p.stride + p.KS;
  p.IW = (TW - 1) * p.stride + p.KS;
  // extra rows so that pixels past the tile (BM not a multiple of TW) still read inside the allocation
  const int rowsNeeded = (cdiv(BM, TW) - 1) * p.stride + p.KS;
  const int IHalloc = rowsNeeded > p.IH ? rowsNeeded : p.IH;
  p.PS = p.CKT * 64;  // unpadded: LDS-DMA writes contiguous kilobytes; bank conflicts are handled by the XOR swizzle
  p.magicIW = (unsigned)((0x100000000ULL + p.IW - 1) / p.IW);
  p.stageRows = (WM * WN * 64) / (p.IW * CKT * 4);
  if (p.stageRows < 1) p.stageRows = 1;  // halo row wider than the workgroup: threads stride over its columns
  size_t lds = (((size_t)IHalloc * p.IW * (p.CKT * 4) + 63) & ~(size_t)63) * 16 + 1024;
  const size_t ldsOut = (size_t)BM * (BN + 4) * sizeof(float);
  if (sizeof(T) == 4 && ldsOut > lds) lds = ldsOut;  // the bf16 epilogue does not go through LDS
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N), (unsigned)cdiv(p.NTn * 16, BN));
  auto kern = conv_igemm_kernel<T, WM, WN, MTW, NTW, CKT>;
  if (hipError_t e = upa_full_lds<conv_igemm_kernel<T, WM, WN, MTW, NTW, CKT>>(); e != hipSuccess) {
    upa_set_error("conv: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <typename T, int WM, int WN, int MTW, int NTW>
int launch_conv(ConvParams& p, hipStream_t stream) {
  if (p.CKT == 4) return launch_conv_ckt<T, WM, WN, MTW, NTW, 4>(p, stream);
  if (p.CKT == 2) return launch_conv_ckt<T, WM, WN, MTW, NTW, 2>(p, stream);
  return launch_conv_ckt<T, WM, WN, MTW, NTW, 1>(p, stream);
}


// ---- weights-stationary launcher: returns UPA_EUNSUPPORTED when the problem does not qualify (caller falls back)
template <typename T, int WM, int WN, int MTW, int NTW, int KTT>
int launch_ws(ConvParams& p, hipStream_t stream) {
  constexpr int BM = WM * MTW * 16;
  constexpr int NTB = WN * NTW;
  const int taps = p.KS * p.KS;
  const size_t wBytes = (size_t)taps * KTT * NTB * 1024;
  int TW, TH;
  if (p.KS == 1 && p.stride == 1 && p.pad == 0) {
    const long P = (long)p.N * p.H * p.W;
    p.N = 1; p.H = 1; p.W = (int)P; p.OH = 1; p.OW = (int)P;
    TH = 1; TW = BM;
  } else {
    TW = 16;
    TH = BM / TW;
  }
  p.TH = TH; p.TW = TW;
  p.tw_shift = 0;
  while ((1 << p.tw_shift) < TW) ++p.tw_shift;
  p.tilesX = cdiv(p.OW, TW);
  p.tilesY = cdiv(p.OH, TH);
  p.IH = (TH - 1) * p.stride + p.KS;
  p.IW = (TW - 1) * p.stride + p.KS;
  p.magicIW = (unsigned)((0x100000000ULL + p.IW - 1) / p.IW);
  p.CKT = KTT;
  p.PS = KTT * 64;
  const size_t halo = (((size_t)p.IH * p.IW * (KTT * 4) + 63) & ~(size_t)63) * 16;
  const size_t lds = wBytes + 2 * halo + 1024;
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  p.numTiles = p.tilesX * p.tilesY * p.N;
  p.wsNTB = NTB;
  const int gridY = cdiv(p.NTn, NTB);
  static int numCU = 0;
  if (!numCU) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCU <= 0) numCU = 256;
  }
  int perCU = (int)((160 * 1024) / lds);  // co-resident workgroups per CU (LDS-limited), capped by the wave slots
  if (perCU > 4) perCU = 4;
  // measured on MI355X: with one workgroup per CU (4 waves, nothing to overlap the epilogue / LDS latency with) the
  // persistent form loses to the one-tile-per-workgroup kernel at 4 workgroups/CU (64->64 3x3 @80x80: 39.8 vs 28.3 us);
  // with >= 2 co-resident workgroups it wins (32->32 3x3: 15.5 vs 17.3 us, 16->16 3x3 @160x160: 28.7 vs 35.0 us)
  if (perCU < 2) return UPA_EUNSUPPORTED;
  int gx = numCU * perCU / gridY;
  if (gx < 1) gx = 1;
  if (gx > p.numTiles) gx = p.numTiles;
  g_last_variant = (1 << 20) | (KTT << 16) | (WM << 12) | (WN << 8) | (MTW << 4) | NTW;
  if (g_query_only) return UPA_OK;
  auto kern = conv_ws_kernel<T, WM, WN, MTW, NTW, KTT>;
  if (hipError_t e = upa_full_lds<conv_ws_kernel<T, WM, WN, MTW, NTW, KTT>>(); e != hipSuccess) {
    upa_set_error("conv ws: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gridY), dim3(WM * WN * 64), lds, stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <typename T, int KTT>
int dispatch_ws_ktt(ConvParams p, hipStream_t stream) {  // p by value: launch_ws mutates it
  const int ntn = p.NTn;
  const long M = (long)p.N * p.OH * p.OW;
  if (M < 64 * 1024) return UPA_EUNSUPPORTED;              // few tiles: the one-tile-per-workgroup kernel parallelises better
  if (ntn == 1) return launch_ws<T, 4, 1, 4, 1, KTT>(p, stream);   // 256 px x 16 ch
  if (ntn == 2) return launch_ws<T, 4, 1, 4, 2, KTT>(p, stream);   // 256 px x 32 ch
  if (ntn == 4) {
    ConvParams q = p;
    int rc = launch_ws<T, 4, 1, 4, 4, KTT>(q, stream);             // 256 px x 64 ch
    if (rc != UPA_EUNSUPPORTED) return rc;
    return launch_ws<T, 2, 2, 4, 2, KTT>(p, stream);               // 128 px x 64 ch
  }
  return UPA_EUNSUPPORTED;
}

template <typename T>
int dispatch_ws(const ConvParams& p, hipStream_t stream) {
  if (UPA_OPT(p.opts, no_ws)) return UPA_EUNSUPPORTED;
  if (p.KS != 3 || p.stride != 1) return UPA_EUNSUPPORTED;  // 1x1 and stride-2 layers measured faster on the igemm kernel
  if (p.KTT == 1) return dispatch_ws_ktt<T, 1>(p, stream);
  if (p.KTT == 2) return dispatch_ws_ktt<T, 2>(p, stream);
  if (p.KTT == 4) return dispatch_ws_ktt<T, 4>(p, stream);
  return UPA_EUNSUPPORTED;
}

// Tuning hook (development): upa_opts.conv_force = {WM, WN, MTW, NTW} forces one of the extra bf16 instantiations below for
// every conv whose Cout fits it; used by tools/bench_conv.py sweeps, never set in production.
template <typename T>
int dispatch_forced(ConvParams& p, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    const upa_opts* o = p.opts;
    if (!o || o->size < offsetof(upa_opts, conv_force) + sizeof(o->conv_force) || o->conv_force[0] == 0) return UPA_EUNSUPPORTED;
    const int wm = o->conv_force[0], wn = o->conv_force[1], mt = o->conv_force[2], nt = o->conv_force[3];
    if (wn <= 0 || nt <= 0) return UPA_EUNSUPPORTED;
    if (p.NTn % (wn * nt) != 0 && p.NTn > wn * nt) return UPA_EUNSUPPORTED;
    if (p.NTn < wn * nt) return UPA_EUNSUPPORTED;
#define UPA_TRY(A, B, C, D) if (wm == A && wn == B && mt == C && nt == D) return launch_conv<T, A, B, C, D>(p, stream);
    UPA_TRY(4, 1, 4, 2) UPA_TRY(4, 1, 2, 4) UPA_TRY(2, 2, 8, 2) UPA_TRY(4, 2, 2, 2) UPA_TRY(4, 1, 4, 4) UPA_TRY(2, 4, 4, 1)
    UPA_TRY(4, 2, 4, 1) UPA_TRY(8, 1, 2, 2) UPA_TRY(4, 1, 4, 1) UPA_TRY(8, 1, 2, 1) UPA_TRY(4, 2, 4, 2)
#undef UPA_TRY
  }
  return UPA_EUNSUPPORTED;
}

template <typename T>
int dispatch_conv(ConvParams& p, hipStream_t stream) {
  {
    const int rcf = dispatch_forced<T>(p, stream);
    if (rcf != UPA_EUNSUPPORTED) return rcf;
    const int rc = dispatch_ws<T>(p, stream);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  const int ntn = p.NTn;
  const long M = (long)p.N * p.OH * p.OW;
  // n-tiling: prefer covering all output channels in one workgroup (input tile read once)
  // candidates (WM, WN, MTW, NTW): BM = WM*MTW*16, BN = WN*NTW*16
  if (ntn == 1) return launch_conv<T, 4, 1, 2, 1>(p, stream);                     // BN=16,  BM=128
  if (ntn == 2) return launch_conv<T, 4, 1, 2, 2>(p, stream);                     // BN=32,  BM=128
  if (ntn == 3) return launch_conv<T, 4, 1, 2, 3>(p, stream);                     // BN=48
  if (ntn == 5) {  // Cout = 80 (Detect class branch): run as 96 = 2 x 3 n-tiles, the 6th tile is zero-filled
    if (M >= 128 * 1024) return launch_conv<T, 2, 2, 4, 3>(p, stream);            // BN=96, BM=128
    return launch_conv<T, 2, 2, 2, 3>(p, stream);                                 // BN=96, BM=64
  }
  if (ntn % 4 == 0) {
    if (M >= 128 * 1024 || ntn == 4) return launch_conv<T, 2, 2, 4, 2>(p, stream);  // BN=64, BM=128
    return launch_conv<T, 2, 2, 2, 2>(p, stream);                                 // BN=64, BM=64 (small maps)
  }
  if (ntn % 2 == 0) return launch_conv<T, 4, 1, 2, 2>(p, stream);
  return launch_conv<T, 4, 1, 2, 1>(p, stream);
}

}  // namespace

extern "C" int upa_conv2d_bias_act(const void*, int, int, int, int, int, const void*, const float*, void*, int, int,
                                   const void*, int, int, int, int, int, int, const upa_opts*, void*);

// Which template instantiation (WM<<12 | WN<<8 | MTW<<4 | NTW) upa_conv2d_bias_act would launch for this problem;
// lets bench.py attribute algorithmic FLOPs to the kernel names rocprofv3 reports.
extern "C" int upa_conv_variant(int n, int h, int w, int cin, int cout, int k, int stride, int pad, int dtype,
                                const upa_opts* opts) {
  static char dummy[64];
  g_query_only = 1;
  g_last_variant = 0;
  const int E = 16 / upa_elem_size(dtype);
  int rc = upa_conv2d_bias_act(dummy, n, h, w, cin, (cin + E - 1) / E * E, dummy, nullptr, dummy, cout, cout, nullptr, 0, k,
                               stride, pad, 0, dtype, opts, nullptr);
  g_query_only = 0;
  return rc == UPA_OK ? g_last_variant : rc;
}

extern "C" int upa_conv2d_bias_act(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed,
                                   const float* bias, void* y, int cout, int ldy, const void* residual, int ldr, int k,
                                   int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && w_packed && y, "conv2d: null pointer");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv2d: bad shape");
  UPA_CHECK_ARG(k >= 1 && k <= 7 && stride >= 1 && stride <= 2 && pad >= 0 && pad < k, "conv2d: unsupported k/s/p");
  UPA_CHECK_ARG(dtype == UPA_F32 || dtype == UPA_BF16, "conv2d: bad dtype");
  const int es = upa_elem_size(dtype);
  const int E = 16 / es;
  UPA_CHECK_ARG(cin % E == 0 && ldx % E == 0, "conv2d: cin/ldx must be multiples of %d elements", E);
  UPA_CHECK_ARG(cout % E == 0 && ldy % E == 0 && (!residual || ldr % E == 0),
                "conv2d: cout/ldy/ldr must be multiples of %d elements (16-byte row stores)", E);
  UPA_CHECK_ARG(g_query_only || (((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                                 (!residual || (uintptr_t)residual % 16 == 0)), "conv2d: misaligned view");
  if (upa_conv_ws3_eligible(n, h, w, cin, ldx, cout, ldy, residual != nullptr, k, stride, pad, act, dtype, opts)) {
    BigParams q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.res = (const char*)residual; q.w = (const char*)w_packed; q.bias = bias;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.ldr = ldr; q.OH = h; q.OW = w;
    q.KS = 3; q.stride = 1; q.pad = 1; q.act = act;
    const int rc = upa_conv_ws3_launch(q, g_query_only, &g_last_variant, stream, opts);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (upa_conv_p8_eligible(n, h, w, cin, ldx, cout, ldy, residual ? ldr : 0, k, stride, pad, act, dtype, opts)) {
    BigParams q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.res = (const char*)residual; q.w = (const char*)w_packed; q.bias = bias;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.ldr = ldr; q.act = act;
    const int rc = upa_conv_p8_launch(q, g_query_only, &g_last_variant, stream, opts);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (upa_conv_mm_eligible(n, h, w, cin, ldx, cout, ldy, residual ? ldr : 0, k, stride, pad, act, dtype, opts)) {
    BigParams q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.res = (const char*)residual; q.w = (const char*)w_packed; q.bias = bias;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.ldr = ldr; q.act = act;
    const int rc = upa_conv_mm_launch(q, g_query_only, &g_last_variant, stream, opts);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (upa_conv_big_eligible(n, h, w, cin, ldx, cout, ldy, residual ? ldr : 0, k, stride, pad, act, dtype, opts)) {
    BigParams q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.res = (const char*)residual; q.w = (const char*)w_packed; q.bias = bias;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.ldr = ldr;
    q.OH = (h + 2 * pad - k) / stride + 1;
    q.OW = (w + 2 * pad - k) / stride + 1;
    q.KS = k; q.stride = stride; q.pad = pad; q.act = act;
    const int rc = upa_conv_big_launch(q, g_query_only, &g_last_variant, stream, opts);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (upa_conv_pipe_eligible(n, h, w, cin, ldx, cout, ldy, residual ? ldr : 0, k, stride, pad, act, dtype, opts)) {
    PipeParams q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.res = (const char*)residual; q.w = (const char*)w_packed; q.bias = bias;
    q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.ldr = ldr; q.act = act;
    return upa_conv_pipe_launch(q, g_query_only, &g_last_variant, stream, opts);
  }
  if (upa_conv1x1_eligible(n, h, w, cin, ldx, cout, ldy, residual != nullptr, k, stride, pad, act, dtype, opts)) {
    C1Params q;
    memset(&q, 0, sizeof(q));
    q.x = (const char*)x; q.y = (char*)y; q.w = (const char*)w_packed; q.bias = bias;
    q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.act = act;
    return upa_conv1x1_launch(q, n * h * w, g_query_only, &g_last_variant, stream, opts);
  }
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.res = (const char*)residual; p.w = (const char*)w_packed; p.bias = bias;
  p.N = n; p.H = h; p.W = w; p.Cin = cin; p.ldx = ldx;
  p.OH = (h + 2 * pad - k) / stride + 1;
  p.OW = (w + 2 * pad - k) / stride + 1;
  p.Cout = cout; p.ldy = ldy; p.ldr = ldr;
  p.KS = k; p.stride = stride; p.pad = pad; p.act = act;
  p.opts = opts;
#ifdef UPA_ABLATE
  p.ablate = UPA_OPT(opts, ablate_conv);
#endif
  const int ktch = 64 / es;
  p.KTT = cdiv(cin, ktch);
  p.NTn = cdiv(cout, 16);
  // chunk: up to 4 k-tiles (256 B of channels per pixel) for stride 1, 2 for stride 2 (bigger halo)
  // chunk = CKT k-tiles; CKT must divide KTT (every chunk full). Up to 4 (256 B of channels per pixel) for stride 1,
  // 2 for stride 2 (bigger halo tile) and for wide-N variants (register budget of the per-tap weight buffers).
  int ckt = (p.KTT % 4 == 0) ? 4 : ((p.KTT % 2 == 0) ? 2 : 1);
  if ((stride == 2 || p.NTn == 5 || p.NTn == 3) && ckt > 2) ckt = 2;
  if (k > 3 && ckt > 1) ckt = 1;
  if (const int v = UPA_OPT(opts, conv_ckt); (v == 1 || v == 2 || v == 4) && p.KTT % v == 0 && v <= ckt) ckt = v;  // tuning override
  p.CKT = ckt;
  hipStream_t s = (hipStream_t)stream;
  int rc = dtype == UPA_BF16 ? dispatch_conv<bf16_t>(p, s) : dispatch_conv<float>(p, s);
  if (rc == UPA_EUNSUPPORTED) upa_set_error("conv2d: tile does not fit LDS (k=%d s=%d cin=%d)", k, stride, cin);
  return rc;
}

// Training forward of Conv (conv.py:177-186 in train mode): z = conv2d(x) (no bias, no activation) AND nn.BatchNorm2d's batch statistics
// of z - mean / biased variance + the running update - in one call.  Where the layer runs on a kernel with a statistics epilogue
// (conv_big.hip TAIL 3; ...) the per-channel sums come from the convolution's own workgroups (sums of the bf16-rounded values they store,
// one row per pixel tile) and a combine launch adds the rows in a fixed order: no pass over z.  Elsewhere: the convolution, then
// upa_bn_stats + upa_bn_finalize.  ws: upa_channel_reduce_workspace_bytes(cout) bytes (shared with the other reductions, one stream).
int upa_bn_finalize_rows(const float* rows, int nrows, int ld, long npix, int c, float momentum, float* mean, float* var,
                         float* running_mean, float* running_var, void* stream);  // train.hip
extern "C" size_t upa_channel_reduce_workspace_bytes(int c);
extern "C" int upa_bn_stats(const void* z, long npix, int c, int ldz, double* ws, int dtype, void* stream);
extern "C" int upa_bn_finalize(const double* ws, long npix, int c, float momentum, float* mean, float* var, float* running_mean,
                               float* running_var, void* stream);
extern "C" int upa_conv2d_bn_stats(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, void* z, int cout, int ldz,
                                   int k, int stride, int pad, float momentum, float* mean, float* var, float* running_mean,
                                   float* running_var, double* ws, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && w_packed && z && mean && var && ws, "conv2d_bn_stats: null pointer");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && k >= 1 && stride >= 1, "conv2d_bn_stats: bad shape");
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  const long npix = (long)n * oh * ow;
  const int mode = UPA_OPT(opts, no_epi_stats);  // 1 = always the separate reduction (A/B)
  if (!mode && dtype == UPA_BF16 && cout % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)z % 16) == 0) {
    float* rows = reinterpret_cast<float*>(ws);
    const long max_rows = (long)(upa_channel_reduce_workspace_bytes(cout) / ((size_t)2 * cout * sizeof(float)));
    int nrows = 0, rc = UPA_EUNSUPPORTED;
    if (upa_conv_ws3_eligible(n, h, w, cin, ldx, cout, ldz, false, k, stride, pad, UPA_ACT_NONE, dtype, opts)) {
      BigParams q;
      memset(&q, 0, sizeof(q));
      q.x = (const char*)x; q.y = (char*)z; q.w = (const char*)w_packed;
      q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldz; q.OH = h; q.OW = w;
      q.KS = 3; q.stride = 1; q.pad = 1; q.act = UPA_ACT_NONE;
      q.stats = rows;
      rc = upa_conv_ws3_launch_stats(q, &nrows, max_rows, stream, opts);
    } else if (upa_conv_big_eligible(n, h, w, cin, ldx, cout, ldz, 0, k, stride, pad, UPA_ACT_NONE, dtype, opts)) {
      BigParams q;
      memset(&q, 0, sizeof(q));
      q.x = (const char*)x; q.y = (char*)z; q.w = (const char*)w_packed;
      q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldz; q.OH = oh; q.OW = ow;
      q.KS = k; q.stride = stride; q.pad = pad; q.act = UPA_ACT_NONE;
      q.stats = rows;
      rc = upa_conv_big_launch_stats(q, &nrows, max_rows, stream, opts);
    } else if (!upa_conv_ws3_eligible(n, h, w, cin, ldx, cout, ldz, false, k, stride, pad, UPA_ACT_NONE, dtype, opts) &&
               !upa_conv_pipe_eligible(n, h, w, cin, ldx, cout, ldz, 0, k, stride, pad, UPA_ACT_NONE, dtype, opts) &&
               upa_conv1x1_eligible(n, h, w, cin, ldx, cout, ldz, false, k, stride, pad, UPA_ACT_NONE, dtype, opts)) {
      C1Params q;
      memset(&q, 0, sizeof(q));
      q.x = (const char*)x; q.y = (char*)z; q.w = (const char*)w_packed;
      q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldz; q.act = UPA_ACT_NONE;
      q.stats = rows;
      rc = upa_conv1x1_launch_stats(q, n * h * w, &nrows, max_rows, stream, opts);
    }
    if (rc == UPA_OK) return upa_bn_finalize_rows(rows, nrows, cout, npix, cout, momentum, mean, var, running_mean, running_var, stream);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (const int rc = upa_conv2d_bias_act(x, n, h, w, cin, ldx, w_packed, nullptr, z, cout, ldz, nullptr, 0, k, stride, pad, UPA_ACT_NONE,
                                         dtype, opts, stream); rc != UPA_OK)
    return rc;
  if (const int rc = upa_bn_stats(z, npix, cout, ldz, ws, dtype, stream); rc != UPA_OK) return rc;
  return upa_bn_finalize(ws, npix, cout, momentum, mean, var, running_mean, running_var, stream);
}

// Data gradient of a 3x3 stride-2 pad-1 convolution (training): dx (n, h, w, cin) (+)= the four 2x2 phase correlations over dz
// (n, oh, ow, cout) with the stacked phase weights (upa_dgrad_s2_phase_weights packed as a cout -> 4 cin, k = 2 conv), written straight into
// dx's interleaved pixels by the convolution's epilogue - no (n, oh + 1, ow + 1, 4 cin) phase tensor, no upa_interleave2x pass.
// UPA_EUNSUPPORTED (nothing launched) outside conv_big's 128-channel-column shapes: the caller runs conv + upa_interleave2x.
extern "C" int upa_conv2d_dgrad_s2(const void* dz, int n, int oh, int ow, int cout, int lddz, const void* phase_w_packed, void* dx, int h,
                                   int w, int cin, int lddx, int accumulate, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(dz && phase_w_packed && dx, "conv2d_dgrad_s2: null pointer");
  if (dtype != UPA_BF16 || cin % 8 != 0 || lddx % 8 != 0 || lddz % 8 != 0 || ((uintptr_t)dz % 16) != 0 || ((uintptr_t)dx % 16) != 0 ||
      !upa_conv_big_eligible(n, oh, ow, cout, lddz, 4 * cin, 4 * cin, 0, 2, 1, 1, UPA_ACT_NONE, dtype, opts))
    return UPA_EUNSUPPORTED;
  BigParams q;
  memset(&q, 0, sizeof(q));
  q.x = (const char*)dz; q.y = (char*)dx; q.w = (const char*)phase_w_packed; q.res = accumulate ? (const char*)dx : nullptr;
  q.N = n; q.H = oh; q.W = ow; q.Cin = cout; q.ldx = lddz; q.Cout = 4 * cin; q.ldy = lddx; q.ldr = lddx; q.OH = oh + 1; q.OW = ow + 1;
  q.KS = 2; q.stride = 1; q.pad = 1; q.act = UPA_ACT_NONE;
  q.il_h = h; q.il_w = w; q.il_c = cin;
  return upa_conv_big_launch_interleave(q, stream, opts);
}

// The whole training forward of a Conv in ONE call (conv.py:177-186 in train mode): upa_conv2d_bn_stats, then
// y = act(gamma * (z - mean) / sqrt(var + eps) + beta) (+ residual) = upa_bn_act_fwd - the same launches, one crossing of the language
// boundary per layer instead of two (the eager training step issues ~600 launches from Python; its small-map phases are host-bound).
extern "C" int upa_bn_act_fwd(const void* z, long npix, int c, int ldz, const float* mean, const float* var, const float* gamma,
                              const float* beta, float eps, int act, void* y, int ldy, const void* residual, int ldr, int dtype,
                              void* stream);
extern "C" int upa_conv2d_bn_act_fwd(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, void* z, int cout, int ldz,
                                     int k, int stride, int pad, float momentum, float* mean, float* var, float* running_mean,
                                     float* running_var, const float* gamma, const float* beta, float eps, int act, void* y, int ldy,
                                     const void* residual, int ldr, double* ws, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(y && gamma && beta, "conv2d_bn_act_fwd: null pointer");
  if (const int rc = upa_conv2d_bn_stats(x, n, h, w, cin, ldx, w_packed, z, cout, ldz, k, stride, pad, momentum, mean, var, running_mean,
                                         running_var, ws, dtype, opts, stream); rc != UPA_OK)
    return rc;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  return upa_bn_act_fwd(z, (long)n * oh * ow, cout, ldz, mean, var, gamma, beta, eps, act, y, ldy, residual, ldr, dtype, stream);
}

// Conv(k = 3, s = 1, p = 1) + SiLU followed by nn.MaxPool2d(2, 2, 0) as ONE launch (yolov3-tiny.yaml rows 2-7: the full-resolution
// activation - 210 + 105 + 52 MB at batch 32 - is neither written nor read back): y = the pooled (n, h / 2, w / 2, cout) view.
// Bit-identical to upa_conv2d_bias_act + upa_maxpool2d (the pool runs on the bf16-rounded activations in the conv epilogue,
// csrc/conv_pipe.hip).  UPA_EUNSUPPORTED outside the fused form: the caller runs the two layers.
extern "C" int upa_conv2d_pool2(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, const float* bias, void* y,
                                int cout, int ldy, int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && w_packed && y, "conv2d_pool2: null pointer");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv2d_pool2: bad shape");
  const long px = (long)n * h * w;
  if (UPA_OPT(opts, no_pipe) || dtype != UPA_BF16 || k != 3 || stride != 1 || pad != 1 || act != UPA_ACT_SILU || h % 8 != 0 || w % 16 != 0 ||
      cin % 8 != 0 || cin > 128 || cout % 32 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 ||
      px * ldx * 2 >= (1L << 31) || px / 4 * ldy * 2 >= (1L << 31)) {
    upa_set_error("conv2d_pool2: outside the fused form (bf16, k 3 s 1 p 1, SiLU, h %% 8 == 0, w %% 16 == 0, cin <= 128, cout %% 32 == 0)");
    return UPA_EUNSUPPORTED;
  }
  PipeParams q;
  memset(&q, 0, sizeof(q));
  q.x = (const char*)x; q.y = (char*)y; q.w = (const char*)w_packed; q.bias = bias;
  q.N = n; q.H = h; q.W = w; q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.act = act; q.pool = 1;
  return upa_conv_pipe_launch(q, 0, nullptr, stream, opts);
}

// Several independent convolutions with the same kernel size / stride / padding / activation (the first convs of a Detect head's
// branches on different levels): identical to one upa_conv2d_bias_act per problem, but neighbours that land on the same 128-pixel
// conv_big instantiation share ONE grid (conv_big.hip: conv_big_pair_kernel) - a 400- and a 100-workgroup launch become one partial round.
extern "C" int upa_conv2d_bias_act_group(const upa_conv_problem* probs, int count, int k, int stride, int pad, int act, int dtype,
                                         const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(probs && count >= 1 && count <= 8, "conv2d_group: 1..8 problems");
  auto big_ok = [&](const upa_conv_problem& q) {
    return q.x && q.y && q.w_packed && ((uintptr_t)q.x % 16) == 0 && ((uintptr_t)q.y % 16) == 0 && !q.residual &&
           !upa_conv_ws3_eligible(q.n, q.h, q.w, q.cin, q.ldx, q.cout, q.ldy, false, k, stride, pad, act, dtype, opts) &&
           upa_conv_big_eligible(q.n, q.h, q.w, q.cin, q.ldx, q.cout, q.ldy, 0, k, stride, pad, act, dtype, opts);
  };
  auto fill = [&](const upa_conv_problem& q) {
    BigParams b;
    memset(&b, 0, sizeof(b));
    b.x = (const char*)q.x; b.y = (char*)q.y; b.w = (const char*)q.w_packed; b.bias = q.bias;
    b.N = q.n; b.H = q.h; b.W = q.w; b.Cin = q.cin; b.ldx = q.ldx; b.Cout = q.cout; b.ldy = q.ldy;
    b.OH = (q.h + 2 * pad - k) / stride + 1;
    b.OW = (q.w + 2 * pad - k) / stride + 1;
    b.KS = k; b.stride = stride; b.pad = pad; b.act = act;
    b.no_xcd = UPA_OPT(opts, no_xcd);
    return b;
  };
  for (int i = 0; i < count;) {
    if (i + 1 < count && k == 3 && stride == 1 && big_ok(probs[i]) && big_ok(probs[i + 1])) {
      BigParams b[4];
      int m = 2;
      b[0] = fill(probs[i]); b[1] = fill(probs[i + 1]);
      if (i + 2 < count && big_ok(probs[i + 2])) { b[2] = fill(probs[i + 2]); m = 3; }
      if (m == 3 && i + 3 < count && big_ok(probs[i + 3])) { b[3] = fill(probs[i + 3]); m = 4; }
      int consumed = 0;
      const int rc = upa_conv_big_launch_group(b, m, &consumed, stream, opts);
      if (rc == UPA_OK && consumed > 0) { i += consumed; continue; }
      if (rc != UPA_EUNSUPPORTED) return rc;
    }
    const upa_conv_problem& q = probs[i];
    if (const int rc = upa_conv2d_bias_act(q.x, q.n, q.h, q.w, q.cin, q.ldx, q.w_packed, q.bias, q.y, q.cout, q.ldy, q.residual, q.ldr,
                                           k, stride, pad, act, dtype, opts, stream); rc != UPA_OK)
      return rc;
    ++i;
  }
  return UPA_OK;
}

