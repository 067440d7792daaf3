// Software-pipelined 3x3 stride-1 convolution for gfx950 (bf16 perf mode): the high-resolution layers of the network.
//
// Same math and packed-weight layout as conv.hip (Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197, BN folded
// per utils/torch_utils.py:236-266, Bottleneck residual nn/modules/block.py:668), different execution shape.  PMC on the
// tile-per-workgroup kernel (64->64 3x3 @ 80x80, bs 32): a wave lived 27k cycles for 2.3k cycles of MFMA - 1270 VALU +
// 580 SALU instructions around 144 MFMAs (swizzled LDS addresses and 64-bit weight pointers recomputed per tap), every
// workgroup resident at once, so load / compute / store phases of the whole grid line up instead of overlapping.  Here:
//   * one wave = one persistent worker owning 8 rows x 16 pixels x (NTW*16) output channels per tile: 32 accumulator
//     tiles (NTW = 4) in AGPRs, no workgroup barriers at all (LDS halo buffers are wave private);
//   * the halo of the NEXT k-chunk / tile is brought in by LDS-DMA while the current one is multiplied (double buffer);
//     weight fragments are prefetched one tap column (3 taps) ahead, so no load latency sits in front of an MFMA;
//   * all geometry is compile time: LDS reads and weight loads use immediate offsets (zero VALU per tap); the LDS image
//     is padded (80 B per pixel per 64-B k-tile) instead of swizzled, which keeps addresses affine and ds_read_b128
//     conflict free;
//   * halo rows are reused across the three vertical taps: 10 row fragments per tap column feed 24 (row, tap) products,
//     30 LDS fragment reads per 288 MFMAs instead of 72;
//   * epilogue straight from the accumulators: bias, SiLU, bf16 pack, a v_permlane16_swap pairs up neighbouring
//     channel quads so every lane stores 16 contiguous bytes (64 B contiguous per pixel per instruction).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* pgptr_t;
typedef __attribute__((address_space(3))) void* plptr_t;

__device__ __attribute__((aligned(16))) unsigned g_pipe_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
constexpr int TH = 8, TW = 16;             // output tile of one wave
constexpr int IH = TH + 2, IW = TW + 2;    // halo
constexpr int PS = 80;                     // LDS bytes per halo pixel: one 64-byte k-tile + 16 pad
constexpr int NSLOT = IH * IW * 5;         // 16-byte DMA slots (4 data + 1 pad per pixel)
constexpr int NDMA = (NSLOT + 63) / 64;    // wave-wide DMA instructions per halo (15)
constexpr int HB = NDMA * 1024;            // bytes per halo buffer
constexpr int WAVES = 4;
}  // namespace

// POOL: the layer is followed by nn.MaxPool2d(2, 2, 0) (yolov3-tiny.yaml rows 2-5): the epilogue takes the maximum of the f32 activations
// over the two rows of a pair in the lane's own registers and over the two pixels of a pair with one lane exchange, rounds to bf16 and
// writes only the pooled (H / 2, W / 2) tensor - bit-identical to conv -> store -> maxpool (rounding is monotone: it commutes with max).
template <int NTW, int ACT, bool RES, bool POOL = false>
__global__ __launch_bounds__(WAVES * 64, 1) void conv3x3_pipe_kernel(const PipeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if UPA_ABL(p, 32) return;  // debug: launch + workgroup dispatch only
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* hb = smem + wave * (2 * HB);  // wave-private double buffer
  const int kg = lane >> 4, p16 = lane & 15;

  // ---- DMA slot decode (tile independent): slot -> halo pixel (py, px) and 16-byte group
  int rel[NDMA], meta[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int slot = k * 64 + lane;
    const int pixel = slot / 5, grp = slot - pixel * 5;
    const int py = pixel / IW, px = pixel - py * IW;
    rel[k] = ((py * p.W + px) * p.ldx + grp * 8) * 2;
    meta[k] = (py == 0 ? 1 : 0) | (py == IH - 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == IW - 1 ? 8 : 0) |
              ((grp == 4 || slot >= NSLOT) ? 16 : 0) | (grp << 5);
  }
  const int gw = blockIdx.x * WAVES + wave, GW = gridDim.x * WAVES;
  int tile = gw;
  if (tile >= p.numTiles) return;

  const int tilesPerImg = p.tilesX * p.tilesY;
  struct TileCtx {
    int n, oy0, ox0, em;
    const char* xb;  // address of halo pixel (0, 0) (may lie outside the tensor: those slots are never fetched)
  };
  auto decode = [&](int t) __attribute__((always_inline)) {
    TileCtx c;
    c.n = t / tilesPerImg;
    const int t2 = t - c.n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    c.oy0 = tyi * TH;
    c.ox0 = txi * TW;
    c.em = (c.oy0 == 0 ? 1 : 0) | (c.oy0 + TH >= p.H ? 2 : 0) | (c.ox0 == 0 ? 4 : 0) | (c.ox0 + TW >= p.W ? 8 : 0);
    c.xb = p.x + ((long)((c.n * p.H + c.oy0 - 1) * p.W + c.ox0 - 1) * p.ldx) * 2;
    return c;
  };
  auto issue_dma = [&](const char* xbase, int emask, int kt, int buf) __attribute__((always_inline)) {
    const int grpmax = (p.Cin - kt * 32 + 7) >> 3;  // valid 16-byte channel groups of this k-tile (>= 4: all)
    const char* xk = xbase + kt * 64;
    if UPA_ABL(p, 1) return;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      const int m = meta[k];
      const bool valid = ((m & (emask | 16)) == 0) && ((m >> 5) < grpmax);
      const char* src = valid ? xk + rel[k] : reinterpret_cast<const char*>(g_pipe_zero16);
      __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(hb + buf * HB + k * 1024), 16, 0, 0);
    }
  };
  // weight fragments: packed [tap][ktile][ntile][lane][16 B]; tap = kh*3 + kw
  const char* wl = p.w + ((size_t)p.nt0 * 1024 + lane * 16);
  const int wTile = p.NTn * 1024;
  u32x4 A[2][3][NTW];
  auto load_A = [&](u32x4(&dst)[3][NTW], int kt, int dx) __attribute__((always_inline)) {
    if UPA_ABL(p, 2) return;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const char* wb = wl + (size_t)((dy * 3 + dx) * p.KTT + kt) * wTile;
#pragma unroll
      for (int j = 0; j < NTW; ++j) dst[dy][j] = *reinterpret_cast<const u32x4*>(wb + j * 1024);
    }
  };
  f32x4 biasv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
    biasv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + (p.nt0 + j) * 16 + kg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 acc[TH][NTW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  const int bfrag = p16 * PS + kg * 16;  // this lane's byte offset inside a halo row fragment

  // one tap column (dx) of one chunk: 10 halo-row fragments x 3 vertical taps
  auto group = [&](const u32x4(&Ac)[3][NTW], const char* hbuf, int dx) __attribute__((always_inline)) {
    if UPA_ABL(p, 8) return;
#pragma unroll
    for (int r = 0; r < IH; ++r) {
      u32x4 b = *reinterpret_cast<const u32x4*>(hbuf + bfrag + (r * IW + dx) * PS);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int i = r - dy;
        if (i < 0 || i >= TH) continue;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&Ac[dy][j]),
                                                              *reinterpret_cast<const bf16x8*>(&b), acc[i][j], 0, 0, 0);
      }
    }
  };

  auto act = [](float v) __attribute__((always_inline)) {
    if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    else return v;
  };
  auto epilogue_pool = [&](const TileCtx& c) __attribute__((always_inline)) {
    static_assert(!POOL || (!RES && (NTW % 2) == 0), "pooled form: no residual, channel tiles in pairs");
    const int co0 = p.nt0 * 16;
    const int HP = p.H >> 1, WP = p.W >> 1;
#pragma unroll
    for (int i = 0; i < TH; i += 2) {
      const unsigned pixp = (unsigned)((c.n * HP + ((c.oy0 + i) >> 1)) * WP + ((c.ox0 + p16) >> 1));
      char* yrow = p.y + ((size_t)pixp * p.ldy + co0) * 2;
#pragma unroll
      for (int j = 0; j + 1 < NTW; j += 2) {
        const int cb = 16 * (j + (kg & 1)) + 8 * (kg >> 1);
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (the max is taken on the f32 activations: the bf16 rounding of the pack below is monotone, so rounding the four values first - what
          // conv -> store -> maxpool does - gives the same bits)
          float a = fmaxf(act(acc[i][j][q] + biasv[j][q]), act(acc[i + 1][j][q] + biasv[j][q]));
          float b = fmaxf(act(acc[i][j + 1][q] + biasv[j + 1][q]), act(acc[i + 1][j + 1][q] + biasv[j + 1][q]));
          v0[q] = fmaxf(a, __shfl_xor(a, 1));  // the neighbouring pixel of the pair (lanes p16 ^ 1 of the same 16-lane row)
          v1[q] = fmaxf(b, __shfl_xor(b, 1));
        }
        auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
        if ((p16 & 1) == 0) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
  };
  auto epilogue = [&](const TileCtx& c) __attribute__((always_inline)) {
    if UPA_ABL(p, 16) return;
    if constexpr (POOL) {
      epilogue_pool(c);
      return;
    }
    const int co0 = p.nt0 * 16;
    unsigned pix = (unsigned)((c.n * p.H + c.oy0) * p.W + c.ox0 + p16);
#pragma unroll
    for (int i = 0; i < TH; ++i, pix += p.W) {
      char* yrow = p.y + ((size_t)pix * p.ldy + co0) * 2;
      const char* rrow = RES ? p.res + ((size_t)pix * p.ldr + co0) * 2 : nullptr;
#pragma unroll
      for (int j = 0; j + 1 < NTW; j += 2) {
        // lanes of 16-lane row kg hold channels 16j+4kg..+3 (tile j) and 16(j+1)+4kg..+3 (tile j+1); after the swap
        // even rows own 8 consecutive channels of tile j, odd rows 8 of tile j+1
        const int cb = 16 * (j + (kg & 1)) + 8 * (kg >> 1);
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = act(acc[i][j][q] + biasv[j][q]);
          v1[q] = act(acc[i][j + 1][q] + biasv[j + 1][q]);
        }
        if constexpr (RES) {
          float x8[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[q]), __float_as_uint(v1[q]), false, false);
            x8[q] = __uint_as_float(s[0]);
            x8[4 + q] = __uint_as_float(s[1]);
          }
          const u32x4 rv = *reinterpret_cast<const u32x4*>(rrow + cb * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            x8[2 * q] += __uint_as_float(rv[q] << 16);
            x8[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
          }
          *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{pack_bf16x2(x8[0], x8[1]), pack_bf16x2(x8[2], x8[3]),
                                                          pack_bf16x2(x8[4], x8[5]), pack_bf16x2(x8[6], x8[7])};
        } else {
          auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if (!UPA_ABL(p, 4)) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      }
      if constexpr (NTW & 1) {
        constexpr int j = NTW - 1;
        const int cb = 16 * j + 4 * kg;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act(acc[i][j][q] + biasv[j][q]);
        if constexpr (RES) {
          const u32x2 rv = *reinterpret_cast<const u32x2*>(rrow + cb * 2);
          v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
          v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
        }
        *reinterpret_cast<u32x2*>(yrow + cb * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  };

  // ---- prologue
  TileCtx cur = decode(tile);
  issue_dma(cur.xb, cur.em, 0, 0);
  load_A(A[0], 0, 0);
  zero_acc();
  int kt = 0, buf = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // One k-chunk (64 B of channels of the current tile).  PAR = which A buffer holds tap column 0.  Order of vector
  // memory operations inside a chunk: [weights column 1] [halo DMA of the next chunk] ... [weights column 2] ...
  // vmcnt(0) before column 2 (its weights are needed there anyway, the DMA is older: it has landed too) ... [weights
  // column 0 of the next chunk] - so no wait ever sits in front of data that was not requested a full column
  // (96 MFMAs at NTW 4) earlier.
  bool running = true;
  auto chunk = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    int nkt = kt + 1, ntile = tile;
    if (nkt == p.KTT) { nkt = 0; ntile = tile + GW; }
    const bool lastOfTile = nkt == 0;
    const bool hasNext = ntile < p.numTiles;
    const char* hbuf = hb + buf * HB;
    load_A(A[PAR ^ 1], kt, 1);
    TileCtx nxt = cur;
    if (hasNext) {
      if (lastOfTile) nxt = decode(ntile);
      issue_dma(nxt.xb, nxt.em, nkt, buf ^ 1);
    }
    group(A[PAR], hbuf, 0);
    load_A(A[PAR], kt, 2);
    group(A[PAR ^ 1], hbuf, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (hasNext) load_A(A[PAR ^ 1], nkt, 0);
    group(A[PAR], hbuf, 2);
    if (lastOfTile) {
      epilogue(cur);
      zero_acc();
    }
    running = hasNext;
    cur = nxt;
    kt = nkt;
    tile = ntile;
    buf ^= 1;
  };
  while (true) {
    chunk(std::integral_constant<int, 0>{});
    if (!running) break;
    chunk(std::integral_constant<int, 1>{});
    if (!running) break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16 -> 16 channel variant (the Bottleneck convs of the first C2f, 160 x 160 at 640 px input: 819 k pixels per launch).
// With 16 input channels a 64-byte k-tile is half padding and the generic kernels spend ~150 VALU instructions per
// 16-pixel MFMA column on addresses and padding (PMC: the layer took the same 33 us with loads, stores and MFMAs removed).
// Here the halo pixel is 32 data bytes (+16 pad) and ONE MFMA k-step covers TWO taps: lane groups kg 0/1 carry the 16
// channels of tap 2s, groups 2/3 those of tap 2s+1 (the A fragment is assembled from the standard packed weights by a lane
// remap; the tenth tap is zero), so a tile row costs 5 MFMAs and 5 immediate-offset ds_read_b128 instead of 9 + 9.
// The epilogue pairs two pixel ROWS through v_permlane16_swap: every lane stores 16 bytes (8 channels of one pixel).
namespace c16 {
constexpr int PS = 48;                       // LDS bytes per halo pixel
constexpr int NSLOT = IH * IW * 3;           // 16-byte DMA slots (2 data + 1 pad per pixel)
constexpr int NDMA = (NSLOT + 63) / 64;      // 9
constexpr int HB = NDMA * 1024;
}  // namespace c16

// NT = output n-tiles per wave: 1 (16 -> 16) or 2 (16 -> 32: yolov3-tiny row 2, 102 k pixels x 32 images at 320 x 320 - on the generic
// pipelined kernel its 64-byte k-tiles are half padding and it ran at 1 TB/s); the B fragment of a tap pair feeds both n-tiles.
// POOL (NT = 2): MaxPool2d(2, 2, 0) of the activations in the epilogue, as conv3x3_pipe_kernel<.., POOL>.
template <int ACT, bool RES, int NT = 1, bool POOL = false>
__global__ __launch_bounds__(WAVES * 64, 2) void conv3x3_c16_kernel(const PipeParams p) {
  static_assert(NT == 1 || (NT == 2 && !RES), "two n-tiles: no residual form");
  static_assert(!POOL || NT == 2, "pooled form: 32 output channels");
  using namespace c16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* hb = smem + wave * (2 * c16::HB);
  const int kg = lane >> 4, p16 = lane & 15;

  int rel[c16::NDMA], meta[c16::NDMA];
#pragma unroll
  for (int k = 0; k < c16::NDMA; ++k) {
    const int slot = k * 64 + lane;
    const int pixel = slot / 3, grp = slot - pixel * 3;
    const int py = pixel / IW, px = pixel - py * IW;
    rel[k] = ((py * p.W + px) * p.ldx + grp * 8) * 2;
    meta[k] = (py == 0 ? 1 : 0) | (py == IH - 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == IW - 1 ? 8 : 0) |
              ((grp == 2 || slot >= c16::NSLOT) ? 16 : 0);
  }
  const int gw = blockIdx.x * WAVES + wave, GW = gridDim.x * WAVES;
  if (gw >= p.numTiles) return;
  const int tilesPerImg = p.tilesX * p.tilesY;
  struct TileCtx {
    int n, oy0, ox0, em;
    const char* xb;
  };
  auto decode = [&](int t) __attribute__((always_inline)) {
    TileCtx c;
    c.n = t / tilesPerImg;
    const int t2 = t - c.n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    c.oy0 = tyi * TH;
    c.ox0 = txi * TW;
    c.em = (c.oy0 == 0 ? 1 : 0) | (c.oy0 + TH >= p.H ? 2 : 0) | (c.ox0 == 0 ? 4 : 0) | (c.ox0 + TW >= p.W ? 8 : 0);
    c.xb = p.x + ((long)((c.n * p.H + c.oy0 - 1) * p.W + c.ox0 - 1) * p.ldx) * 2;
    return c;
  };
  auto issue_dma = [&](const TileCtx& c, int buf) __attribute__((always_inline)) {
    if UPA_ABL(p, 1) return;
#pragma unroll
    for (int k = 0; k < c16::NDMA; ++k) {
      const bool valid = (meta[k] & (c.em | 16)) == 0;
      const char* src = valid ? c.xb + rel[k] : reinterpret_cast<const char*>(g_pipe_zero16);
      __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(hb + buf * c16::HB + k * 1024), 16, 0, 0);
    }
  };
  // A fragments of the five tap pairs (standard packing: [tap][ktile 0][ntile 0][lane][16 B], lanes kg 0/1 = channels 0-15)
  u32x4 A[NT][5];
  int boffs[5];  // this lane's byte offset of its tap inside the halo, relative to (row 0, pixel p16)
#pragma unroll
  for (int s2 = 0; s2 < 5; ++s2) {
    const int tap = 2 * s2 + (kg >> 1);
    const int tapc = tap < 9 ? tap : 8;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      A[nt][s2] = (tap < 9 && !UPA_ABL(p, 2))
                      ? *reinterpret_cast<const u32x4*>(p.w + ((size_t)(tap * p.NTn + nt) * 1024 + ((kg & 1) * 16 + p16) * 16))
                      : u32x4{0u, 0u, 0u, 0u};
    const int dy = tapc / 3, dx = tapc - dy * 3;
    boffs[s2] = ((dy * IW + dx + p16) * c16::PS) + (kg & 1) * 16;
  }
  f32x4 biasn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) biasn[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nt * 16 + kg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 biasv = biasn[0];
  auto act = [](float v) __attribute__((always_inline)) {
    if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    else return v;
  };

  int tile = gw, buf = 0;
  bool first = true;
  TileCtx cur = decode(tile);
  issue_dma(cur, 0);
  while (true) {
    const int ntile = tile + GW;
    const bool hasNext = ntile < p.numTiles;
    TileCtx nxt = cur;
    // this tile's halo must have landed; the TH / 2 stores of the previous tile are the youngest vector-memory operations of
    // the wave (operations retire in issue order) and may stay in flight
    if (first || UPA_ABL(p, 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NT == 2 && !POOL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // TH stores per tile in that form
    else if constexpr (POOL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // its stores are exec-masked (even pixels): not counted on
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    first = false;
    static_assert(TH == 8, "vmcnt(4) above = TH / 2 epilogue stores");
    if (hasNext) {
      nxt = decode(ntile);
      issue_dma(nxt, buf ^ 1);
    }
    const char* hbuf = hb + buf * c16::HB;
    f32x4 accn[NT][TH];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < TH; ++i) accn[nt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!UPA_ABL(p, 8)) {
#pragma unroll
      for (int i = 0; i < TH; ++i) {
#pragma unroll
        for (int s2 = 0; s2 < 5; ++s2) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(hbuf + boffs[s2] + i * (IW * c16::PS));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            accn[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&A[nt][s2]),
                                                                  *reinterpret_cast<const bf16x8*>(&b), accn[nt][i], 0, 0, 0);
        }
      }
    }
    if constexpr (NT == 2) {
      // two n-tiles: lanes of 16-lane row kg hold channels 4kg..+3 (tile 0) and 16 + 4kg..+3 (tile 1) of pixel p16; after the swap
      // even rows own 8 consecutive channels of tile 0, odd rows 8 of tile 1 (as conv3x3_pipe_kernel)
      const int cb = 16 * (kg & 1) + 8 * (kg >> 1);
      if constexpr (POOL) {
        const int HP = p.H >> 1, WP = p.W >> 1;
#pragma unroll
        for (int i = 0; i < TH; i += 2) {
          const unsigned pixp = (unsigned)((cur.n * HP + ((cur.oy0 + i) >> 1)) * WP + ((cur.ox0 + p16) >> 1));
          float v0[4], v1[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = fmaxf(act(accn[0][i][q] + biasn[0][q]), act(accn[0][i + 1][q] + biasn[0][q]));   // (f32 max, then the monotone rounding)
            const float b = fmaxf(act(accn[1][i][q] + biasn[1][q]), act(accn[1][i + 1][q] + biasn[1][q]));
            v0[q] = fmaxf(a, __shfl_xor(a, 1));
            v1[q] = fmaxf(b, __shfl_xor(b, 1));
          }
          auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if ((p16 & 1) == 0 && !UPA_ABL(p, 4))
            *reinterpret_cast<u32x4*>(p.y + ((size_t)pixp * p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      } else {
        unsigned pix = (unsigned)((cur.n * p.H + cur.oy0) * p.W + cur.ox0 + p16);
#pragma unroll
        for (int i = 0; i < TH; ++i, pix += p.W) {
          float v0[4], v1[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v0[q] = act(accn[0][i][q] + biasn[0][q]);
            v1[q] = act(accn[1][i][q] + biasn[1][q]);
          }
          auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if (!UPA_ABL(p, 4)) *reinterpret_cast<u32x4*>(p.y + ((size_t)pix * p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      }
      if (!hasNext) break;
      cur = nxt;
      tile = ntile;
      buf ^= 1;
      continue;
    }
    f32x4 (&acc)[TH] = accn[0];
    // epilogue: rows i (even) and i+1 swap halves - even 16-lane groups end up with 8 consecutive channels of row i, odd
    // groups with 8 of row i+1
    const unsigned pixbase = (unsigned)((cur.n * p.H + cur.oy0) * p.W + cur.ox0 + p16);
#pragma unroll
    for (int i = 0; i < TH; i += 2) {
      float v0[4], v1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] = act(acc[i][q] + biasv[q]);
        v1[q] = act(acc[i + 1][q] + biasv[q]);
      }
      const unsigned pix = pixbase + (unsigned)((i + (kg & 1)) * p.W);
      const int cb = 8 * (kg >> 1);
      char* ydst = p.y + ((size_t)pix * p.ldy + cb) * 2;
      if constexpr (RES) {
        float x8[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[q]), __float_as_uint(v1[q]), false, false);
          x8[q] = __uint_as_float(sw[0]);
          x8[4 + q] = __uint_as_float(sw[1]);
        }
        const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + ((size_t)pix * p.ldr + cb) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x8[2 * q] += __uint_as_float(rv[q] << 16);
          x8[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
        }
        if (!UPA_ABL(p, 4))
          *reinterpret_cast<u32x4*>(ydst) = u32x4{pack_bf16x2(x8[0], x8[1]), pack_bf16x2(x8[2], x8[3]),
                                                  pack_bf16x2(x8[4], x8[5]), pack_bf16x2(x8[6], x8[7])};
      } else {
        auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
        if (!UPA_ABL(p, 4)) *reinterpret_cast<u32x4*>(ydst) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
    if (!hasNext) break;
    cur = nxt;
    tile = ntile;
    buf ^= 1;
  }
}

static int launch_c16(const PipeParams& p, hipStream_t s, const upa_opts* opts) {
  const size_t lds = (size_t)WAVES * 2 * c16::HB;
  const int max_wgs = UPA_OPT(opts, c16_wgs) > 0 ? UPA_OPT(opts, c16_wgs) : 512;
  int grid = (p.numTiles + WAVES - 1) / WAVES;
  if (grid > max_wgs) grid = max_wgs;
  if (p.Cout == 32) {  // two n-tiles per wave (no residual form: the dispatcher keeps those on the generic kernel)
    if (p.pool) {
      (void)upa_full_lds<conv3x3_c16_kernel<UPA_ACT_SILU, false, 2, true>>();
      hipLaunchKernelGGL((conv3x3_c16_kernel<UPA_ACT_SILU, false, 2, true>), dim3(grid), dim3(WAVES * 64), lds, s, p);
    } else if (p.act == UPA_ACT_SILU) {
      (void)upa_full_lds<conv3x3_c16_kernel<UPA_ACT_SILU, false, 2, false>>();
      hipLaunchKernelGGL((conv3x3_c16_kernel<UPA_ACT_SILU, false, 2, false>), dim3(grid), dim3(WAVES * 64), lds, s, p);
    } else {
      (void)upa_full_lds<conv3x3_c16_kernel<UPA_ACT_NONE, false, 2, false>>();
      hipLaunchKernelGGL((conv3x3_c16_kernel<UPA_ACT_NONE, false, 2, false>), dim3(grid), dim3(WAVES * 64), lds, s, p);
    }
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
#define UPA_C16_LAUNCH(ACT_, RES_)                                                                        \
  do {                                                                                                    \
    (void)upa_full_lds<conv3x3_c16_kernel<ACT_, RES_>>();                                                 \
    hipLaunchKernelGGL((conv3x3_c16_kernel<ACT_, RES_>), dim3(grid), dim3(WAVES * 64), lds, s, p);        \
  } while (0)
  if (p.act == UPA_ACT_SILU) {
    if (p.res) UPA_C16_LAUNCH(UPA_ACT_SILU, true);
    else UPA_C16_LAUNCH(UPA_ACT_SILU, false);
  } else {
    if (p.res) UPA_C16_LAUNCH(UPA_ACT_NONE, true);
    else UPA_C16_LAUNCH(UPA_ACT_NONE, false);
  }
#undef UPA_C16_LAUNCH
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int NTW>
static int launch_pipe(const PipeParams& p, int grid, hipStream_t s) {
  const size_t lds = (size_t)WAVES * 2 * HB;
#define UPA_PIPE_LAUNCH(ACT_, RES_)                                                                              \
  do {                                                                                                           \
    auto kern = conv3x3_pipe_kernel<NTW, ACT_, RES_>;                                                            \
    (void)upa_full_lds<conv3x3_pipe_kernel<NTW, ACT_, RES_>>();                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, p);                                           \
  } while (0)
  if constexpr (NTW == 2 || NTW == 4) {
    if (p.pool) {  // SiLU, no residual (checked by upa_conv2d_pool2)
      auto kern = conv3x3_pipe_kernel<NTW, UPA_ACT_SILU, false, true>;
      (void)upa_full_lds<conv3x3_pipe_kernel<NTW, UPA_ACT_SILU, false, true>>();
      hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, p);
      UPA_LAUNCH_CHECK();
      return UPA_OK;
    }
  }
  if (p.act == UPA_ACT_SILU) {
    if (p.res) UPA_PIPE_LAUNCH(UPA_ACT_SILU, true);
    else UPA_PIPE_LAUNCH(UPA_ACT_SILU, false);
  } else {
    if (p.res) UPA_PIPE_LAUNCH(UPA_ACT_NONE, true);
    else UPA_PIPE_LAUNCH(UPA_ACT_NONE, false);
  }
#undef UPA_PIPE_LAUNCH
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

bool upa_conv_pipe_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad,
                            int act, int dtype, const upa_opts* opts) {
  if (UPA_OPT(opts, no_pipe)) return false;
  if (dtype != UPA_BF16 || k != 3 || stride != 1 || pad != 1) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE) return false;
  if (h % TH != 0 || w % TW != 0 || cin % 8 != 0 || cout % 16 != 0) return false;
  if (cin > 128) return false;
  // measured on MI355X (bs 32): the pipelined kernel wins for 32 / 64 output channels per launch (14.2 vs 15.4 us,
  // 28.9 vs 29.4) and for 80->80 (59.1 vs 64.8); 16-channel launches (16->16, the +16 tail of 64->80) re-read the
  // input for too little work (31.3 vs 28.6, 46.9 vs 41.0)
  const int ntn = (cout + 15) / 16;
  const bool all_shapes = UPA_OPT(opts, pipe_all) != 0;
  const bool no_c16 = UPA_OPT(opts, no_c16) != 0;
  const bool c16_shape = cin == 16 && cout == 16 && !no_c16;  // conv3x3_c16_kernel
  if (!all_shapes && !c16_shape && (ntn & 1) && !(ntn >= 5 && cin >= 80)) return false;
  const long px = (long)n * h * w;
  if (px * ldx * 2 >= (1L << 31) || px * ldy * 2 >= (1L << 31) || px * ldr * 2 >= (1L << 31)) return false;
  // enough wave tiles to fill the chip; low-resolution layers stay on the tile-per-workgroup kernel
  const int min_tiles = UPA_OPT(opts, pipe_min_tiles) > 0 ? UPA_OPT(opts, pipe_min_tiles) : 1024;
  if (px / (TH * TW) < min_tiles) return false;
  return true;
}

int upa_conv_pipe_launch(PipeParams p, int query_only, int* variant, void* stream, const upa_opts* opts) {
  hipStream_t s = (hipStream_t)stream;
  p.tilesX = p.W / TW;
  p.tilesY = p.H / TH;
  p.numTiles = p.tilesX * p.tilesY * p.N;
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
#ifdef UPA_ABLATE
  p.ablate = UPA_OPT(opts, ablate_pipe);
#endif
  if (p.Cin == 16 && (p.Cout == 16 || (p.Cout == 32 && !p.res)) && !UPA_OPT(opts, no_c16)) {
    if (variant) *variant = (1 << 21) | (1 << 8) | (p.Cout == 32 ? 2 : 1);
    return query_only ? UPA_OK : launch_c16(p, s, opts);
  }
  const int max_wgs = UPA_OPT(opts, pipe_wgs) > 0 ? UPA_OPT(opts, pipe_wgs) : 256;
  int grid = (p.numTiles + WAVES - 1) / WAVES;
  if (grid > max_wgs) grid = max_wgs;
  // output channels in launches of 64 / 32 / 16 (NTW 4 / 2 / 1); 80 = 64 + 16, 48 = 32 + 16
  int nt = 0, first = 1;
  while (nt < p.NTn) {
    const int left = p.NTn - nt;
    const int ntw = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
    if (first && variant) *variant = (1 << 21) | ntw;
    first = 0;
    if (!query_only) {
      PipeParams q = p;
      q.nt0 = nt;
      int rc = ntw == 4 ? launch_pipe<4>(q, grid, s) : (ntw == 2 ? launch_pipe<2>(q, grid, s) : launch_pipe<1>(q, grid, s));
      if (rc != UPA_OK) return rc;
    }
    nt += ntw;
  }
  return UPA_OK;
}
