// C2f(32 -> 32, n = 1, shortcut) as ONE kernel (bf16): cv1 (1x1, 32 -> 2 x 16) -> Bottleneck(3x3 16 -> 16, 3x3 16 -> 16, + input)
// -> cv2 (1x1 over cat(y0, y1, b) = 48 -> 32), every conv with BN folded and SiLU.
//   C2f.forward        ultralytics/nn/modules/block.py:457-488   y = list(cv1(x).chunk(2, 1)); y.extend(m(y[-1]) ...); cv2(cat(y, 1))
//   Bottleneck.forward ultralytics/nn/modules/block.py:644-668   x + cv2(cv1(x))
// This is model.2 of yolov8n at 160 x 160: four launches that move 52 MB in, 52 + 26 + 26 MB of intermediates out and back in
// (one of them three times) and 52 MB out.  Fused, a workgroup owns a 16 x 16 output tile and the intermediates only ever
// exist as LDS tiles:
//   0. x halo tile (20 x 20 px x 32 ch = 64 B / px) by LDS-DMA, zero page outside the image;
//   A. y1 = SiLU(cv1 upper half) on all 400 halo pixels (one v_mfma_f32_16x16x32_bf16 per 16 px), ZERO outside the image (it is
//      the 3x3 conv's padding), bf16 tile of 32 B / px;
//   B. t = SiLU(conv3x3(y1)) on the 18 x 18 inner pixels: 16 input channels are half an MFMA k-step, so a k-step pairs two
//      taps (lane groups 0-1 take tap 2s, groups 2-3 tap 2s + 1): 5 k-steps; zero outside the image; bf16 tile of 32 B / px;
//   C. b = y1 + SiLU(conv3x3(t)) on the 16 x 16 tile, kept in registers: the D layout of a 16 x 16 accumulator tile (lane (g, r):
//      channels 4g .. 4g + 3 of pixel r) is the B-operand layout of v_mfma_f32_16x16x16_bf16, and so are the 8-byte records of
//      the y1 tile; y0 = SiLU(cv1 lower half) of the tile's own pixels is computed here the same way;
//   D. out = SiLU(cv2 . [y0 | y1 | b]) = three 16-wide k-steps per 16-channel n-tile from those three operands; bf16 pack,
//      v_permlane16_swap pairs the two n-tiles into 16-byte NHWC stores.
// All weights (standard upa_pack_conv_weight layout, only the address of a lane's bytes differs) live in registers.
// Rounding points (bf16 y0, y1, t, b, out; f32 accumulation and f32 residual add) are those of the four separate launches.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
UPA_STAMP_DEFINE(c2f)

typedef __attribute__((address_space(1))) const void* cgptr_t;
typedef __attribute__((address_space(3))) void* clptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __attribute__((aligned(16))) unsigned g_c2f_zero16[4] = {0u, 0u, 0u, 0u};

struct C2fParams {
  const char* x; char* y;
  const char *w1, *wa, *wb, *w2;
  const float *b1, *ba, *bb, *b2;
  int N, H, W, ldx, ldy, tilesX, tilesY;
  int xcd;  // 1 = XCD-aware tile order (upa_xcd_tile, common.h)
};

namespace c2f {
constexpr int TH = 16, TW = 16;
constexpr int XH = TH + 4, XW = TW + 4;      // x / y1 halo tile 20 x 20
constexpr int MH = TH + 2, MW = TW + 2;      // t tile 18 x 18
constexpr int XPX = XH * XW, MPX = MH * MW;  // 400, 324
constexpr int XITEMS = XPX * 4;              // 16-byte items of the x tile (a multiple of 64: whole waves)
constexpr int XS_BYTES = XITEMS * 16;        // 25600
// y1 tile: 32 B / px records with a pixel PITCH of 26 instead of 20.  Stage B enumerates its 18 x 18 pixels row-major in 16-pixel
// m-tiles; an m-tile that straddles rows reads pixels whose indices collide mod 8 (= the same banks for ds_read_b128) unless the
// row step is right for the width: upa_lds_pick_pitch(20, 18, 324, 1) = 26 (round 2 measured 48 % conflict cycles at pitch 20)
constexpr int Y1P = 26;
constexpr int Y1_BYTES = XH * Y1P * 32;      // 16640
constexpr int MT_B = (MPX + 15) / 16;        // 21 m-tiles of t
constexpr int TS_BYTES = MT_B * 16 * 32;     // 10752
constexpr int LDS = XS_BYTES + Y1_BYTES + TS_BYTES;

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(&a), *reinterpret_cast<const s16x4*>(&b), c, 0, 0, 0);
}
}  // namespace c2f

// NW waves per workgroup: 4 (84 VGPRs, three workgroups per CU = three waves per SIMD; the default) or 8 (107 VGPRs, two
// workgroups = four per SIMD: measured 52.6 us against 51 and 0.651 against 0.647 ms per step - kept for experiments)
template <int NW>
__global__ __launch_bounds__(NW * 64) void c2f16_fused_kernel(const C2fParams p) {
  using namespace c2f;
  constexpr int NTH = NW * 64;
  constexpr int XIT = (XITEMS + NTH - 1) / NTH;
  static_assert(XITEMS % 64 == 0, "the DMA guard must be wave-uniform");
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xs = sm;
  char* y1s = sm + XS_BYTES;
  char* ts = y1s + Y1_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int tilesPerImg = p.tilesX * p.tilesY;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, tilesPerImg * p.N) : (int)blockIdx.x;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * TH, ox0 = txi * TW;

  // ---- 0. x halo tile
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    if (it * NTH + wave * 64 >= XITEMS) break;  // wave-uniform
    const int item = it * NTH + tid;
    const int px = item >> 2, slot = item & 3;
    const int hy = px / XW, hx = px - hy * XW;
    const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
    const char* src = reinterpret_cast<const char*>(g_c2f_zero16);
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      src = p.x + (((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx * 2 + slot * 16;
    __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(xs + (it * NTH + wave * 64) * 16), 16, 0, 0);
  }

  // ---- weights -> registers while the tile is in flight.  Packed layout: [tap][k-tile][n-tile][lane (g, r)][16 B], lane (g, r) =
  // W[co = 16 nt + r][ci = 32 kt + 8g .. + 7]
  const u32x4 w_y0 = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)(0 * 64 + lane) * 16);
  const u32x4 w_y1 = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)(1 * 64 + lane) * 16);
  u32x4 w_a[5], w_b[5];
  int offB[5], offC[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + (g >> 1);
    const size_t o = ((size_t)(tap < 9 ? tap : 0) * 64 + (g & 1) * 16 + r) * 16;
    w_a[s] = tap < 9 ? *reinterpret_cast<const u32x4*>(p.wa + o) : u32x4{0u, 0u, 0u, 0u};
    w_b[s] = tap < 9 ? *reinterpret_cast<const u32x4*>(p.wb + o) : u32x4{0u, 0u, 0u, 0u};
    const int tc = tap < 9 ? tap : 8;
    const int kh = tc / 3, kw = tc - kh * 3;
    offB[s] = (kh * Y1P + kw) * 32 + (g & 1) * 16;  // into the y1 tile
    offC[s] = (kh * MW + kw) * 32 + (g & 1) * 16;  // into the t tile
  }
  // cv2 as 16-wide k-steps: lane (g, r) of k-step ks, n-tile nt = W2[co = 16 nt + r][ci = 16 ks + 4g .. + 3]
  u32x2 w_2[3][2];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c0 = 16 * ks + 4 * g;
      const int kt = c0 >> 5, gg = (c0 & 31) >> 3, half = (c0 & 7) >> 2;
      w_2[ks][nt] = *reinterpret_cast<const u32x2*>(p.w2 + ((size_t)((kt * 2 + nt) * 64 + gg * 16 + r)) * 16 + half * 8);
    }
  const f32x4 bias_y0 = *reinterpret_cast<const f32x4*>(p.b1 + 4 * g);
  const f32x4 bias_y1 = *reinterpret_cast<const f32x4*>(p.b1 + 16 + 4 * g);
  const f32x4 bias_a = *reinterpret_cast<const f32x4*>(p.ba + 4 * g);
  const f32x4 bias_b = *reinterpret_cast<const f32x4*>(p.bb + 4 * g);
  const f32x4 bias_2[2] = {*reinterpret_cast<const f32x4*>(p.b2 + 4 * g), *reinterpret_cast<const f32x4*>(p.b2 + 16 + 4 * g)};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // interior tiles (the 20 x 20 halo inside the image: 64 % of the tiles at 160 x 160) skip the zero-padding masks of A and B
  const bool interior = oy0 >= 2 && ox0 >= 2 && oy0 + TH + 2 <= p.H && ox0 + TW + 2 <= p.W;  // workgroup-uniform
  auto stage_ab = [&](auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    // ---- A. y1 on the 400 halo pixels
    for (int mt = wave; mt < XPX / 16; mt += NW) {
      const int q = mt * 16 + r;
      const u32x4 b = *reinterpret_cast<const u32x4*>(xs + q * 64 + g * 16);
      const f32x4 acc = mfma32(w_y1, b, bias_y1);
      const int hy = (int)__umulhi((unsigned)q, 0x0CCCCCCDu), hx = q - hy * XW;  // q / 20 (exact for q < 2^17)
      bool in = true;
      if constexpr (MASKED) {
        const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
        in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      }
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc[e]) : 0.f;
      *reinterpret_cast<u32x2*>(y1s + (hy * Y1P + hx) * 32 + g * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    __syncthreads();

    // ---- B. t on the 18 x 18 inner pixels
    for (int mt = wave; mt < MT_B; mt += NW) {
      const int q = mt * 16 + r;
      const int qc = q < MPX ? q : MPX - 1;
      const int ty = qc / MW, tx = qc - ty * MW;
      const char* base = y1s + (ty * Y1P + tx) * 32;
      f32x4 acc = bias_a;
#pragma unroll
      for (int s = 0; s < 5; ++s) acc = mfma32(w_a[s], *reinterpret_cast<const u32x4*>(base + offB[s]), acc);
      bool in = true;
      if constexpr (MASKED) {
        const int iy = oy0 - 1 + ty, ix = ox0 - 1 + tx;
        in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      }
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc[e]) : 0.f;
      *reinterpret_cast<u32x2*>(ts + q * 32 + g * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    __syncthreads();
  };
  if (interior) stage_ab(std::false_type{});
  else stage_ab(std::true_type{});

  // ---- C + D. a wave owns TH / NW tile rows (one m-tile each)
#pragma unroll 1
  for (int rr = 0; rr < TH / NW; ++rr) {
    const int i = wave * (TH / NW) + rr;
    const char* base = ts + (i * MW + r) * 32;
    f32x4 acc = bias_b;
#pragma unroll
    for (int s = 0; s < 5; ++s) acc = mfma32(w_b[s], *reinterpret_cast<const u32x4*>(base + offC[s]), acc);
    const int cpx = (i + 2) * XW + r + 2;  // this pixel in the x halo tile
    const u32x2 y1c = *reinterpret_cast<const u32x2*>(y1s + ((i + 2) * Y1P + r + 2) * 32 + g * 8);
    const float bv0 = silu(acc[0]) + __uint_as_float(y1c[0] << 16);
    const float bv1 = silu(acc[1]) + __uint_as_float(y1c[0] & 0xFFFF0000u);
    const float bv2 = silu(acc[2]) + __uint_as_float(y1c[1] << 16);
    const float bv3 = silu(acc[3]) + __uint_as_float(y1c[1] & 0xFFFF0000u);
    const u32x2 bB = u32x2{pack_bf16x2(bv0, bv1), pack_bf16x2(bv2, bv3)};
    const f32x4 a0 = mfma32(w_y0, *reinterpret_cast<const u32x4*>(xs + cpx * 64 + g * 16), bias_y0);
    const u32x2 y0B = u32x2{pack_bf16x2(silu(a0[0]), silu(a0[1])), pack_bf16x2(silu(a0[2]), silu(a0[3]))};
    f32x4 o[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      o[nt] = mfma16(w_2[0][nt], y0B, bias_2[nt]);
      o[nt] = mfma16(w_2[1][nt], y1c, o[nt]);
      o[nt] = mfma16(w_2[2][nt], bB, o[nt]);
    }
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = silu(o[0][e]);
      v1[e] = silu(o[1][e]);
    }
    auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
    const int cb = 16 * (g & 1) + 8 * (g >> 1);
    const int oy = oy0 + i, ox = ox0 + r;
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<u32x4*>(p.y + ((((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
  }
}

// =====================================================================================================================
// C2f(64 -> 64, n = NB Bottlenecks of 32 channels) as one kernel: model.4 of yolov8n at 80 x 80 (NB = 2), model.2 of yolov8s
// at 160 x 160 (NB = 1).  Same plan as above at twice the width: a channel group of 32 is exactly one MFMA k-step, so a 3x3
// conv is 9 k-steps x 2 n-tiles per 16 pixels and cv2 is (2 + NB) k-steps x 4 n-tiles, one k-step per concatenated tensor.
// A workgroup (16 waves, one per CU: the tiles below fill the LDS) owns a 16 x 16 output tile; with R = 2 NB:
//   A. x halo tile (16 + 2R)^2 px x 128 B by LDS-DMA; y1 = SiLU(cv1 upper half) on every halo pixel, y0 = SiLU(cv1 lower
//      half) on the tile's own pixels; both ZERO outside the image where they feed a 3x3 conv;
//   B.. per Bottleneck: t = SiLU(conv3x3(prev)) on the ring R - 1, b = prev + SiLU(conv3x3(t)) on the ring R - 2, each as an
//      LDS tile of 64 B / pixel (XOR-swizzled 16-byte groups, conflict-free ds_read_b128); the x tile's space is reused for
//      the t tiles and the last b once cv1 is done;
//   F. out = SiLU(cv2 . [y0 | y1 | b1 (| b2)]) from the four tiles, 16-byte NHWC stores.
// Weights: a wave produces ONE 16-channel n-tile (wave & 1) of every 8th m-tile, so it needs 9 A fragments per 3x3 stage
// (36 VGPRs); they are loaded from L2 into registers one stage ahead, no stage waits on them and there is no barrier inside
// a stage - only one between stages.  (First form: 8 waves holding both n-tiles, 208 VGPRs, two waves per SIMD: 84 us for
// model.4 - every wave's MFMA -> SiLU -> store chain ran exposed; 16 waves at <= 128 VGPRs interleave four per SIMD.)
// Recompute on the rings (NB = 2): cv1 2.25x, the four 3x3 convs 1.89 / 1.56 / 1.27 / 1x of a 16 x 16 tile; in exchange the
// 26 + 13 + 13 + 13 + 13 MB of intermediates (written once, read up to three times) never leave the CU.
// =====================================================================================================================
struct C2f32Params {
  const char* x; char* y;
  const char *w1, *w2;
  const char* wm[4];     // m[0].cv1, m[0].cv2, m[1].cv1, m[1].cv2
  const float *b1, *b2;
  const float* bm[4];
  int N, H, W, ldx, ldy, tilesX, tilesY, shortcut;
  // CHUNKED form (cv1 with c1 = 64 * nch input channels, streamed in 64-channel chunks): the first upC channels of a pixel come from
  // pixel (y / 2, x / 2) of the half-resolution tensor `up` (a virtual nn.Upsample + Concat), the rest from x itself
  const char* up; int c1, upC, up_ld;
  int xcd;  // 1 = XCD-aware tile order (upa_xcd_tile)
};

namespace c2f32 {
using c2f::mfma32;
using c2f::silu;
constexpr int T = 16;
constexpr int NW = 16;  // waves per workgroup
__device__ __forceinline__ int swz64(int px) { return (px >> 1) & 3; }  // 64-byte pixel records: 4 groups of 16 B
// (An m-tile of a 3x3 stage that straddles two rows of its (pitch - 2)-wide output reads pixel indices that collide mod 8: with
// 64-byte records and four slots no swizzle of (pixel, row) removes that - searched exhaustively, tools/experiments/lds_swizzle_search.py -
// only a padded pitch does (upa_lds_pick_pitch: 30 / 28 / 26 for the 22 / 20 / 18-wide stages), which the 153 KB of tiles leave
// no room for.  PMC: 34 % of this kernel's LDS cycles are conflict cycles; it is VALU-bound on SiLU, not LDS-bound.)

// address of the 8 bytes holding channels 16j + 4g .. + 3 of pixel px in a 64 B / px tile
__device__ __forceinline__ int quad_addr(int px, int j, int g) { return px * 64 + (((2 * j + (g >> 1)) ^ swz64(px)) << 4) + (g & 1) * 8; }


// the nine A fragments (one per tap) of n-tile j of a 3x3 conv 32 -> 32: packed [tap][1 k-tile][2 n-tiles][lane][16 B]
__device__ __forceinline__ void load_w9(u32x4 (&w)[9], const char* src, int j, int lane) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) w[tap] = *reinterpret_cast<const u32x4*>(src + ((size_t)(tap * 2 + j) * 64 + lane) * 16);
}

// One 3x3 conv stage between two 64 B / px LDS tiles: dst (SDH x SD pixels, ring RD around the output tile) from src (two more
// rows and columns).
// A wave owns n-tile j (= wave & 1) of every 8th m-tile.  res: tile of side SR whose pixel (yy + OFF, xx + OFF) is added
// after the activation (the Bottleneck shortcut), or nullptr.
template <int SDH, int SD, int RD, int SR, int OFF>
__device__ __forceinline__ void conv3x3_stage(const char* src, char* dst, const u32x4 (&w)[9], const f32x4 bias, const char* res,
                                              int oy0, int ox0, int H, int W, int wave, int g, int r) {
  constexpr int SS = SD + 2, NPX = SDH * SD, NMT = (NPX + 15) / 16;  // dst: SDH rows x SD columns
  const int j = wave & 1;
  for (int mt = wave >> 1; mt < NMT; mt += NW / 2) {
    const int q = mt * 16 + r;
    const int qc = q < NPX ? q : NPX - 1;
    const int yy = qc / SD, xx = qc - yy * SD;
    const int sp = yy * SS + xx;
    f32x4 acc = bias;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int px = sp + (tap / 3) * SS + (tap % 3);
      acc = mfma32(w[tap], *reinterpret_cast<const u32x4*>(src + px * 64 + ((g ^ swz64(px)) << 4)), acc);
    }
    const int gy = oy0 - RD + yy, gx = ox0 - RD + xx;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc[e]) : 0.f;
    if (res) {  // the shortcut tensor is zero outside the image already
      const u32x2 rr = *reinterpret_cast<const u32x2*>(res + quad_addr((yy + OFF) * SR + xx + OFF, j, g));
      v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xFFFF0000u);
      v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xFFFF0000u);
    }
    if (q < NPX) *reinterpret_cast<u32x2*>(dst + quad_addr(q, j, g)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}
}  // namespace c2f32

// TH = output tile rows (16 columns always).  16 x 16 is the default; 10 rows balance the one-per-CU workgroups of yolov8n
// model.4 (1280 tiles = 5.0 rounds instead of 800 = 3.125) but measured no faster - kept as an experiment switch (host side).
template <int NB, int TH, bool CHUNKED = false>
__global__ __launch_bounds__(1024) void c2f32_fused_kernel(const C2f32Params p) {
  using namespace c2f32;
  constexpr int R = 2 * NB;
  constexpr int SX = T + 2 * R, SXH = TH + 2 * R;        // x / y1 tile: SXH rows x SX columns (24 | 20 wide)
  constexpr int XPX = SXH * SX;
  constexpr int XITEMS = XPX * 8;                        // 16-byte items of the x tile (a multiple of 64: whole waves)
  constexpr int XIT = (XITEMS + 1023) / 1024;
  constexpr int XREG = XPX * 128;                        // x region (reused)
  constexpr int Y1B = XPX * 64;
  constexpr int S_T1 = SX - 2, S_B1 = SX - 4;            // widths 22, 20 | 18, 16
  constexpr int H_T1 = SXH - 2, H_B1 = SXH - 4;
  constexpr int T1B = ((H_T1 * S_T1 + 15) / 16) * 16 * 64;
  constexpr int S_T2 = S_B1 - 2, H_T2 = H_B1 - 2;        // 18 wide (NB = 2)
  constexpr int T2B = ((H_T2 * S_T2 + 15) / 16) * 16 * 64;
  static_assert(XITEMS % 64 == 0 && XPX % 16 == 0, "the DMA guard must be wave-uniform, y1 in whole m-tiles");
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xs = sm;                                  // x; then t1 at 0, (NB = 2) t2 after it, the last b after that
  char* y1s = sm + XREG;
  char* y0s = y1s + Y1B;                          // TH x 16 px
  char* b1s = y0s + TH * T * 64;                  // NB = 2 only: b1 on ring 2; NB = 1 keeps b1 in the x region
  char* t1s = xs;
  char* t2s = xs + T1B;
  char* bls = NB == 2 ? xs + T1B + T2B : xs + T1B;  // last Bottleneck's output (TH x 16)
  static_assert(T1B + (NB == 2 ? T2B : 0) + TH * T * 64 <= XREG, "t / b tiles must fit the dead x region");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int j = wave & 1;  // the n-tile (of a 32-channel tensor) this wave produces
  const int tilesPerImg = p.tilesX * p.tilesY;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, tilesPerImg * p.N) : (int)blockIdx.x;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * TH, ox0 = txi * T;

  UPA_STAMP_AT(0);
  UPA_STAMP_HWID();
  u32x4 wA[9], wB[9];
  f32x4 bA, bB;
  if constexpr (!CHUNKED) {
    // ---- x halo tile: 128 B / px, 16-byte group cg of pixel px at slot cg ^ (px & 7)
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = it * 1024 + tid;
      if (it * 1024 + wave * 64 < XITEMS) {
        const int px = idx >> 3, slot = idx & 7;
        const int cg = slot ^ (px & 7);
        const int hy = px / SX, hx = px - hy * SX;
        const int iy = oy0 - R + hy, ix = ox0 - R + hx;
        const char* src = reinterpret_cast<const char*>(g_c2f_zero16);
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + cg * 8) * 2;
        __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(xs + (it * 1024 + wave * 64) * 16), 16, 0, 0);
      }
    }
    // cv1 (64 -> 64: [2 k-tiles][4 n-tiles]): this wave's n-tiles j (y0) and 2 + j (y1); the first 3x3 conv's fragments
    u32x4 w1f[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) w1f[kt][h2] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(kt * 4 + 2 * h2 + j) * 64 + lane) * 16);
    const f32x4 b1y0 = *reinterpret_cast<const f32x4*>(p.b1 + j * 16 + 4 * g);
    const f32x4 b1y1 = *reinterpret_cast<const f32x4*>(p.b1 + (2 + j) * 16 + 4 * g);
    load_w9(wA, p.wm[0], j, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    UPA_STAMP_AT(1);
    bA = *reinterpret_cast<const f32x4*>(p.bm[0] + j * 16 + 4 * g);
    bB = *reinterpret_cast<const f32x4*>(p.bm[1] + j * 16 + 4 * g);

    // ---- A. cv1: y1 on every halo pixel, y0 on the tile's own pixels
    for (int mt = wave >> 1; mt < XPX / 16; mt += NW / 2) {
      const int q = mt * 16 + r;
      f32x4 a = b1y1;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        a = mfma32(w1f[kt][1], *reinterpret_cast<const u32x4*>(xs + q * 128 + (((kt * 4 + g) ^ (q & 7)) << 4)), a);
      const int hy = q / SX, hx = q - hy * SX;
      const int gy = oy0 - R + hy, gx = ox0 - R + hx;
      const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = in ? silu(a[e]) : 0.f;
      *reinterpret_cast<u32x2*>(y1s + quad_addr(q, j, g)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    for (int i = wave >> 1; i < TH; i += NW / 2) {
      const int q = (i + R) * SX + R + r;
      f32x4 a = b1y0;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        a = mfma32(w1f[kt][0], *reinterpret_cast<const u32x4*>(xs + q * 128 + (((kt * 4 + g) ^ (q & 7)) << 4)), a);
      *reinterpret_cast<u32x2*>(y0s + quad_addr(i * T + r, j, g)) = u32x2{pack_bf16x2(silu(a[0]), silu(a[1])), pack_bf16x2(silu(a[2]), silu(a[3]))};
    }
  } else {
    // ---- cv1 over nch 64-channel chunks of the input (c1 = 64 nch; yolov8n model.15: 192 = upsampled 128 + skip 64).  Two chunk
    // buffers: xs and a second x-sized region behind the other tiles; the DMA of chunk c + 1 lands under the MFMAs of chunk c; a wave
    // keeps its accumulators (y1: every 8th m-tile of the halo'd tile, y0: two tile rows) across the chunks.
    char* xb2 = b1s + (NB == 2 ? ((H_B1 * S_B1 + 15) / 16) * 16 * 64 : 0);
    unsigned xoff[XIT], uoff[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = it * 1024 + tid;
      const int px = idx >> 3, slot = idx & 7;
      const int cg = slot ^ (px & 7);
      const int hy = px / SX, hx = px - hy * SX;
      const int iy = oy0 - R + hy, ix = ox0 - R + hx;
      const bool ok = idx < XITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      xoff[it] = ok ? (unsigned)((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + cg * 8) * 2u : 0xffffffffu;
      uoff[it] = (ok && p.up) ? (unsigned)((((size_t)n * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + (ix >> 1)) * (size_t)p.up_ld + cg * 8) * 2u : 0u;
    }
    const int nch = p.c1 >> 6;
    auto stage_chunk = [&](int c) __attribute__((always_inline)) {
      char* buf = (c & 1) ? xb2 : xs;
      const bool fromUp = c * 64 < p.upC;  // uniform: whole chunks come from one tensor (upC % 64 == 0)
      const char* base = fromUp ? p.up : p.x;
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        if (it * 1024 + wave * 64 < XITEMS) {
          const char* src = xoff[it] != 0xffffffffu ? base + (size_t)(fromUp ? uoff[it] : xoff[it]) + c * 128
                                                    : reinterpret_cast<const char*>(g_c2f_zero16);
          __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(buf + (it * 1024 + wave * 64) * 16), 16, 0, 0);
        }
      }
    };
    stage_chunk(0);
    auto load_w1 = [&](u32x4 (&a)[2][2], int c) __attribute__((always_inline)) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          a[kt][h2] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)((c * 2 + kt) * 4 + 2 * h2 + j) * 64 + lane) * 16);
    };
    u32x4 w1c[2][2], w1n[2][2];
    load_w1(w1c, 0);
    const f32x4 b1y0 = *reinterpret_cast<const f32x4*>(p.b1 + j * 16 + 4 * g);
    const f32x4 b1y1 = *reinterpret_cast<const f32x4*>(p.b1 + (2 + j) * 16 + 4 * g);
    bA = *reinterpret_cast<const f32x4*>(p.bm[0] + j * 16 + 4 * g);
    bB = *reinterpret_cast<const f32x4*>(p.bm[1] + j * 16 + 4 * g);
    constexpr int NMT1 = XPX / 16, A1 = (NMT1 + NW / 2 - 1) / (NW / 2), A0 = (TH + NW / 2 - 1) / (NW / 2);
    f32x4 acc1[A1], acc0[A0];
#pragma unroll
    for (int k = 0; k < A1; ++k) acc1[k] = b1y1;
#pragma unroll
    for (int k = 0; k < A0; ++k) acc0[k] = b1y0;
    for (int c = 0; c < nch; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of chunk c (and its fragments) has landed
      __syncthreads();                                  // ... everyone's; everyone is done with the other buffer
      if (c == 0) UPA_STAMP_AT(1);
      if (c + 1 < nch) {
        stage_chunk(c + 1);
        load_w1(w1n, c + 1);
      }
      const char* buf = (c & 1) ? xb2 : xs;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int k = 0; k < A1; ++k) {
          const int q = ((wave >> 1) + k * (NW / 2)) * 16 + r;
          if (q < XPX) acc1[k] = mfma32(w1c[kt][1], *reinterpret_cast<const u32x4*>(buf + q * 128 + (((kt * 4 + g) ^ (q & 7)) << 4)), acc1[k]);
        }
#pragma unroll
        for (int k = 0; k < A0; ++k) {
          const int i = (wave >> 1) + k * (NW / 2);
          const int q = (i + R) * SX + R + r;
          if (i < TH) acc0[k] = mfma32(w1c[kt][0], *reinterpret_cast<const u32x4*>(buf + q * 128 + (((kt * 4 + g) ^ (q & 7)) << 4)), acc0[k]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) w1c[kt][h2] = w1n[kt][h2];
    }
    load_w9(wA, p.wm[0], j, lane);  // the first 3x3 stage's fragments arrive under the SiLU / store epilogue below
#pragma unroll
    for (int k = 0; k < A1; ++k) {
      const int mt = (wave >> 1) + k * (NW / 2);
      if (mt >= NMT1) break;  // uniform
      const int q = mt * 16 + r;
      const int hy = q / SX, hx = q - hy * SX;
      const int gy = oy0 - R + hy, gx = ox0 - R + hx;
      const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc1[k][e]) : 0.f;
      *reinterpret_cast<u32x2*>(y1s + quad_addr(q, j, g)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
#pragma unroll
    for (int k = 0; k < A0; ++k) {
      const int i = (wave >> 1) + k * (NW / 2);
      if (i >= TH) break;  // uniform
      *reinterpret_cast<u32x2*>(y0s + quad_addr(i * T + r, j, g)) =
          u32x2{pack_bf16x2(silu(acc0[k][0]), silu(acc0[k][1])), pack_bf16x2(silu(acc0[k][2]), silu(acc0[k][3]))};
    }
  }
  load_w9(wB, p.wm[1], j, lane);  // every later stage's weights are fetched one stage ahead
  __syncthreads();  // y1 complete; x is dead
  UPA_STAMP_AT(2);

  // ---- B. t1 = SiLU(conv3x3(y1)) on ring R - 1 (into the x region)
  conv3x3_stage<H_T1, S_T1, R - 1, 1, 0>(y1s, t1s, wA, bA, nullptr, oy0, ox0, p.H, p.W, wave, g, r);
  __syncthreads();
  UPA_STAMP_AT(3);
  const char* sc1 = p.shortcut ? y1s : nullptr;
  constexpr int K2 = 2 + NB;
  u32x4 w2f[K2][2];  // cv2 [k-tile][4 n-tiles]: this wave's n-tiles 2j, 2j + 1
  if constexpr (NB == 2) {
    load_w9(wA, p.wm[2], j, lane);
    bA = *reinterpret_cast<const f32x4*>(p.bm[2] + j * 16 + 4 * g);
    // ---- C. b1 = y1 + SiLU(conv3x3(t1)) on ring 2
    conv3x3_stage<H_B1, S_B1, R - 2, SX, 2>(t1s, b1s, wB, bB, sc1, oy0, ox0, p.H, p.W, wave, g, r);
    __syncthreads();
    UPA_STAMP_AT(4);
    load_w9(wB, p.wm[3], j, lane);
    bB = *reinterpret_cast<const f32x4*>(p.bm[3] + j * 16 + 4 * g);
    // ---- D. t2 = SiLU(conv3x3(b1)) on ring 1
    conv3x3_stage<H_T2, S_T2, 1, 1, 0>(b1s, t2s, wA, bA, nullptr, oy0, ox0, p.H, p.W, wave, g, r);
    __syncthreads();
    UPA_STAMP_AT(5);
  }
#pragma unroll
  for (int kt = 0; kt < K2; ++kt)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) w2f[kt][h2] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)(kt * 4 + 2 * j + h2) * 64 + lane) * 16);
  const f32x4 b2v[2] = {*reinterpret_cast<const f32x4*>(p.b2 + (2 * j) * 16 + 4 * g), *reinterpret_cast<const f32x4*>(p.b2 + (2 * j + 1) * 16 + 4 * g)};
  if constexpr (NB == 2) {
    // ---- E. b2 = b1 + SiLU(conv3x3(t2)) on the tile
    conv3x3_stage<TH, T, 0, S_B1, 2>(t2s, bls, wB, bB, p.shortcut ? b1s : nullptr, oy0, ox0, p.H, p.W, wave, g, r);
  } else {
    // ---- C. b1 = y1 + SiLU(conv3x3(t1)) on the tile
    conv3x3_stage<TH, T, 0, SX, 2>(t1s, bls, wB, bB, sc1, oy0, ox0, p.H, p.W, wave, g, r);
  }
  __syncthreads();
  UPA_STAMP_AT(6);

  // ---- F. cv2 over [y0 | y1 | b1 (| b2)] of the tile's own pixels: a wave owns output channels 32j .. 32j + 31 of two rows
  for (int i = wave >> 1; i < TH; i += NW / 2) {
    const int d = i * T + r;
    const int qy1 = (i + R) * SX + R + r;
    u32x4 op[K2];
    op[0] = *reinterpret_cast<const u32x4*>(y0s + d * 64 + ((g ^ swz64(d)) << 4));
    op[1] = *reinterpret_cast<const u32x4*>(y1s + qy1 * 64 + ((g ^ swz64(qy1)) << 4));
    if constexpr (NB == 2) {
      const int qb1 = (i + 2) * S_B1 + 2 + r;
      op[2] = *reinterpret_cast<const u32x4*>(b1s + qb1 * 64 + ((g ^ swz64(qb1)) << 4));
      op[3] = *reinterpret_cast<const u32x4*>(bls + d * 64 + ((g ^ swz64(d)) << 4));
    } else {
      op[2] = *reinterpret_cast<const u32x4*>(bls + d * 64 + ((g ^ swz64(d)) << 4));
    }
    f32x4 o0 = b2v[0], o1 = b2v[1];
#pragma unroll
    for (int kt = 0; kt < K2; ++kt) {
      o0 = mfma32(w2f[kt][0], op[kt], o0);
      o1 = mfma32(w2f[kt][1], op[kt], o1);
    }
    const int oy = oy0 + i, ox = ox0 + r;
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = silu(o0[e]);
      v1[e] = silu(o1[e]);
    }
    auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
    const int cb = 16 * (2 * j + (g & 1)) + 8 * (g >> 1);
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<u32x4*>(p.y + ((((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
  }
  UPA_STAMP_AT(7);
#ifdef UPA_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UPA_STAMP_AT(8);
#endif
}

// the line-buffer form of the n = 2 block (c2f_stream.hip)
int upa_c2f32_stream_launch(const void* x, int n, int h, int w, int ldx, int shortcut, const void* w1, const float* b1,
                            const void* const* wm, const float* const* bm, const void* w2, const float* b2, void* y, int ldy,
                            const upa_opts* opts, hipStream_t s);
int upa_c2f32_stream1_launch(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int shortcut,
                             const void* w1, const float* b1, const void* const* wm, const float* const* bm, const void* w2, const float* b2,
                             void* y, int ldy, const upa_opts* opts, hipStream_t s);

int upa_c2f16_stream_launch(const void* x, int n, int h, int w, int ldx, const void* w1, const float* b1, const void* wa, const float* ba,
                            const void* wb, const float* bb, const void* w2, const float* b2, void* y, int ldy, const upa_opts* opts,
                            hipStream_t s);

// x: (n, h, w, c1) NHWC bf16 view; w1 / b1: cv1 (1x1, c1 -> 2c); wm[2i], wm[2i + 1] / bm[..]: Bottleneck i's two 3x3 convs
// (c -> c); w2 / b2: cv2 (1x1, (2 + nb) c -> c2) - all packed by upa_pack_conv_weight(bf16) with BN folded; y: (n, h, w, c2).
extern "C" int upa_c2f_fused(const void* x, int n, int h, int w, int c1, int ldx, int c, int nb, int shortcut, const void* w1,
                             const float* b1, const void* const* wm, const float* const* bm, const void* w2, const float* b2,
                             void* y, int c2, int ldy, int act, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && y && w1 && b1 && wm && bm && w2 && b2 && n > 0 && h > 0 && w > 0 && nb >= 1, "c2f_fused: bad args");
  const int off = UPA_OPT(opts, c2f);  // 1: never, 2: not the 16-wide, 3: not the 32-wide
  const bool f16 = c1 == 32 && c == 16 && c2 == 32 && nb == 1 && shortcut && off != 2;
  const bool f32 = c1 == 64 && c == 32 && c2 == 64 && (nb == 1 || nb == 2) && off != 3;
  if (off == 1 || dtype != UPA_BF16 || act != UPA_ACT_SILU || !(f16 || f32) || ldx % 8 != 0 || ldy % 8 != 0 ||
      ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0) {
    upa_set_error("c2f_fused: outside the fused forms (bf16, SiLU; C2f(32, 32, n=1, shortcut) or C2f(64, 64, n=1|2))");
    return UPA_EUNSUPPORTED;  // the caller runs the separate convolutions
  }
  for (int i = 0; i < 2 * nb; ++i) UPA_CHECK_ARG(wm[i] && bm[i], "c2f_fused: null Bottleneck weights");
  const int tx = cdiv(w, 16);
  int ty = cdiv(h, 16);
  long tiles = (long)tx * ty * n;
  UPA_CHECK_ARG(tiles < (1L << 31) / 2, "c2f_fused: too many tiles");
  hipStream_t s = (hipStream_t)stream;
  if (f16 && UPA_OPT(opts, c2f_stream) != 1 && UPA_OPT(opts, c2f16_waves) == 0) {  // the line-buffer form (csrc/c2f16_stream.hip); c2f_stream = 1 or c2f16_waves = 4 | 8: the 16 x 16 tile form below (A/B)
    const int rc = upa_c2f16_stream_launch(x, n, h, w, ldx, w1, b1, wm[0], bm[0], wm[1], bm[1], w2, b2, y, ldy, opts, s);
    if (rc != UPA_EUNSUPPORTED) {
      if (rc == UPA_OK) UPA_LAUNCH_CHECK();
      return rc;
    }
  }
  if (f16) {
    C2fParams p;
    p.x = (const char*)x; p.y = (char*)y;
    p.w1 = (const char*)w1; p.wa = (const char*)wm[0]; p.wb = (const char*)wm[1]; p.w2 = (const char*)w2;
    p.b1 = b1; p.ba = bm[0]; p.bb = bm[1]; p.b2 = b2;
    p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.tilesX = tx; p.tilesY = ty;
    p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
    const int nw = UPA_OPT(opts, c2f16_waves) == 8 ? 8 : 4;
    if (nw == 4) hipLaunchKernelGGL(c2f16_fused_kernel<4>, dim3((unsigned)tiles), dim3(256), c2f::LDS, s, p);
    else hipLaunchKernelGGL(c2f16_fused_kernel<8>, dim3((unsigned)tiles), dim3(512), c2f::LDS, s, p);
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
  C2f32Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 2 * nb; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.tilesX = tx; p.tilesY = ty; p.shortcut = shortcut ? 1 : 0;
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  // LDS of <NB, TH>: x region + y1 + y0 (+ b1 on ring 2 for NB = 2)
  auto lds_of = [](int nbk, int th) {
    const int r = 2 * nbk, sxw = 16 + 2 * r, sxh = th + 2 * r;
    size_t b = (size_t)sxh * sxw * (128 + 64) + (size_t)th * 16 * 64;
    if (nbk == 2) b += (size_t)(((sxh - 4) * (sxw - 4) + 15) / 16) * 16 * 64;
    return b;
  };
  if (nb == 2 && UPA_OPT(opts, c2f_stream) != 1 && UPA_OPT(opts, c2f32_th) == 0) {
    const int rc = upa_c2f32_stream_launch(x, n, h, w, ldx, shortcut, w1, b1, wm, bm, w2, b2, y, ldy, opts, s);
    if (rc == UPA_OK) { UPA_LAUNCH_CHECK(); return UPA_OK; }
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (nb == 1 && UPA_OPT(opts, c2f_stream) != 1) {
    const int rc = upa_c2f32_stream1_launch(x, n, h, w, 64, ldx, nullptr, 0, 0, shortcut, w1, b1, wm, bm, w2, b2, y, ldy, opts, s);
    if (rc == UPA_OK) { UPA_LAUNCH_CHECK(); return UPA_OK; }
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  if (nb == 2) {
    // 10-row tiles only on request (upa_opts.c2f32_th = 10): measured 80.1 against 83 us for model.4 alone and 0.636 against 0.626 ms
    // per step with four steps in flight - a tile's time is set by its six barrier-separated stages and the DMA wait more than
    // by its (m-tile, n-tile) unit count (222 against 354), so five balanced rounds do not beat 3.125 ragged ones
    const bool th10 = UPA_OPT(opts, c2f32_th) == 10;
    p.tilesY = th10 ? cdiv(h, 10) : cdiv(h, 16);
    tiles = (long)tx * p.tilesY * n;
    if (th10) {
      if (upa_full_lds<c2f32_fused_kernel<2, 10>>() != hipSuccess) return UPA_ELAUNCH;
      hipLaunchKernelGGL((c2f32_fused_kernel<2, 10>), dim3((unsigned)tiles), dim3(1024), lds_of(2, 10), s, p);
    } else {
      if (upa_full_lds<c2f32_fused_kernel<2, 16>>() != hipSuccess) return UPA_ELAUNCH;
      hipLaunchKernelGGL((c2f32_fused_kernel<2, 16>), dim3((unsigned)tiles), dim3(1024), lds_of(2, 16), s, p);
    }
  } else {
    if (upa_full_lds<c2f32_fused_kernel<1, 16>>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL((c2f32_fused_kernel<1, 16>), dim3((unsigned)tiles), dim3(1024), lds_of(1, 16), s, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// C2f(c1, 64, n = 1) with 32-channel halves and c1 = 64 * nch input channels (yolov8n model.15: 192 = nn.Upsample(128) + Concat(64)) as
// ONE launch: the c2f32 kernel with cv1 streamed over 64-channel chunks; the first up_c channels of a pixel are read from pixel (y / 2,
// x / 2) of `up` (n, h / 2, w / 2, up_c) - the upsampled tensor and the concat never exist (up = NULL: all channels from x).  Replaces
// upa_conv1x1_upcat + upa_bottleneck_pair_cv2 for that block.  Arguments as upa_c2f64_fused.
extern "C" int upa_c2f32_up_fused(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int nb,
                                  int shortcut, const void* w1, const float* b1, const void* const* wm, const float* const* bm,
                                  const void* w2, const float* b2, void* y, int c2, int ldy, int act, int dtype, const upa_opts* opts,
                                  void* stream) {
  UPA_CHECK_ARG(x && y && w1 && b1 && wm && bm && w2 && b2 && n > 0 && h > 0 && w > 0, "c2f32_up_fused: bad args");
  const int off = UPA_OPT(opts, c2f);  // 1: never, 3: not the 32-wide forms
  if (off == 1 || off == 3 || UPA_OPT(opts, no_c2f32_up) || dtype != UPA_BF16 || act != UPA_ACT_SILU || nb != 1 || c2 != 64 || c1 % 64 != 0 || c1 < 128 || c1 > 512 ||
      ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 ||
      (up && (up_c % 64 != 0 || up_c <= 0 || up_c > c1 || up_ld % 8 != 0 || (h & 1) || (w & 1) || ((uintptr_t)up % 16) != 0)) ||
      (long)n * h * w * (long)(ldx > up_ld ? ldx : up_ld) * 2 >= (1L << 32)) {
    upa_set_error("c2f32_up_fused: outside the fused form (bf16, SiLU; C2f(64 k, 64, n = 1), even map for the half-resolution source)");
    return UPA_EUNSUPPORTED;
  }
  for (int i = 0; i < 2; ++i) UPA_CHECK_ARG(wm[i] && bm[i], "c2f32_up_fused: null Bottleneck weights");
  if (UPA_OPT(opts, c2f_stream) != 1) {  // the line-buffer form (c2f_stream.hip) up to 192 input channels
    const int rc = upa_c2f32_stream1_launch(x, n, h, w, c1, ldx, up, up_c, up_ld, shortcut, w1, b1, wm, bm, w2, b2, y, ldy, opts, (hipStream_t)stream);
    if (rc == UPA_OK) { UPA_LAUNCH_CHECK(); return UPA_OK; }
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  C2f32Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 2; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.tilesX = cdiv(w, 16); p.tilesY = cdiv(h, 16); p.shortcut = shortcut ? 1 : 0;
  p.up = (const char*)up; p.c1 = c1; p.upC = up ? up_c : 0; p.up_ld = up_ld;
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long tiles = (long)p.tilesX * p.tilesY * n;
  UPA_CHECK_ARG(tiles < (1L << 31) / 2, "c2f32_up_fused: too many tiles");
  // LDS of <1, 16>: x region + y1 + y0, + the second chunk buffer
  const size_t lds = (size_t)20 * 20 * (128 + 64) + (size_t)16 * 16 * 64 + (size_t)20 * 20 * 128;
  if (upa_full_lds<c2f32_fused_kernel<1, 16, true>>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL((c2f32_fused_kernel<1, 16, true>), dim3((unsigned)tiles), dim3(1024), lds, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

