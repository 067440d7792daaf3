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

#include "common.h"

typedef __attribute__((address_space(1))) const void* cgptr_t;
typedef __attribute__((address_space(3))) void* clptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __attribute__((aligned(16))) unsigned g_c2f_zero16[4] = {0u, 0u, 0u, 0u};

struct C2fParams {
  const char* x; char* y;
  const char *w1, *wa, *wb, *w2;
  const float *b1, *ba, *bb, *b2;
  int N, H, W, ldx, ldy, tilesX, tilesY;
};

namespace c2f {
constexpr int TH = 16, TW = 16;
constexpr int XH = TH + 4, XW = TW + 4;      // x / y1 halo tile 20 x 20
constexpr int MH = TH + 2, MW = TW + 2;      // t tile 18 x 18
constexpr int XPX = XH * XW, MPX = MH * MW;  // 400, 324
constexpr int XITEMS = XPX * 4;              // 16-byte items of the x tile
constexpr int XIT = (XITEMS + 255) / 256;    // DMA rounds of 256 lanes
constexpr int XS_BYTES = XIT * 256 * 16;     // 28672
constexpr int Y1_BYTES = XPX * 32;           // 12800
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

__global__ __launch_bounds__(256) void c2f16_fused_kernel(const C2fParams p) {
  using namespace c2f;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xs = sm;
  char* y1s = sm + XS_BYTES;
  char* ts = y1s + Y1_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * TH, ox0 = txi * TW;

  // ---- 0. x halo tile
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int item = it * 256 + tid;
    const int px = item >> 2, slot = item & 3;
    const int hy = px / XW, hx = px - hy * XW;
    const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
    const char* src = reinterpret_cast<const char*>(g_c2f_zero16);
    if (item < XITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      src = p.x + (((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx * 2 + slot * 16;
    __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(xs + (it * 256 + wave * 64) * 16), 16, 0, 0);
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
    offB[s] = (kh * XW + kw) * 32 + (g & 1) * 16;  // into the y1 tile
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

  // ---- A. y1 on the 400 halo pixels
  for (int mt = wave; mt < XPX / 16; mt += 4) {
    const int q = mt * 16 + r;
    const u32x4 b = *reinterpret_cast<const u32x4*>(xs + q * 64 + g * 16);
    const f32x4 acc = mfma32(w_y1, b, bias_y1);
    const int hy = q / XW, hx = q - hy * XW;
    const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
    const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc[e]) : 0.f;
    *reinterpret_cast<u32x2*>(y1s + q * 32 + g * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
  __syncthreads();

  // ---- B. t on the 18 x 18 inner pixels
  for (int mt = wave; mt < MT_B; mt += 4) {
    const int q = mt * 16 + r;
    const int qc = q < MPX ? q : MPX - 1;
    const int ty = qc / MW, tx = qc - ty * MW;
    const char* base = y1s + (ty * XW + tx) * 32;
    f32x4 acc = bias_a;
#pragma unroll
    for (int s = 0; s < 5; ++s) acc = mfma32(w_a[s], *reinterpret_cast<const u32x4*>(base + offB[s]), acc);
    const int iy = oy0 - 1 + ty, ix = ox0 - 1 + tx;
    const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc[e]) : 0.f;
    *reinterpret_cast<u32x2*>(ts + q * 32 + g * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
  __syncthreads();

  // ---- C + D. wave w owns tile rows 4w .. 4w + 3 (one m-tile each)
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int i = wave * 4 + rr;
    const char* base = ts + (i * MW + r) * 32;
    f32x4 acc = bias_b;
#pragma unroll
    for (int s = 0; s < 5; ++s) acc = mfma32(w_b[s], *reinterpret_cast<const u32x4*>(base + offC[s]), acc);
    const int cpx = (i + 2) * XW + r + 2;  // this pixel in the halo tiles
    const u32x2 y1c = *reinterpret_cast<const u32x2*>(y1s + cpx * 32 + g * 8);
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

// x: (n, h, w, 32) NHWC bf16 view; w1 / b1: cv1 (1x1, 32 -> 32); wa / ba, wb / bb: the Bottleneck's two 3x3 convs (16 -> 16);
// w2 / b2: cv2 (1x1, 48 -> 32) - all packed by upa_pack_conv_weight(bf16) with BN folded; y: (n, h, w, 32) view.
extern "C" int upa_c2f_fused(const void* x, int n, int h, int w, int c1, int ldx, int c, int nb, int shortcut, const void* w1,
                             const float* b1, const void* wa, const float* ba, const void* wb, const float* bb, const void* w2,
                             const float* b2, void* y, int c2, int ldy, int act, int dtype, void* stream) {
  UPA_CHECK_ARG(x && y && w1 && b1 && wa && ba && wb && bb && w2 && b2 && n > 0 && h > 0 && w > 0, "c2f_fused: bad args");
  static const int off = getenv("UPA_NO_C2F") ? atoi(getenv("UPA_NO_C2F")) : 0;
  if (off || dtype != UPA_BF16 || act != UPA_ACT_SILU || c1 != 32 || c != 16 || c2 != 32 || nb != 1 || !shortcut || ldx % 8 != 0 ||
      ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0) {
    upa_set_error("c2f_fused: outside the fused form (bf16, SiLU, C2f(32, 32, n = 1, shortcut))");
    return UPA_EUNSUPPORTED;  // the caller runs the four convolutions
  }
  C2fParams p;
  p.x = (const char*)x; p.y = (char*)y;
  p.w1 = (const char*)w1; p.wa = (const char*)wa; p.wb = (const char*)wb; p.w2 = (const char*)w2;
  p.b1 = b1; p.ba = ba; p.bb = bb; p.b2 = b2;
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy;
  p.tilesX = cdiv(w, c2f::TW); p.tilesY = cdiv(h, c2f::TH);
  const long tiles = (long)p.tilesX * p.tilesY * n;
  UPA_CHECK_ARG(tiles < (1L << 31), "c2f_fused: too many tiles");
  hipLaunchKernelGGL(c2f16_fused_kernel, dim3((unsigned)tiles), dim3(256), c2f::LDS, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
