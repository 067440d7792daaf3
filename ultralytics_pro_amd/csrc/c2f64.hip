// C2f with 64-channel halves as ONE kernel (bf16): cv1 (1x1, c1 -> 2 x 64) -> NB x Bottleneck(3x3 64 -> 64, 3x3 64 -> 64 [+ input])
// -> cv2 (1x1 over cat(y0, y1, b1 [, b2]) = (2 + NB) x 64 -> c2 = 128), every conv with BN folded and SiLU.
//   C2f.forward        ultralytics/nn/modules/block.py:457-488   y = list(cv1(x).chunk(2, 1)); y.extend(m(y[-1]) ...); cv2(cat(y, 1))
//   Bottleneck.forward ultralytics/nn/modules/block.py:644-668   x + cv2(cv1(x))
// These are the 40 x 40 blocks of yolov8n - model.6 = C2f(128, 128, n = 2), model.12 = C2f(384, 128, n = 1) behind
// Upsample + Concat, model.18 = C2f(192, 128, n = 1) - six / four / four launches of 11-15 us each, every one of them a single
// round of 100-400 workgroups that is launch ramp, DMA latency, nine barrier-separated taps and an epilogue in sequence
// (DESIGN section 8).  Fused, a workgroup of 8 waves owns a TH x TW output tile of one image with ALL channels and the
// intermediates only ever exist as LDS tiles of 128 B per pixel (same plan as c2f_fused.hip at twice the width):
//   A. cv1: the input halo tile ((TH + 2R) x (TW + 2R) pixels, R = 2 NB) streams through LDS in 64-channel chunks by LDS-DMA,
//      double buffered between the region that later holds t1 and the region that later holds y1 itself, while the accumulators
//      of every (m-tile, n-tile) unit stay in registers across the chunks; the first up_c channels of a pixel may come from a
//      half-resolution tensor at (y / 2, x / 2) - the virtual Upsample + Concat of upa_conv1x1_upcat, yolov8.yaml rows 10-15;
//      y1 = SiLU(upper half) on every halo pixel (ZERO outside the image: it is the 3x3 conv's padding), y0 on the tile's own;
//   B. per Bottleneck: t = SiLU(conv3x3(prev)) on the ring R - 1, b = prev + SiLU(conv3x3(t)) on the ring R - 2;
//   C. out = SiLU(cv2 . [y0 | y1 | b1 (| b2)]) on the tile, 16-byte NHWC stores.
// A wave owns ONE 16-channel n-tile (wave & 3) of every second m-tile, so the 18 A fragments of a 3x3 conv (9 taps x 2 k-tiles)
// live in 72 VGPRs and the next stage's are fetched from L2 while this one runs; one barrier per stage.
// Tiles: NB = 2 -> 10 x 10 (40 x 40 maps: 16 tiles per image, no ragged edge; LDS 137 KB), NB = 1 -> 10 x 20 (8 tiles per image:
// 256 workgroups at batch 32 = one round of the chip; LDS 149 KB).  Rounding points (bf16 y0, y1, t, b, out; f32 accumulation and
// f32 residual add) are those of the separate launches.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
UPA_STAMP_DEFINE(c2f64)

typedef __attribute__((address_space(1))) const void* c6gptr_t;
typedef __attribute__((address_space(3))) void* c6lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_c2f64_zero16[4] = {0u, 0u, 0u, 0u};

struct C2f64Params {
  const char* x; char* y;
  const char* up;        // half-resolution tensor holding the first upC channels of every pixel at (y / 2, x / 2), or nullptr
  const char *w1, *w2;
  const char* wm[4];     // m[0].cv1, m[0].cv2, m[1].cv1, m[1].cv2
  const float *b1, *b2;
  const float* bm[4];
  int N, H, W, c1, ldx, ldy, upC, up_ld, tilesX, tilesY, shortcut;
};

namespace c2f64 {
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
// 128-byte pixel records, 16-byte group cg of pixel px at slot cg ^ (px & 7)
__device__ __forceinline__ int rec_addr(int px, int cg) { return px * 128 + ((cg ^ (px & 7)) << 4); }
// the 8 bytes holding channels 16 j + 4 g .. + 3 of pixel px
__device__ __forceinline__ int quad_addr(int px, int j, int g) { return px * 128 + (((2 * j + (g >> 1)) ^ (px & 7)) << 4) + (g & 1) * 8; }
constexpr int pad16(int px) { return (px + 15) / 16 * 16; }

// the 18 A fragments (tap x k-tile) of n-tile j of a 3x3 conv 64 -> 64: packed [tap][2 k-tiles][4 n-tiles][lane][16 B]
__device__ __forceinline__ void load_w18(u32x4 (&w)[18], const char* src, int j, int lane) {
#pragma unroll
  for (int f = 0; f < 18; ++f) w[f] = *reinterpret_cast<const u32x4*>(src + ((size_t)(f * 4 + j) * 64 + lane) * 16);
}

// One 3x3 conv stage between two LDS tiles: dst (DH x DW pixels, ring RD around the output tile) from src (two more rows / columns).
// A wave owns n-tile j of every second m-tile (mg = wave >> 2).  res: tile of pitch SR whose pixel (yy + OFF, xx + OFF) is added after
// the activation (the Bottleneck shortcut), or nullptr.
// SP / DP: pixel pitches of src / dst (>= DW + 2 / DW).
template <int DH, int DW, int RD, int SP, int DP, int SR, int OFF>
__device__ __forceinline__ void conv3x3_stage(const char* src, char* dst, const u32x4 (&w)[18], const f32x4 bias, const char* res,
                                              int oy0, int ox0, int H, int W, int j, int mg, int g, int r) {
  constexpr int NPX = DH * DW, NMT = (NPX + 15) / 16;
  for (int mt = mg; mt < NMT; mt += 2) {
    const int q = mt * 16 + r;
    const int qc = q < NPX ? q : NPX - 1;
    const int yy = qc / DW, xx = qc - yy * DW;
    const int sp = yy * SP + xx;
    f32x4 acc = bias;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int px = sp + (tap / 3) * SP + (tap % 3);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        acc = mfma32(w[tap * 2 + kt], *reinterpret_cast<const u32x4*>(src + rec_addr(px, kt * 4 + g)), acc);
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
    if (q < NPX) *reinterpret_cast<u32x2*>(dst + quad_addr(yy * DP + xx, j, g)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}
}  // namespace c2f64

template <int NB, int TH, int TW>
__global__ __launch_bounds__(512, 2) void c2f64_fused_kernel(const C2f64Params p) {
  using namespace c2f64;
  constexpr int R = 2 * NB;
  constexpr int SXH = TH + 2 * R, SXW = TW + 2 * R, XPX = SXH * SXW;   // x / y1 halo tile
  constexpr int XIT = XPX * 8;                                         // 16-byte items of one 64-channel chunk
  constexpr int NPASS = (XIT + 511) / 512;                             // DMA passes of the 512 threads
  constexpr int CHB = NPASS * 512 * 16;                                // chunk buffer = region size (bytes)
  constexpr int NMT1 = (XPX + 15) / 16;                                // m-tiles of y1
  constexpr int TPX = TH * TW, NMT0 = (TPX + 15) / 16;                 // the tile's own pixels
  constexpr int H1 = SXH - 2, W1 = SXW - 2, H2 = SXH - 4, W2 = SXW - 4, H3 = SXH - 6, W3 = SXW - 6;
  // Pixel pitches of the tiles a 3x3 stage reads: its m-tiles enumerate a (pitch - 2)-wide output row-major, and one that straddles
  // two rows reads pixel indices that collide mod 8 (= LDS bank conflicts on ds_read_b128) unless the row step suits the width:
  // upa_lds_pick_pitch (common.h).  These stages issue one fragment read per MFMA - they are LDS-read-bound - and the unpadded
  // pitches cost the 14- / 12- / 10-wide stages 1.7x the LDS cycles.
  constexpr int PY1 = upa_lds_pick_pitch(SXW, W1, H1 * W1, 1);                       // y1, read by the t1 stage
  constexpr int PT1 = upa_lds_pick_pitch(W1, NB == 2 ? W2 : TW, NB == 2 ? H2 * W2 : TPX, 1);  // t1, read by the b1 stage
  constexpr int PB1 = NB == 2 ? upa_lds_pick_pitch(W2, W3, H3 * W3, 1) : 0;          // b1 (NB = 2), read by the t2 stage
  constexpr int PT2 = NB == 2 ? upa_lds_pick_pitch(W3, TW, TPX, 1) : 0;              // t2 (NB = 2), read by the b2 stage
  constexpr int Y1B = (SXH * PY1 * 128 + 1023) / 1024 * 1024 > CHB ? (SXH * PY1 * 128 + 1023) / 1024 * 1024 : CHB;  // y1 region (holds chunk buffer B)
  constexpr int T1B = pad16(H1 * PT1) * 128, T2B = NB == 2 ? pad16(H3 * PT2) * 128 : 0;
  constexpr int B1B = NB == 2 ? pad16(H2 * PB1) * 128 : 0;
  constexpr int TLB = pad16(TPX) * 128;                                // a TH x TW tile (y0, last b)
  static_assert(T1B <= CHB && T2B + TLB <= CHB, "t tiles and the last b must fit the dead chunk buffer");
  static_assert(NMT1 <= 22 && NMT0 <= 14, "accumulator arrays");
  constexpr int A1 = (NMT1 + 1) / 2, A0 = (NMT0 + 1) / 2;              // m-tiles per wave (every second one)
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* y1s = sm;                 // chunk buffer B during cv1, then y1
  char* xr = sm + Y1B;            // chunk buffer A during cv1, then t1, then (NB = 2) t2 | b2
  char* b1s = xr + CHB;           // NB = 2: b1 on ring 2;  NB = 1: the Bottleneck's output on the tile
  char* y0s = b1s + (NB == 2 ? B1B : TLB);
  char* t1s = xr;
  char* t2s = xr;
  char* bls = NB == 2 ? xr + T2B : b1s;  // last Bottleneck's output (TH x TW)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int j = wave & 3, mg = wave >> 2;
  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * TH, ox0 = txi * TW;

  // ---- this thread's items of a chunk: pixel offsets into x and into the half-resolution tensor (independent of the chunk)
  unsigned xoff[NPASS], uoff[NPASS];
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    const int idx = it * 512 + tid;
    const int px = idx >> 3, slot = idx & 7;
    const int cg = slot ^ (px & 7);
    const int hy = px / SXW, hx = px - hy * SXW;
    const int iy = oy0 - R + hy, ix = ox0 - R + hx;
    const bool ok = px < XPX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    xoff[it] = ok ? (unsigned)((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + cg * 8) * 2u : 0xffffffffu;
    uoff[it] = (ok && p.up) ? (unsigned)((((size_t)n * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + (ix >> 1)) * (size_t)p.up_ld + cg * 8) * 2u : 0u;
  }
  const int nch = p.c1 >> 6;
  auto stage_chunk = [&](int c) __attribute__((always_inline)) {
    char* buf = ((nch - 1 - c) & 1) ? y1s : xr;   // the LAST chunk lands in xr: y1 can be written while it is still being read
    const bool fromUp = c * 64 < p.upC;            // uniform: whole chunks come from one tensor (upC % 64 == 0)
    const char* base = fromUp ? p.up : p.x;
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const char* src = xoff[it] != 0xffffffffu ? base + (size_t)(fromUp ? uoff[it] : xoff[it]) + c * 128
                                                : reinterpret_cast<const char*>(g_c2f64_zero16);
      __builtin_amdgcn_global_load_lds((c6gptr_t)src, (c6lptr_t)(buf + (it * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };
  UPA_STAMP_AT(0);
  UPA_STAMP_HWID();
  stage_chunk(0);

  // cv1 fragments: packed [k-tile][8 n-tiles][lane][16 B]; this wave's n-tiles j (y0) and 4 + j (y1)
  auto load_w1 = [&](u32x4 (&a)[4], int c) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
        a[kt * 2 + h2] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)((c * 2 + kt) * 8 + 4 * h2 + j) * 64 + lane) * 16);
  };
  u32x4 w1c[4], w1n[4];
  load_w1(w1c, 0);
  const f32x4 b1y0 = *reinterpret_cast<const f32x4*>(p.b1 + j * 16 + 4 * g);
  const f32x4 b1y1 = *reinterpret_cast<const f32x4*>(p.b1 + (4 + j) * 16 + 4 * g);
  f32x4 acc1[A1], acc0[A0];
#pragma unroll
  for (int i = 0; i < A1; ++i) acc1[i] = b1y1;
#pragma unroll
  for (int i = 0; i < A0; ++i) acc0[i] = b1y0;
  // halo index of this lane's pixel of the wave's y0 m-tiles
  int p0[A0];
#pragma unroll
  for (int i = 0; i < A0; ++i) {
    const int q = (mg + 2 * i) * 16 + r;
    const int qc = q < TPX ? q : TPX - 1;
    const int ty = qc / TW, tx = qc - ty * TW;
    p0[i] = (ty + R) * SXW + tx + R;
  }

  // ---- A. cv1 over the chunks
  for (int c = 0; c < nch; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of chunk c (and its fragments) has landed
    __syncthreads();                                  // ... everyone's; everyone is done with the other buffer
    if (c == 0) UPA_STAMP_AT(1);
    if (c + 1 < nch) {
      stage_chunk(c + 1);
      load_w1(w1n, c + 1);
    }
    const char* buf = ((nch - 1 - c) & 1) ? y1s : xr;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < A1; ++i) {
        const int mt = mg + 2 * i;
        if (mt < NMT1) acc1[i] = mfma32(w1c[kt * 2 + 1], *reinterpret_cast<const u32x4*>(buf + rec_addr(mt * 16 + r, kt * 4 + g)), acc1[i]);
      }
#pragma unroll
      for (int i = 0; i < A0; ++i) {
        const int mt = mg + 2 * i;
        if (mt < NMT0) acc0[i] = mfma32(w1c[kt * 2], *reinterpret_cast<const u32x4*>(buf + rec_addr(p0[i], kt * 4 + g)), acc0[i]);
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) w1c[f] = w1n[f];
  }
  u32x4 wA[18], wB[18];
  load_w18(wA, p.wm[0], j, lane);
  // y1 (every halo pixel, zero outside the image) and y0 (the tile's own pixels): y1s is free - the last chunk sits in xr
#pragma unroll
  for (int i = 0; i < A1; ++i) {
    const int mt = mg + 2 * i;
    if (mt >= NMT1) continue;
    const int q = mt * 16 + r;
    const int hy = q / SXW, hx = q - hy * SXW;
    const int gy = oy0 - R + hy, gx = ox0 - R + hx;
    const bool in = q < XPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = in ? silu(acc1[i][e]) : 0.f;
    if (q < XPX) *reinterpret_cast<u32x2*>(y1s + quad_addr(hy * PY1 + hx, j, g)) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
#pragma unroll
  for (int i = 0; i < A0; ++i) {
    const int mt = mg + 2 * i;
    if (mt >= NMT0) continue;
    const int q = mt * 16 + r;
    *reinterpret_cast<u32x2*>(y0s + quad_addr(q, j, g)) =
        u32x2{pack_bf16x2(silu(acc0[i][0]), silu(acc0[i][1])), pack_bf16x2(silu(acc0[i][2]), silu(acc0[i][3]))};
  }
  load_w18(wB, p.wm[1], j, lane);  // every later stage's weights are fetched one stage ahead
  f32x4 bA = *reinterpret_cast<const f32x4*>(p.bm[0] + j * 16 + 4 * g);
  f32x4 bB = *reinterpret_cast<const f32x4*>(p.bm[1] + j * 16 + 4 * g);
  __syncthreads();  // y1, y0 complete; the chunk buffers are dead
  UPA_STAMP_AT(2);

  // ---- B. t1 = SiLU(conv3x3(y1)) on ring R - 1 (into the chunk region)
  conv3x3_stage<H1, W1, R - 1, PY1, PT1, 1, 0>(y1s, t1s, wA, bA, nullptr, oy0, ox0, p.H, p.W, j, mg, g, r);
  __syncthreads();
  UPA_STAMP_AT(3);
  const char* sc1 = p.shortcut ? y1s : nullptr;
  if constexpr (NB == 2) {
    load_w18(wA, p.wm[2], j, lane);
    bA = *reinterpret_cast<const f32x4*>(p.bm[2] + j * 16 + 4 * g);
    // b1 = y1 + SiLU(conv3x3(t1)) on ring 2
    conv3x3_stage<H2, W2, R - 2, PT1, PB1, PY1, 2>(t1s, b1s, wB, bB, sc1, oy0, ox0, p.H, p.W, j, mg, g, r);
    __syncthreads();
    UPA_STAMP_AT(4);
    load_w18(wB, p.wm[3], j, lane);
    bB = *reinterpret_cast<const f32x4*>(p.bm[3] + j * 16 + 4 * g);
    // t2 = SiLU(conv3x3(b1)) on ring 1
    conv3x3_stage<H3, W3, 1, PB1, PT2, 1, 0>(b1s, t2s, wA, bA, nullptr, oy0, ox0, p.H, p.W, j, mg, g, r);
    __syncthreads();
    UPA_STAMP_AT(5);
  }
  // cv2 fragments (into the registers of the finished stage's weights): packed [k-tile][8 n-tiles][lane][16 B]; this wave's
  // n-tiles 2 j, 2 j + 1 over all K2 = 2 (2 + NB) k-tiles
  constexpr int K2 = 2 * (2 + NB);
  static_assert(2 * K2 <= 18, "cv2 fragments reuse the 3x3 fragment registers");
#pragma unroll
  for (int kt = 0; kt < K2; ++kt)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) wA[kt * 2 + h2] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)(kt * 8 + 2 * j + h2) * 64 + lane) * 16);
  const f32x4 b2v[2] = {*reinterpret_cast<const f32x4*>(p.b2 + (2 * j) * 16 + 4 * g), *reinterpret_cast<const f32x4*>(p.b2 + (2 * j + 1) * 16 + 4 * g)};
  if constexpr (NB == 2) {
    // b2 = b1 + SiLU(conv3x3(t2)) on the tile
    conv3x3_stage<TH, TW, 0, PT2, TW, PB1, 2>(t2s, bls, wB, bB, p.shortcut ? b1s : nullptr, oy0, ox0, p.H, p.W, j, mg, g, r);
  } else {
    // b1 = y1 + SiLU(conv3x3(t1)) on the tile
    conv3x3_stage<TH, TW, 0, PT1, TW, PY1, 2>(t1s, bls, wB, bB, sc1, oy0, ox0, p.H, p.W, j, mg, g, r);
  }
  __syncthreads();
  UPA_STAMP_AT(6);

  // ---- C. cv2 over [y0 | y1 | b1 (| b2)] of the tile's own pixels: a wave owns output channels 32 j .. 32 j + 31 of every second m-tile
  for (int mt = mg; mt < NMT0; mt += 2) {
    const int q = mt * 16 + r;
    const int qc = q < TPX ? q : TPX - 1;
    const int ty = qc / TW, tx = qc - ty * TW;
    const int qy1 = (ty + R) * PY1 + tx + R;
    f32x4 o0 = b2v[0], o1 = b2v[1];
#pragma unroll
    for (int kt = 0; kt < K2; ++kt) {
      const int t = kt >> 1, cg = (kt & 1) * 4 + g;  // tensor, 16-byte group of its pixel record
      u32x4 b;
      if (t == 0) b = *reinterpret_cast<const u32x4*>(y0s + rec_addr(qc, cg));
      else if (t == 1) b = *reinterpret_cast<const u32x4*>(y1s + rec_addr(qy1, cg));
      else if (NB == 2 && t == 2) b = *reinterpret_cast<const u32x4*>(b1s + rec_addr((ty + 2) * PB1 + tx + 2, cg));
      else b = *reinterpret_cast<const u32x4*>(bls + rec_addr(qc, cg));
      o0 = mfma32(wA[kt * 2], b, o0);
      o1 = mfma32(wA[kt * 2 + 1], b, o1);
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = silu(o0[e]);
      v1[e] = silu(o1[e]);
    }
    auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
    const int cb = 16 * (2 * j + (g & 1)) + 8 * (g >> 1);
    if (q < TPX && oy < p.H && ox < p.W)
      *reinterpret_cast<u32x4*>(p.y + ((((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.ldy + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
  }
  UPA_STAMP_AT(7);
#ifdef UPA_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UPA_STAMP_AT(8);
#endif
}

namespace {
template <int NB, int TH, int TW>
constexpr size_t c2f64_lds() {  // the kernel's region sizes
  constexpr int R = 2 * NB, SXH = TH + 2 * R, SXW = TW + 2 * R, XIT = SXH * SXW * 8, NPASS = (XIT + 511) / 512;
  constexpr int W1 = SXW - 2, H1 = SXH - 2, W2 = SXW - 4, H2 = SXH - 4, W3 = SXW - 6, H3 = SXH - 6;
  constexpr size_t CHB = (size_t)NPASS * 512 * 16;
  constexpr int PY1 = upa_lds_pick_pitch(SXW, W1, H1 * W1, 1);
  constexpr int PB1 = NB == 2 ? upa_lds_pick_pitch(W2, W3, H3 * W3, 1) : 0;
  constexpr size_t Y1B = ((size_t)SXH * PY1 * 128 + 1023) / 1024 * 1024 > CHB ? ((size_t)SXH * PY1 * 128 + 1023) / 1024 * 1024 : CHB;
  constexpr size_t TLB = (size_t)c2f64::pad16(TH * TW) * 128;
  constexpr size_t B1B = NB == 2 ? (size_t)c2f64::pad16(H2 * PB1) * 128 : 0;
  return Y1B + CHB + (NB == 2 ? B1B : TLB) + TLB;
}
template <int NB, int TH, int TW>
int c2f64_launch(C2f64Params& p, hipStream_t s) {
  p.tilesX = cdiv(p.W, TW);
  p.tilesY = cdiv(p.H, TH);
  const long tiles = (long)p.tilesX * p.tilesY * p.N;
  if (tiles >= (1L << 31)) return UPA_EUNSUPPORTED;
  if (hipError_t e = upa_full_lds<c2f64_fused_kernel<NB, TH, TW>>(); e != hipSuccess) return UPA_ELAUNCH;
  const size_t lds = c2f64_lds<NB, TH, TW>();
  hipLaunchKernelGGL((c2f64_fused_kernel<NB, TH, TW>), dim3((unsigned)tiles), dim3(512), lds, s, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
}  // namespace

// x: (n, h, w, c1) NHWC bf16 view, c1 % 64 == 0; up (may be NULL): (n, h / 2, w / 2, up_c) holding the first up_c (% 64 == 0) channels of
// every pixel at half resolution (the Upsample(2x nearest) + Concat in front of the block, never materialised); w1 / b1: cv1 (1x1,
// c1 -> 128); wm[2i], wm[2i + 1] / bm[..]: Bottleneck i's two 3x3 convs (64 -> 64); w2 / b2: cv2 (1x1, (2 + nb) 64 -> 128) - all
// packed by upa_pack_conv_weight(bf16) with BN folded; y: (n, h, w, 128).  UPA_EUNSUPPORTED outside that form.
extern "C" int upa_c2f64_fused(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int nb,
                               int shortcut, const void* w1, const float* b1, const void* const* wm, const float* const* bm,
                               const void* w2, const float* b2, void* y, int c2, int ldy, int act, int dtype, const upa_opts* opts,
                               void* stream) {
  UPA_CHECK_ARG(x && y && w1 && b1 && wm && bm && w2 && b2 && n > 0 && h > 0 && w > 0, "c2f64_fused: bad args");
  const int off = UPA_OPT(opts, c2f);  // 1: never, 4: not this form, 6: only the n = 1 blocks (1.16x ring recompute; n = 2: 1.74x)
  // measured on MI355X (round 3): at 40 x 40 (51 k pixels at batch 32) the fused block replaces 4-6 launches that are each one
  // latency-bound round of the chip (yolov8n serial step 1.08 -> 1.02 ms); at 80 x 80 (yolov8s model.4 / model.15, 205 k pixels)
  // the separate launches fill the chip on their own and the tile-ring recompute loses: 20.2 k -> 19.2 k images/s
  const long max_px = UPA_OPT(opts, c2f64_max_px) == 0 ? 100000 : (UPA_OPT(opts, c2f64_max_px) < 0 ? (1L << 40) : UPA_OPT(opts, c2f64_max_px));
  if (off == 1 || off == 4 || (off == 6 && nb != 1) || (long)n * h * w > max_px || dtype != UPA_BF16 || act != UPA_ACT_SILU || c2 != 128 || c1 <= 0 || c1 % 64 != 0 || !(nb == 1 || nb == 2) ||
      ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 ||
      (long)n * h * w * ldx * 2 >= (1L << 32) - 4096 ||
      (up && (up_c <= 0 || up_c % 64 != 0 || up_c >= c1 || up_ld % 8 != 0 || ((uintptr_t)up % 16) != 0 || (h & 1) || (w & 1) ||
              (long)n * (h / 2) * (w / 2) * up_ld * 2 >= (1L << 32) - 4096))) {
    upa_set_error("c2f64_fused: outside the fused form (bf16, SiLU, C2f(c1 %% 64 == 0 -> 128, c = 64, n = 1 | 2))");
    return UPA_EUNSUPPORTED;  // the caller runs the separate convolutions
  }
  for (int i = 0; i < 2 * nb; ++i) UPA_CHECK_ARG(wm[i] && bm[i], "c2f64_fused: null Bottleneck weights");
  C2f64Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.up = (const char*)up; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 2 * nb; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.c1 = c1; p.ldx = ldx; p.ldy = ldy; p.upC = up ? up_c : 0; p.up_ld = up_ld; p.shortcut = shortcut ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  return nb == 2 ? c2f64_launch<2, 10, 10>(p, s) : c2f64_launch<1, 10, 20>(p, s);
}
