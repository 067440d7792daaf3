// PERSISTENT 3x3 convolution with REGISTER-RESIDENT weights (bf16, stride 1, pad 1, Cin <= 64, Cout = 64): the Detect box branch's
// first conv on the 80 x 80 level (Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 with BN folded per
// utils/torch_utils.py:236-266; head.py:94-100) - 204800 pixels at batch 32, K = 576.
//
// What the phase stamps (-DUPA_STAMP, tools/bench_conv.py --stamps) showed on conv_big for this layer: 800 one-tile workgroups in 1.56
// rounds; every workgroup waits ~6400 cycles for its halo (all of them load at once), multiplies for ~10500 (the matrix pipe is 85 %
// busy in that phase), spends ~2900 in the epilogue and drains its stores - load, multiply and store run in lock step across the
// chip, so the pipe sits idle for half of every workgroup's life.  Ablations of a first persistent form with the conv_big operand
// scheme (all weights in LDS, 4 A + 2 B fragment reads per 8 MFMAs) pointed at the LDS fragment reads (20.8 us of 26.8 remained
// without MFMAs, stores and halo DMA; 8.7 without the reads too).  This kernel therefore
//   * keeps the WEIGHTS IN REGISTERS: a workgroup is four waves, wave w holds the 18 A fragments (9 taps x 2 k-tiles = 72 VGPRs) of
//     output channels 16 w .. 16 w + 15 for the whole launch - no weight traffic through LDS at all;
//   * walks 16 x 8 pixel tiles persistently, TWO workgroups per CU (47 KB of LDS each): the halo of a tile (10 x 18 pixels x 128 B,
//     XOR-swizzled) is double buffered, the DMA of tile t + 1 is issued at the top of tile t; ONE barrier per tile, none per tap;
//   * re-uses every B fragment for the three kernel rows that touch it: per (k-tile, kernel column) step a wave reads TEN halo-row
//     fragments for 24 MFMAs - 60 ds_read_b128 per 144 MFMAs (conv_big: 108 per 144), conflict-free at pitch 18;
//   * pins the read -> MFMA order with sched_barrier: left alone the scheduler sinks every ds_read to just before its first use (to
//     shorten live ranges) and the wave waits out one LDS latency per three MFMAs - 37 cycles per MFMA measured, 19.6 after pinning
//     (16 = the pipe's rate); the 60 fragment addresses are 16 per-lane registers + instruction offsets (hoisted out of the tile
//     loop they would spill);
//   * waits for the next halo BEFORE issuing a tile's stores: vmcnt counts stores too on gfx9 and mixed loads / stores retire out of
//     order, so a wait placed after the stores would sit out their acknowledgement.
// Measured (64 -> 64 @80x80, batch 32): 24.9 us against 27.9 for conv_big (batch 256: 159 against 185).  Per tile and workgroup in
// steady state: next-halo issue 1850 cycles, MFMA loop 2820 (144 MFMAs), epilogue 3930; the start-up (first halo + weights) costs
// 8400.  What is left is the epilogue's vector work and that start-up, not the matrix pipe.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"
UPA_STAMP_DEFINE(conv_ws3)

typedef __attribute__((address_space(1))) const void* wgptr_t;
typedef __attribute__((address_space(3))) void* wlptr_t;

__device__ __attribute__((aligned(16))) unsigned g_ws3_zero16[4] = {0u, 0u, 0u, 0u};

namespace ws3 {
template <int ACT>
__device__ __forceinline__ float act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
constexpr int TH = 8, IH = TH + 2;  // tile rows, halo rows
constexpr int IW = 18;           // halo pitch = tile width 16 + 2
}  // namespace ws3

// NW waves = NW n-tiles (4: 64 output channels).  Wave w holds the weights of channels 16 w .. 16 w + 15 and computes them for ALL
// 8 x 16 pixels of the tile.
// STATS (training forward, act none, no residual): the lane also keeps the running sum / sum of squares of the bf16-ROUNDED values it
// stores (its four channels, its pixel column, every tile the workgroup walks); at the end the 16 pixel columns of a row group meet by
// a fixed butterfly and row blockIdx.x of p.stats receives the workgroup's sums - every wave owns its channels, nothing crosses waves.
template <int NW, bool STATS = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_ws3_kernel(const BigParams p) {
  using namespace ws3;
  constexpr int NTHR = NW * 64;
  constexpr int HIT = IH * IW * 8;                       // 16-byte items of a halo
  constexpr int HITP = (HIT + 63) / 64 * 64;             // ... padded to whole wave-instructions (1 KiB of LDS each)
  constexpr int HPASS = (HITP + NTHR - 1) / NTHR;
  constexpr int HB = HITP * 16;                          // halo buffer bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hal = smem;  // two buffers of HB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int numTiles = tilesPerImg * p.N;
  UPA_STAMP_AT(0);
  UPA_STAMP_HWID();

  // this thread's halo items (pixel, channel group) are the same for every tile: packed (row | column << 8 | channel group << 16) for
  // the bounds test, and as a byte offset from the tile's first halo pixel (32-bit arithmetic: an image is far below 4 GB)
  int hitem[HPASS], hoff[HPASS];
#pragma unroll
  for (int it = 0; it < HPASS; ++it) {
    const int idx = it * NTHR + tid;
    const int pix = idx >> 3, slot = idx & 7;
    const int py = pix / IW, pxc = pix - py * IW, cg = slot ^ (pix & 7);
    hitem[it] = (idx >= HIT ? 255 : py) | (pxc << 8) | (cg << 16);  // row 255: never inside the image
    hoff[it] = ((py * p.W + pxc) * p.ldx + cg * 8) * 2;
  }
  const size_t imgBytes = (size_t)p.H * p.W * (size_t)p.ldx * 2;
  auto stage_halo = [&](int t, int b) __attribute__((always_inline)) {
    const int n = t / tilesPerImg;
    const int t2 = t - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int iy0 = tyi * TH - 1, ix0 = txi * 16 - 1;
    // (scalar) the tile's first halo pixel; may lie outside the image - lanes that would read there take the zero page
    const char* xt = p.x + (size_t)n * imgBytes + ((long)iy0 * p.W + ix0) * (long)p.ldx * 2;
#pragma unroll
    for (int it = 0; it < HPASS; ++it) {
      if (it * NTHR + wave * 64 >= HITP) break;  // wave-uniform: the last pass is partial
      const int iy = iy0 + (hitem[it] & 255), ix = ix0 + ((hitem[it] >> 8) & 255);
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (hitem[it] >> 16) * 8 < p.Cin;  // (row 255 fails iy < H)
      const char* src = ok ? xt + hoff[it] : reinterpret_cast<const char*>(g_ws3_zero16);
      __builtin_amdgcn_global_load_lds((wgptr_t)src, (wlptr_t)(hal + b * HB + (it * NTHR + wave * 64) * 16), 16, 0, 0);
    }
  };

  int t = blockIdx.x;
  if (t >= numTiles) return;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const f32x4 biasv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wave * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  stage_halo(t, 0);
  // The first-tile wait below counts on program order "halo DMA, then exactly 18 weight loads": pin it.  The empty asm with a memory
  // clobber keeps IR passes from hoisting a weight load above the DMA, the scheduling barrier keeps the machine scheduler from it.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // this wave's weights: fragment (tap, kt) of n-tile `wave`  <-  packed [tap][KTT][NTn][lane][16 B]; 18 x 16 B per lane
  u32x4 wr[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      // Cin <= 32 has one k-tile: the second comes from the zero page (an unconditional load, no branch per fragment)
      const char* src = kt < p.KTT ? p.w + (((size_t)(tap * p.KTT + kt) * p.NTn + wave) * 64 + lane) * 16
                                   : reinterpret_cast<const char*>(g_ws3_zero16);
      wr[tap][kt] = *reinterpret_cast<const u32x4*>(src);
    }

  // vmcnt also counts STORES on gfx9 and mixed loads / stores retire out of order, so a wait for the next halo must not sit behind
  // freshly issued epilogue stores (their acknowledgements take ~1 us): the wait for halo t + 1 comes BEFORE the stores of tile t.
  // First tile: wait for the halo only - the 18 weight loads were issued after it (loads retire in order), and the MFMA loop's own
  // counted waits let step 0 start on its three fragments while the other fifteen are still in flight.
  asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  int buf = 0;
  for (; t < numTiles; t += gridDim.x) {
    __syncthreads();  // every wave's share of halo t has landed (each waited before its stores); the other buffer is free
#ifdef UPA_STAMP
    const bool stamped = t == (int)blockIdx.x + (int)gridDim.x;  // the workgroup's second tile
    if (t == (int)blockIdx.x) UPA_STAMP_AT(1);
    if (stamped) UPA_STAMP_AT(2);
#endif
    const int tn = t + gridDim.x;
    if (tn < numTiles) stage_halo(tn, buf ^ 1);
#ifdef UPA_STAMP
    if (stamped) UPA_STAMP_AT(3);
#endif
    const char* hb = hal + buf * HB;
    f32x4 acc[TH];
#pragma unroll
    for (int i = 0; i < TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ten halo rows per (k-tile, kernel column) step: the fragment of halo row q feeds the (up to) three output rows q - kh; as soon
    // as its MFMAs are issued the register takes the same row of the NEXT step, so fragment reads run one step ahead of the MFMAs
    u32x4 b[IH];
    // halo pixel px = 18 q + kw + r sits at px * 128 + ((4 kt + g) ^ (px & 7)) * 16 and px & 7 = (2 q + kw + r) & 7: with c = (2 q + kw) & 7
    // a compile-time constant the lane-dependent part is one of 16 registers ta[kt][c] (rebuilt per tile for the buffer in use) and
    // (18 q + kw) * 128 goes into the instruction's offset field - left to itself the compiler hoists all 60 addresses out of the
    // tile loop and spills
    const char* ta[2][8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int c = 0; c < 8; ++c) ta[kt][c] = hb + r * 128 + (((kt * 4 + g) ^ ((c + r) & 7)) << 4);
    auto read_b = [&](int s, int q) __attribute__((always_inline)) {
      const int kt = s / 3, kw = s - kt * 3;
      return *reinterpret_cast<const u32x4*>(ta[kt][(2 * q + kw) & 7] + (q * IW + kw) * 128);
    };
#pragma unroll
    for (int q = 0; q < IH; ++q) b[q] = read_b(0, q);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int kt = s / 3, kw = s - kt * 3;
#pragma unroll
      for (int q = 0; q < IH; ++q) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int i = q - kh;
          if (i >= 0 && i < TH)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wr[kh * 3 + kw][kt]),
                                                             *reinterpret_cast<const bf16x8*>(&b[q]), acc[i], 0, 0, 0);
        }
        if (s + 1 < 6) b[q] = read_b(s + 1, q);
        // pin the order: left alone the scheduler sinks every read to just before its first use a step later (to shorten live
        // ranges) and the wave then waits out one LDS latency per three MFMAs (37 cycles per MFMA measured instead of 16)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef UPA_STAMP
    if (stamped) UPA_STAMP_AT(4);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // halo t + 1 (issued a whole MFMA loop ago) and the stores of tile t - 1
#ifdef UPA_STAMP
    if (stamped) UPA_STAMP_AT(5);
#endif
    // ---- epilogue: lane (g, r) holds channels 16 wave + 4g .. + 3 of pixel r of row i  ->  8-byte NHWC stores
    const int n = t / tilesPerImg;
    const int t2 = t - n * tilesPerImg;
    const int tyi = t2 / p.tilesX, txi = t2 - tyi * p.tilesX;
    const int ox = txi * 16 + r;
    const int cb = 16 * wave + 4 * g;
    const bool colok = ox < p.OW && cb < p.Cout;
    const size_t pix0 = ((size_t)n * p.OH + (size_t)tyi * TH) * p.OW;  // (scalar) first pixel of the tile's first row
    char* yt = p.y + pix0 * (size_t)p.ldy * 2;
    const int yoff = (ox * p.ldy + cb) * 2, yrow = p.OW * p.ldy * 2;
    u32x2 rv[TH];
    if (p.res) {  // Bottleneck shortcut: same shape as y; all eight row loads in flight before the activation math
      const char* rt = p.res + pix0 * (size_t)p.ldr * 2;
      const int roff = (ox * p.ldr + cb) * 2, rrow = p.OW * p.ldr * 2;
#pragma unroll
      for (int i = 0; i < TH; ++i)
        rv[i] = (colok && tyi * TH + i < p.OH) ? *reinterpret_cast<const u32x2*>(rt + i * rrow + roff) : u32x2{0u, 0u};
    }
    auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act<ACT>(acc[i][q] + biasv[q]);
        if (p.res) {
          v[0] += __uint_as_float(rv[i][0] << 16); v[1] += __uint_as_float(rv[i][0] & 0xFFFF0000u);
          v[2] += __uint_as_float(rv[i][1] << 16); v[3] += __uint_as_float(rv[i][1] & 0xFFFF0000u);
        }
        const unsigned pk0 = pack_bf16x2(v[0], v[1]), pk1 = pack_bf16x2(v[2], v[3]);
        const bool ok = colok && tyi * TH + i < p.OH;
        if constexpr (STATS) {
          const float mk = ok ? 1.f : 0.f;
          const float a0 = __uint_as_float(pk0 << 16) * mk, a1 = __uint_as_float(pk0 & 0xFFFF0000u) * mk;
          const float a2 = __uint_as_float(pk1 << 16) * mk, a3 = __uint_as_float(pk1 & 0xFFFF0000u) * mk;
          ssum[0] += a0; ssq[0] = fmaf(a0, a0, ssq[0]);
          ssum[1] += a1; ssq[1] = fmaf(a1, a1, ssq[1]);
          ssum[2] += a2; ssq[2] = fmaf(a2, a2, ssq[2]);
          ssum[3] += a3; ssq[3] = fmaf(a3, a3, ssq[3]);
        }
        if (ok) *reinterpret_cast<u32x2*>(yt + i * yrow + yoff) = u32x2{pk0, pk1};
      }
    };
    if (p.act == UPA_ACT_SILU) epilogue(std::integral_constant<int, UPA_ACT_SILU>{});
    else if (p.act == UPA_ACT_RELU) epilogue(std::integral_constant<int, UPA_ACT_RELU>{});
    else epilogue(std::integral_constant<int, UPA_ACT_NONE>{});
#ifdef UPA_STAMP
    if (stamped) UPA_STAMP_AT(6);
#endif
    buf ^= 1;
  }
  if constexpr (STATS) {
    auto row16 = [](float v) __attribute__((always_inline)) {
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, true));
      return v;
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) { ssum[q] = row16(ssum[q]); ssq[q] = row16(ssq[q]); }
    if (r == 0) {
      float* row = p.stats + (size_t)blockIdx.x * 2 * p.stats_ld + 16 * wave + 4 * g;
      *reinterpret_cast<f32x4*>(row) = f32x4{ssum[0], ssum[1], ssum[2], ssum[3]};
      *reinterpret_cast<f32x4*>(row + p.stats_ld) = f32x4{ssq[0], ssq[1], ssq[2], ssq[3]};
    }
  }
  UPA_STAMP_AT(7);
}

namespace {
int ws3_num_cu() {
  static int numCU = 0;
  if (!numCU) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCU <= 0) numCU = 256;
  }
  return numCU;
}
}  // namespace

// Dispatch rule (upa_opts.conv_ws3: 0 = by size, 1 = never, 2 = every shape the kernel can run).
bool upa_conv_ws3_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, bool residual, int k, int stride, int pad,
                           int act, int dtype, const upa_opts* opts) {
  const int mode = UPA_OPT(opts, conv_ws3);
  if (mode == 1) return false;
  if (dtype != UPA_BF16 || k != 3 || stride != 1 || pad != 1) return false;
  if (cin > 64 || cin % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || cout != 64) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE && act != UPA_ACT_RELU) return false;
  if (mode == 2) return true;
  if (mode == 3) return cin == 64 && (long)n * h * w >= 8192 && (long)n * h * w < 100000;  // one tile per workgroup at most: no resident walk
  // measured against conv_big at batch 32 (tools/bench_conv.py): 64 -> 64 @80x80 23.6 vs 27.0 us, @40x40 11.0 vs 12.0, @20x20 8.4 vs 9.0
  return cin == 64 && (long)n * h * w >= 8192;
}

static int ws3_launch(BigParams p, int query_only, int* variant, int* rows, void* stream);
int upa_conv_ws3_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts) {
  p.stats = nullptr;
  return ws3_launch(p, query_only, variant, nullptr, stream);
}
// p.stats != nullptr (act none, no residual, no bias): the convolution + the first stage of the batch statistics; *rows = rows written
// (one per workgroup, at most two per CU)
int upa_conv_ws3_launch_stats(BigParams p, int* rows, long max_rows, void* stream, const upa_opts* opts) {
  if (!p.stats || p.res || p.bias || p.act != UPA_ACT_NONE || max_rows < 2 * ws3_num_cu()) return UPA_EUNSUPPORTED;
  return ws3_launch(p, 0, nullptr, rows, stream);
}
static int ws3_launch(BigParams p, int query_only, int* variant, int* rows, void* stream) {
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
  if (variant) *variant = (1 << 24) | (4 << 4) | 4;
  if (query_only) return UPA_OK;
  constexpr int HIT = ws3::IH * ws3::IW * 8, HITP = (HIT + 63) / 64 * 64;
  const size_t lds = 2 * (size_t)HITP * 16;
  p.TH = ws3::TH; p.TW = 16;
  p.tilesX = cdiv(p.OW, 16);
  p.tilesY = cdiv(p.OH, ws3::TH);
  const long tiles = (long)p.tilesX * p.tilesY * p.N;
  if (tiles >= (1L << 31)) return UPA_EUNSUPPORTED;
  if (hipError_t e = upa_full_lds<conv_ws3_kernel<4>>(); e != hipSuccess) {
    upa_set_error("conv_ws3: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  const int cus = ws3_num_cu();
  const unsigned grid = (unsigned)(tiles < 2 * cus ? tiles : 2 * cus);
  if (p.stats) {
    if (hipError_t e = upa_full_lds<conv_ws3_kernel<4, true>>(); e != hipSuccess) {
      upa_set_error("conv_ws3: cannot raise LDS limit: %s", hipGetErrorString(e));
      return UPA_ELAUNCH;
    }
    p.stats_ld = 64;
    *rows = (int)grid;
    hipLaunchKernelGGL((conv_ws3_kernel<4, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(conv_ws3_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
