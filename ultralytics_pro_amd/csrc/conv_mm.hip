// 3x3 stride-1 convolution for the MFMA-bound layers (bf16, Cin a multiple of 64, Cout a multiple of 128: darknet53 in yolov3-rtdetr,
// the 128+-channel layers of yolov8s / yolov3-tiny): Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 (+ the Bottleneck
// shortcut, block.py:668), BN folded per utils/torch_utils.py:236-266, same packed weights as every other conv kernel.
//
// conv_big.hip runs these layers at 0.35-0.41 of the bf16 MFMA peak: eight waves of 64 x 64 wave tiles on v_mfma_f32_16x16x32_bf16
// meet at a barrier after every 32 MFMAs (512 matrix-pipe cycles) and spend ~3 vector instructions per MFMA around them.  Same
// LDS plan here (halo of a 64-channel chunk staged once by LDS-DMA, one (tap, chunk) weight slab of 16 KB double buffered, two
// workgroups per CU) but HALF the waves with TWICE the tile on the 32 x 32 x 16 instruction:
//   * workgroup = 4 waves (2 pixel halves x 2 channel halves) = 256 pixels x 128 channels; wave tile 128 pixels x 64 channels = 4 x 2
//     MFMA tiles of 32 x 32, 128 accumulator registers, <= 256 registers per lane (two waves per SIMD);
//   * per (tap, chunk) a wave issues 32 MFMAs of 32 cycles (1024 matrix-pipe cycles between barriers instead of 512) around 24
//     ds_read_b128 (instead of 32 per 1024 cycles) and the address arithmetic of 4 m-tiles instead of 2 x 4;
//   * A fragments (32 couts x 16 cins) are assembled from the standard packed layout [k-tile][n-tile of 16][lane][16 B] by a lane
//     remap (lanes 0-15 / 16-31 take two neighbouring 16-cout tiles, lanes 32-63 the next 8 input channels): conflict-free as is;
//   * B fragments: an m-tile is 32 consecutive pixels of the TH x TW tile in row-major order; lane l reads 16-byte group
//     2 s + l / 32 of its pixel's 128-byte record.  The record's groups are XOR-swizzled by ((column >> 1) + c * row) & 7 with c picked on
//     the host per tile width (upa_mm_pick_c: exhaustive check of the two 16-lane service groups of ds_read_b128 over every m-tile and
//     tap) - conflict-free at the tight pitch for 16-, 20- and 40-pixel-wide tiles;
//   * epilogue from the accumulators: lane l holds, for pixel l % 32, channel quads 8 q + 4 (l / 32) .. + 3; one v_permlane32_swap per
//     register pair gives every lane 8 consecutive channels = one 16-byte NHWC store (residual read the same way).
//
// MEASURED (round 4, MI355X, yolov3-rtdetr bs 16, tools/bench_conv.py): correct (tests/test_hip_ops.py: test_conv_mm_kernel) but SLOWER
// than conv_big on every layer - 512->256 @40x40 102.7 vs 72.8 us, 256->128 @80x80 80.7 vs 60.0, 128->256 @80x80 (+res) 98.1 vs 70.6,
// 256->512 @40x40 88.2 vs 64.2 - with the fragment reads pinned one k16-step ahead of their MFMAs as well as with the compiler's
// own order (103.5 us).  Two waves per SIMD that meet at every tap's barrier cannot cover each other's waits the way four do, and the
// 32x32x16 shape holds a lower clock than 16x16x32 under load (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x).  Kept as an
// opt-in experiment (upa_opts.conv_mm = 2); the default dispatch never takes it.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* mgptr_t;
typedef __attribute__((address_space(3))) void* mlptr_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(16))) unsigned g_mm_zero16[4] = {0u, 0u, 0u, 0u};

struct MmParams {
  BigParams b;      // x, y, res, w, bias, N, H, W, Cin, ldx, OH, OW, Cout, ldy, ldr, TH, TW, tilesX, tilesY, IH, IW, IWp, KTT, NTn, act, magicTW, magicIW
  int swzC;         // row multiplier of the halo swizzle
};

namespace {
constexpr int MM_NTB = 8;                  // 16-cout n-tiles per workgroup (128 channels)
constexpr int MM_WBUF = 2 * MM_NTB * 1024;  // one (tap, 64-channel chunk) weight slab

template <int ACT>
__device__ __forceinline__ float mm_act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
}  // namespace

template <int ACT, bool RES>
__global__ __launch_bounds__(256, 2) void conv_mm_kernel(const MmParams q) {
  const BigParams& p = q.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const int tilesPerImg = p.tilesX * p.tilesY;
  int bid = p.no_xcd ? (int)blockIdx.x : upa_xcd_tile((int)blockIdx.x, tilesPerImg * p.N);
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX, txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int ntb0 = blockIdx.y * MM_NTB;

  const int haloItems = p.IH * p.IWp * 8;
  const int haloPadded = (haloItems + 63) & ~63;
  char* hal = smem;
  char* wbuf = smem + (size_t)haloPadded * 16;

  // this lane's pixel of each of the wave's four 32-pixel m-tiles
  int pty[4], ptx[4];
  bool pin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pp = (wm * 4 + i) * 32 + l32;
    int ty = (int)__umulhi((unsigned)pp, p.magicTW);
    int tx = pp - ty * p.TW;
    pin[i] = ty < p.TH;
    if (!pin[i]) { ty = 0; tx = 0; }  // past the tile (TH * TW < 256): multiplied on halo pixel (0, 0), never stored
    pty[i] = ty;
    ptx[i] = tx;
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage_halo = [&](int c) __attribute__((always_inline)) {
    const int c0 = c * 64;
    for (int base = wave * 64; base < haloPadded; base += 256) {
      const int idx = base + lane;
      const int pix = idx >> 3, slot = idx & 7;
      const int py = (int)__umulhi((unsigned)pix, p.magicIW);
      const int qx = pix - py * p.IWp;
      const int cg = slot ^ (((qx >> 1) + q.swzC * py) & 7);
      const int iy = iy0 + py, ix = ix0 + qx;
      const int ch = c0 + cg * 8;
      const char* src = reinterpret_cast<const char*>(g_mm_zero16);
      if (idx < haloItems && qx < p.IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin)
        src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * 2;
      __builtin_amdgcn_global_load_lds((mgptr_t)src, (mlptr_t)(hal + base * 16), 16, 0, 0);
    }
  };
  // weight slab of (tap, chunk c) -> buffer b: 16 fragments of 1 KiB (f = kt * 8 + j), wave w brings fragments w, w + 4, ...
  auto stage_w = [&](int c, int tap, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int f0 = 0; f0 < 2 * MM_NTB; f0 += 4) {
      const int f = f0 + wave;
      const int kt = f >> 3, j = f & 7;
      const int ktg = c * 2 + kt, nt = ntb0 + j;
      const char* src = reinterpret_cast<const char*>(g_mm_zero16);
      if (ktg < p.KTT && nt < p.NTn) src = p.w + (((size_t)(tap * p.KTT + ktg) * p.NTn + nt) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((mgptr_t)src, (mlptr_t)(wbuf + b * MM_WBUF + f * 1024), 16, 0, 0);
    }
  };
  // A fragment (32 couts x 16 cins) of n32-tile j, k16-step s from a slab in packed order: lane l -> 16-cout tile wn * 4 + 2 j + (l32 >> 4),
  // row l & 15, input-channel group 2 (s & 1) + hi of k-tile s >> 1
  const int aoff = ((wn * 4 + (l32 >> 4)) * 64 + hi * 16 + (lane & 15)) * 16;

  const int nChunks = (p.KTT + 1) >> 1;
  stage_halo(0);
  stage_w(0, 0, 0);
  int buf = 0;
  for (int c = 0; c < nChunks; ++c) {
    int kh = 0, kw = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tap + 1 < 9) stage_w(c, tap + 1, buf ^ 1);
      const char* wb = wbuf + buf * MM_WBUF + aoff;
      int paddr[4], pswz[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hy = pty[i] + kh, hx = ptx[i] + kw;
        paddr[i] = (hy * p.IWp + hx) * 128;
        pswz[i] = ((hx >> 1) + q.swzC * hy) & 7;
      }
      // fragment reads one k16-step AHEAD of the MFMAs that use them, the written order pinned (left alone the scheduler sinks
      // every ds_read to just before its first use to shorten live ranges and the wave sits out an LDS latency per two MFMAs)
      u32x4 a[2][2], b[2][4];
      auto load_frags = [&](int s, u32x4 (&aa)[2], u32x4 (&bb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) aa[j] = *reinterpret_cast<const u32x4*>(wb + ((s >> 1) * MM_NTB + 2 * j) * 1024 + (s & 1) * 512);
#pragma unroll
        for (int i = 0; i < 4; ++i) bb[i] = *reinterpret_cast<const u32x4*>(hal + paddr[i] + (((2 * s + hi) ^ pswz[i]) << 4));
      };
      load_frags(0, a[0], b[0]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) load_frags(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a[s & 1][j]),
                                                                *reinterpret_cast<const bf16x8*>(&b[s & 1][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      buf ^= 1;
      if (++kw == 3) { kw = 0; ++kh; }
    }
    if (c + 1 < nChunks) {
      __syncthreads();  // every wave is done with this chunk's halo before it is overwritten
      stage_halo(c + 1);
      stage_w(c + 1, 0, buf);
    }
  }

  // ---- epilogue: acc[i][j][4 qd + e] = channel wn * 64 + 32 j + 8 qd + 4 hi + e of pixel l32 of m-tile i
  const int cw = (blockIdx.y * MM_NTB + wn * 4) * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
    const bool pok = pin[i] && oy < p.OH && ox < p.OW;
    const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
    char* yrow = p.y + (pixoff * p.ldy + cw) * 2;
    const char* rrow = RES ? p.res + (pixoff * p.ldr + cw) * 2 : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // register groups qd = 2 t (channels 16 t + 4 hi ..) and 2 t + 1 (16 t + 8 + 4 hi ..): after the swap the lower 32 lanes hold
        // channels 16 t .. 16 t + 7 of their pixel, the upper 32 lanes 16 t + 8 .. 16 t + 15
        const int cb = 32 * j + 16 * t + 8 * hi;
        const f32x4 bv0 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cw + 32 * j + 16 * t + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 bv1 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cw + 32 * j + 16 * t + 8 + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
        float x0[4], x1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x0[e] = mm_act<ACT>(acc[i][j][8 * t + e] + bv0[e]);
          x1[e] = mm_act<ACT>(acc[i][j][8 * t + 4 + e] + bv1[e]);
        }
        const bool ok = pok && cw + cb < p.Cout;
        if constexpr (RES) {
          float v8[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0[e]), __float_as_uint(x1[e]), false, false);
            v8[e] = __uint_as_float(sw[0]);
            v8[4 + e] = __uint_as_float(sw[1]);
          }
          if (ok) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(rrow + cb * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v8[2 * e] += __uint_as_float(rv[e] << 16);
              v8[2 * e + 1] += __uint_as_float(rv[e] & 0xFFFF0000u);
            }
            *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{pack_bf16x2(v8[0], v8[1]), pack_bf16x2(v8[2], v8[3]), pack_bf16x2(v8[4], v8[5]),
                                                            pack_bf16x2(v8[6], v8[7])};
          }
        } else {
          auto lo = __builtin_amdgcn_permlane32_swap(pack_bf16x2(x0[0], x0[1]), pack_bf16x2(x1[0], x1[1]), false, false);
          auto hh = __builtin_amdgcn_permlane32_swap(pack_bf16x2(x0[2], x0[3]), pack_bf16x2(x1[2], x1[3]), false, false);
          if (ok) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hh[0], lo[1], hh[1]};
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// extra LDS cycles of the B-fragment reads of one workgroup tile under swizzle ((qx >> 1) + c * py) & 7 at pitch P: the two 16-lane
// service groups of ds_read_b128 ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of a half-wave), every m-tile, every tap
constexpr int mm_conflicts(int TW, int TH, int P, int c) {
  constexpr int G[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  int extra = 0;
  const int npx = TW * TH;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw)
      for (int m0 = 0; m0 < 256; m0 += 32)
        for (int g = 0; g < 2; ++g) {
          int cnt[16] = {0};
          int worst = 1;
          for (int k = 0; k < 16; ++k) {
            int pp = m0 + G[g][k];
            int ty = pp / TW, tx = pp % TW;
            if (pp >= npx) { ty = 0; tx = 0; }
            const int py = ty + kh, qx = tx + kw;
            const int pl = py * P + qx;
            const int slot = ((qx >> 1) + c * py) & 7;        // group 0 of the record (any cg XORs every slot alike)
            const int key = (pl & 1) * 8 + slot;               // 16 (bank half, slot) combinations
            // two lanes on the same pixel read the same address (broadcast): only distinct pixels conflict
            bool dup = false;
            for (int k2 = 0; k2 < k; ++k2) {
              int pp2 = m0 + G[g][k2];
              int ty2 = pp2 / TW, tx2 = pp2 % TW;
              if (pp2 >= npx) { ty2 = 0; tx2 = 0; }
              if (ty2 == ty && tx2 == tx) dup = true;
            }
            if (!dup && ++cnt[key] > worst) worst = cnt[key];
          }
          extra += worst - 1;
        }
  return extra;
}
int mm_pick_c(int TW, int TH, int P) {
  int best = 0, bc = 1 << 30;
  for (int c = 0; c < 8; ++c) {
    const int e = mm_conflicts(TW, TH, P, c);
    if (e < bc) { bc = e; best = c; }
  }
  return best;
}
}  // namespace

bool upa_conv_mm_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad, int act,
                          int dtype, const upa_opts* opts) {
  // upa_opts.conv_mm: 2 = every shape the kernel can run (tests, tools/bench_conv.py --opts conv_mm=2); 0 / 1 = never.  NOT a default
  // path: measured on MI355X (round 4, yolov3-rtdetr bs 16) it is 25-40 % SLOWER than conv_big on every layer it can run - see the
  // header of this file
  const int mode = UPA_OPT(opts, conv_mm);
  if (mode != 2) return false;
  if (dtype != UPA_BF16 || k != 3 || stride != 1 || pad != 1) return false;
  if (cin % 64 != 0 || cout % 128 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ldr % 8 != 0) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE && act != UPA_ACT_RELU) return false;
  if (h < 8 || w < 8) return false;
  return true;
}

int upa_conv_mm_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts) {
  p.no_xcd = UPA_OPT(opts, no_xcd);
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
  p.KS = 3; p.stride = 1; p.pad = 1; p.OH = p.H; p.OW = p.W;
  if (variant) *variant = (1 << 25) | (MM_NTB << 4) | 2;
  if (query_only) return UPA_OK;
  // tile: 256 pixels; whole-width rows when the map is narrow (40 x 6, 20 x 12), else 16 x 16 (least halo); fewest tiles first
  long best = -1;
  int btw = 0, bth = 0;
  for (int tw = 8; tw <= 64; tw += 2) {
    if (tw > ((p.OW + 1) & ~1)) break;
    int th = 256 / tw;
    if (th > p.OH) th = p.OH;
    if (th < 2) continue;
    th = cdiv(p.OH, cdiv(p.OH, th));  // balanced rows
    const long tiles = (long)cdiv(p.OW, tw) * cdiv(p.OH, th);
    const long cost = tiles * 65536 + (long)(th + 2) * (tw + 2);
    if (best < 0 || cost < best) { best = cost; btw = tw; bth = th; }
  }
  if (best < 0) return UPA_EUNSUPPORTED;
  p.TW = btw; p.TH = bth;
  p.IH = bth + 2; p.IW = btw + 2;
  p.IWp = (p.IW + 1) & ~1;  // even pitch: the bank half of a record is the parity of its column
  p.HALF = 0;
  p.magicTW = (unsigned)((0x100000000ULL + p.TW - 1) / p.TW);
  p.magicIW = (unsigned)((0x100000000ULL + p.IWp - 1) / p.IWp);
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  const size_t halo = (((size_t)p.IH * p.IWp * 8 + 63) & ~(size_t)63) * 16;
  const size_t lds = halo + 2 * (size_t)MM_WBUF;
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  MmParams q;
  q.b = p;
  q.swzC = mm_pick_c(p.TW, p.TH, p.IWp);
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N), (unsigned)cdiv(p.NTn, MM_NTB));
  hipStream_t s = (hipStream_t)stream;
#define UPA_MM_LAUNCH(ACT_, RES_)                                                                        \
  do {                                                                                                   \
    if (upa_full_lds<conv_mm_kernel<ACT_, RES_>>() != hipSuccess) return UPA_ELAUNCH;                    \
    hipLaunchKernelGGL((conv_mm_kernel<ACT_, RES_>), grid, dim3(256), lds, s, q);                        \
  } while (0)
  if (p.act == UPA_ACT_SILU) { if (p.res) UPA_MM_LAUNCH(UPA_ACT_SILU, true); else UPA_MM_LAUNCH(UPA_ACT_SILU, false); }
  else if (p.act == UPA_ACT_RELU) { if (p.res) UPA_MM_LAUNCH(UPA_ACT_RELU, true); else UPA_MM_LAUNCH(UPA_ACT_RELU, false); }
  else { if (p.res) UPA_MM_LAUNCH(UPA_ACT_NONE, true); else UPA_MM_LAUNCH(UPA_ACT_NONE, false); }
#undef UPA_MM_LAUNCH
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
