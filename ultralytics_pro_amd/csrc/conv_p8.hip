// 3x3 stride-1 convolution for the MFMA-bound layers (bf16, Cin % 64 == 0, Cout % 128 == 0: darknet53 in yolov3-rtdetr, the 128+-channel
// layers of yolov8s / yolov3-tiny) as a PHASED kernel: Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 (+ the Bottleneck
// shortcut, block.py:668), BN folded per utils/torch_utils.py:236-266, same packed weights and the same LDS images as conv_big.hip.
//
// What conv_big leaves on the table on these layers (K = 9 Cin >= 1152): its eight waves all run load -> multiply in the same phase
// and meet at one barrier + vmcnt(0) per tap, the halo of the next 64-channel chunk is fetched with the matrix pipe idle, and it needs
// a second co-resident workgroup to cover any of it - which the 200-400 tile grids of these layers rarely give it (matrix pipe ~50 %
// busy on a busy CU, 0.33-0.42 of the MFMA peak per launch).  This kernel is the guide's 8-wave two-group schedule
// (cdna_hip_programming.md 5, "The 256^2 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD") on conv_big's data layout:
//   * one workgroup per CU, 8 waves = 2 groups of 4 (one wave of each group per SIMD).  Every k32-step of every wave is a LOAD segment
//     (8 ds_read_b128: 4 weight + 4 pixel fragments; its share of the prefetch DMA) and an MFMA segment (16 v_mfma_f32_16x16x32_bf16),
//     separated by raw s_barriers; group 1 runs ONE barrier behind group 0, so on every SIMD one wave multiplies while its partner loads;
//   * nothing is waited for at zero in the loop: weight slabs (one (tap, 64-channel chunk) = 16 KB) go through a ring of FOUR LDS
//     buffers, issued three taps ahead; the NEXT chunk's halo goes into a SECOND halo buffer, one 1 KB piece per wave per tap during
//     taps 1-6 of the current chunk; a counted s_waitcnt vmcnt(N) at the end of a tap's second LOAD segment retires exactly the slab of
//     the next tap (N = 4 younger slab pieces + the halo pieces issued since: static per tap, the nine taps are unrolled);
//   * hazards by construction (the guide's placement rules): a staged buffer is read one barrier after every wave's counted wait for
//     it (two for the lagging group); a buffer is re-staged only after a barrier that follows the lgkmcnt(0) of its last readers - slab
//     s + 3 overwrites slab s - 1 in the second LOAD segment of tap s, the halo of chunk c + 1 overwrites that of chunk c - 1 from tap 1;
//   * tile = 256 pixels x 128 channels (wave: 64 x 64, 16 accumulator tiles), LDS = 2 halo images (<= 44 KB each) + 4 x 16 KB slabs
//     + 8 KB that absorbs the pieces past a short halo: <= 160 KB.
// Epilogue as conv_big (bias, SiLU, bf16, v_permlane16_swap -> 16-byte NHWC stores, residual read the same way).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* p8gptr_t;
typedef __attribute__((address_space(3))) void* p8lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_p8_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
constexpr int P8_NTB = 8;                   // n-tiles per workgroup (128 channels)
constexpr int P8_WBUF = 2 * P8_NTB * 1024;  // one (tap, chunk) weight slab
constexpr int P8_RING = 4;                  // slabs in LDS
constexpr int P8_HP = 6;                    // halo pieces (64 items of 16 B) per wave per chunk: halo <= 8 * 6 * 64 items
constexpr int P8_TRASH = 8 * 1024;          // where the pieces past the end of the halo image land

template <int ACT>
__device__ __forceinline__ float p8_act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
// s_waitcnt vmcnt(n) with n a constant after unrolling (the instruction takes an immediate)
__device__ __forceinline__ void p8_wait_vm(const int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// halo piece issued in the first LOAD segment of this tap?
constexpr int p8_h(int tap) { return tap >= 1 && tap <= P8_HP ? 1 : 0; }
}  // namespace

__global__ __launch_bounds__(512, 2) void conv_p8_kernel(const BigParams p) {
  constexpr int MT = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // waves w and w + 4 share a SIMD
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  const int tilesPerImg = p.tilesX * p.tilesY;
  int bid = p.no_xcd ? (int)blockIdx.x : upa_xcd_tile((int)blockIdx.x, tilesPerImg * p.N);
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX;
  const int txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int ntb0 = blockIdx.y * P8_NTB;

  const int haloItems = p.IH * p.IWp * 8;
  const int haloPadded = (haloItems + 63) & ~63;
  char* const hal0 = smem;
  char* const wbuf = smem + (size_t)haloPadded * 32;
  char* const trash = wbuf + P8_RING * P8_WBUF;

  int pl0[MT], pty[MT], ptx[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int pp = (wm * MT + i) * 16 + r;
    int ty = (int)__umulhi((unsigned)pp, p.magicTW);
    int tx = pp - ty * p.TW;
    if (ty >= p.TH) { ty = p.TH; tx = 0; }  // past the tile (TH * TW < 256): multiplied on halo pixel 0, never stored
    pty[i] = ty;
    ptx[i] = tx;
    pl0[i] = ty < p.TH ? ty * p.IWp + tx : 0;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nChunks = (p.KTT + 1) >> 1;

  // piece q (0 .. P8_HP - 1) of this wave's share of the halo of chunk c -> halo buffer c & 1 (pieces past the image: zeros to `trash`)
  auto stage_halo_piece = [&](int c, int q) __attribute__((always_inline)) {
    const int base = (q * 8 + wave) * 64;
    const int idx = base + lane;
    const int pix = idx >> 3;
    const int slot = idx & 7;
    const int cg = slot ^ (pix & 7);
    const int py = (int)__umulhi((unsigned)pix, p.magicIW);
    const int qx = pix - py * p.IWp;
    const int iy = iy0 + py, ix = ix0 + qx;
    const int ch = c * 64 + cg * 8;
    const char* src = reinterpret_cast<const char*>(g_p8_zero16);
    if (idx < haloItems && qx < p.IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin && c < nChunks)
      src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * 2;
    char* dst = base < haloPadded ? hal0 + (size_t)(c & 1) * haloPadded * 16 + base * 16 : trash + wave * 1024;
    __builtin_amdgcn_global_load_lds((p8gptr_t)src, (p8lptr_t)dst, 16, 0, 0);
  };
  // this wave's two fragments (wave, wave + 8 of 16: f = kt * 8 + j) of slab s = chunk * 9 + tap -> ring slot s & 3 (past the end: zeros)
  auto stage_w = [&](int c, int tap, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int ktg = c * 2 + kt;
      const int nt = ntb0 + wave;
      const char* src = reinterpret_cast<const char*>(g_p8_zero16);
      if (ktg < p.KTT && nt < p.NTn && c < nChunks) src = p.w + (((size_t)(tap * p.KTT + ktg) * p.NTn + nt) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((p8gptr_t)src, (p8lptr_t)(wbuf + slot * P8_WBUF + (kt * P8_NTB + wave) * 1024), 16, 0, 0);
    }
  };

  // ---- prologue: halo of chunk 0, slabs 0 .. 2
#pragma unroll
  for (int q = 0; q < P8_HP; ++q) stage_halo_piece(0, q);
  stage_w(0, 0, 0);
  stage_w(0, 1, 1);
  stage_w(0, 2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0 from here on

  const char* const wfrag = wbuf + (wn * NT) * 1024 + lane * 16;
  for (int c = 0; c < nChunks; ++c) {
    const char* const hal = hal0 + (size_t)(c & 1) * haloPadded * 16;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int slot = (c + tap) & 3;  // (9 c + tap) & 3
      const char* const wb = wfrag + slot * P8_WBUF;
      const int tapshift = kh * p.IWp + kw;
      int paddr[MT], pswz[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int pl = pl0[i] + tapshift;
        asm volatile("" : "+v"(pl));  // recomputed per tap (3 VALU): hoisted out of the chunk loop the nine taps' addresses are 72 registers
        paddr[i] = pl * 128;
        pswz[i] = pl & 7;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        // ---- LOAD segment
        u32x4 a[NT], b[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[j] = *reinterpret_cast<const u32x4*>(wb + (kt * P8_NTB + j) * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const u32x4*>(hal + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
        if (kt == 0) {
          if (p8_h(tap)) stage_halo_piece(c + 1, tap - 1);
        } else {
          // slab s + 3 (s = 9 c + tap) into the slot of slab s - 1, whose last readers passed their lgkmcnt(0) two barriers ago
          const int t3 = tap + 3 >= 9 ? tap + 3 - 9 : tap + 3;
          stage_w(tap + 3 >= 9 ? c + 1 : c, t3, (slot + 3) & 3);
          // this wave's pieces of slab s + 1 have landed: younger are slabs s + 2, s + 3 and the halo pieces of taps tap - 1, tap
          p8_wait_vm(4 + p8_h(tap) + p8_h(tap == 0 ? 8 : tap - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                                *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero pieces issued past the last slab

  // ---- epilogue from the accumulators (as conv_big.hip): lane (g, r) holds channels 16j + 4g .. + 3 of pixel r of m-tile i;
  // v_permlane16_swap pairs the quads of two neighbouring n-tiles so every lane stores 16 contiguous bytes
  const int cw = (blockIdx.y * P8_NTB + wn * NT) * 16;
  f32x4 biasv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = cw + j * 16 + g * 4;
    biasv[j] = (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
      const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW;
      const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
      char* yrow = p.y + (pixoff * p.ldy + cw) * 2;
      const char* rrow = p.res ? p.res + (pixoff * p.ldr + cw) * 2 : nullptr;
#pragma unroll
      for (int j = 0; j + 1 < NT; j += 2) {
        const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = p8_act<ACT>(acc[i][j][q] + biasv[j][q]);
          v1[q] = p8_act<ACT>(acc[i][j + 1][q] + biasv[j + 1][q]);
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
    }
  };
  if (p.act == UPA_ACT_SILU) epilogue(std::integral_constant<int, UPA_ACT_SILU>{});
  else if (p.act == UPA_ACT_RELU) epilogue(std::integral_constant<int, UPA_ACT_RELU>{});
  else epilogue(std::integral_constant<int, UPA_ACT_NONE>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t P8_HALO_MAX = (size_t)(160 * 1024 - P8_RING * P8_WBUF - P8_TRASH) / 2;  // 45056
// tile and halo pitch by conv_big's search (fewest tiles, then smallest halo, conflict-free pitch if it fits): the halo image of a
// 64-channel chunk may take 44 KB here (two of them + four slabs + the trash strip = 160 KB)
bool p8_geometry(BigParams& p) {
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
  p.KS = 3; p.stride = 1; p.pad = 1; p.OH = p.H; p.OW = p.W;
  if (!upa_conv_big_pick_tile(p, 256, P8_NTB, P8_HALO_MAX + 2 * (size_t)P8_WBUF + 256)) return false;
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  const size_t halo = (((size_t)p.IH * p.IWp * 8 + 63) & ~(size_t)63) * 16;
  return halo <= P8_HALO_MAX && halo <= (size_t)8 * P8_HP * 1024;
}
}  // namespace

bool upa_conv_p8_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad, int act,
                          int dtype, const upa_opts* opts) {
  const int mode = UPA_OPT(opts, conv_p8);  // 0 = by the size rule, 1 = never, 2 = every shape the kernel can run
  if (mode == 1) return false;
  if (dtype != UPA_BF16 || k != 3 || stride != 1 || pad != 1) return false;
  if (cin % 64 != 0 || cout % 128 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ldr % 8 != 0) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE && act != UPA_ACT_RELU) return false;
  if (h < 4 || w < 4) return false;
  if (mode == 2) return true;
  // Measured on MI355X (round 4, yolov3-rtdetr bs 16, tools/bench_conv.py --opts conv_p8=2 | 1): one workgroup per CU wins where the
  // whole layer is ONE round of 256-pixel x 128-channel tiles and the K loop is long enough to amortise the un-overlapped prologue and
  // epilogue - 512->256 @40x40 (200 tiles) 73.4 -> 59.6 us, 512->1024 @20x20 (256) 75.0 -> 65.3, 768->256 @40x40 (200) 108.5 -> 86.7.
  // With more tiles than CUs (400: 256->512 @40x40 65.1 -> 68.9, 256->128 @80x80 60.7 -> 64.1; 800: 128->256 @80x80 70.9 -> 85.7) the
  // second round is half empty and conv_big's two co-resident workgroups hide each other's prologue; with too few (128: 1024->512
  // @20x20 87.3 -> 98.0) half the chip idles.  One-round layers with K = 2304 win less but still win (yolov3-tiny bs 32 256->512 @20x20,
  // 256 tiles: 45.1 -> 40.6 us; yolov8s 256->128 @40x40, 200 tiles: 40.4 -> 35.9); at K = 1152 (128->128 @40x40: 26.0 -> 24.4, with
  // the shortcut 26.1 -> 26.0) the prologue and epilogue are half the tile and the rule stops.
  if (cin < 256) return false;
  BigParams q;
  memset(&q, 0, sizeof(q));
  q.N = n; q.H = h; q.W = w; q.Cin = cin; q.Cout = cout;
  if (!p8_geometry(q)) return false;
  const long tiles = (long)q.tilesX * q.tilesY * n * cdiv(q.NTn, P8_NTB);
  return tiles >= 176 && tiles <= 256;
}

int upa_conv_p8_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts) {
  p.no_xcd = UPA_OPT(opts, no_xcd);
  if (variant) *variant = (1 << 26) | (P8_NTB << 4) | 2;
  if (query_only) return UPA_OK;
  if (!p8_geometry(p)) return UPA_EUNSUPPORTED;
  const size_t halo = (((size_t)p.IH * p.IWp * 8 + 63) & ~(size_t)63) * 16;
  const size_t lds = 2 * halo + (size_t)P8_RING * P8_WBUF + P8_TRASH;
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N), (unsigned)cdiv(p.NTn, P8_NTB));
  if (upa_full_lds<conv_p8_kernel>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL(conv_p8_kernel, grid, dim3(512), lds, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
