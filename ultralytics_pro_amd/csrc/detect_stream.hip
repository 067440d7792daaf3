// One level of the Detect head (yolov8n's 80 x 80 level: 64 input channels, box branch 64 -> 64 -> 64, class branch 64 -> 80 -> 80 -> nc) as
// ONE launch in LINE-BUFFER form: conv3x3 -> conv3x3 -> 1x1 -> DFL / dist2bbox / sigmoid -> decoded rows + best-class NMS keys.
//   Detect.__init__ / forward   ultralytics/nn/modules/head.py:94-100, 116-126   cv2[i] = Conv(x, c2, 3), Conv(c2, c2, 3), Conv2d(c2, 4 reg_max, 1)
//                                                                                cv3[i] = Conv(x, c3, 3), Conv(c3, c3, 3), Conv2d(c3, nc, 1)
//   Detect._inference           ultralytics/nn/modules/head.py:151-191           dfl, dist2bbox * stride, sigmoid
// The tile form of this level (conv_big.hip: a stacked 144-channel first conv, then conv_big_mix<..5,2 | ..4,1> = second 3x3 + 1x1 + decode)
// writes and re-reads the 144-channel intermediate (236 MB per batch of 32), DMAs every 3x3 weight slab into LDS once per 256-pixel tile and
// pays a halo wait + epilogue per tile: 49 + 70 us at 0.2-0.28 of the MFMA peak.  Here, as in c2f_stream.hip, a workgroup owns a vertical
// STRIP of one image (WS = 20 output columns, L rows) and walks down it two rows per step with ONE s_barrier per step:
//   * a workgroup runs ONE branch (box workgroups and class workgroups share the grid: 375 KB of weights do not fit one CU's registers,
//     156 / 220 KB do);  every wave has a fixed ROLE and loads its weights into registers ONCE:
//       stage A (first 3x3, from the x ring)  and  stage B (second 3x3, from the t1 ring): a wave owns TWO (or the odd one) of the stage's
//       16-channel n-tiles and all three 16-pixel units of the stage's 2-row band: one ds_read_b128 feeds two MFMAs (half the LDS array time);
//       tail wave: 1x1 from the t2 ring + the branch's half of the decode (csrc/detect_epi.h) on the three units of the output band;
//       DMA wave: stages the input band two steps ahead by global_load_lds straight into the PLANAR x ring (one instruction = one 8-channel
//       plane of a 2 x 32-slot band; out-of-image slots read a zero page: the 3x3's zero padding), counted s_waitcnt vmcnt.
//   * the intermediates t1 (22 columns) and t2 (20 columns) only exist as LDS rings of 8 / 4 rows, planar [8-channel group][row][column][16 B]:
//     a 3x3 tap is an immediate offset of one ds_read_b128.
//   * class branch, 80 channels = 2 k-tiles of 32 + one of 16: the 16-wide remainder runs on v_mfma_f32_16x16x16_bf16 (ds_read_b64 operands,
//     the low / high half of the packed third k-tile as A) AFTER the 32-wide chain of a unit (one 8-pass -> 4-pass transition per chain, fenced).
// Rounding points are those of the separate launches (bf16 t1, t2; f32 accumulation from the bias; f32 decode); the f32 summation order inside a
// convolution differs (k-tiles of all taps first, then the 16-wide remainders), so results equal the tile form's up to flipped bf16 ties.
#include <stdlib.h>

#include "common.h"
#include "detect_epi.h"

typedef __attribute__((address_space(1))) const void* ds_gptr_t;
typedef __attribute__((address_space(3))) void* ds_lptr_t;
typedef __attribute__((ext_vector_type(4))) short ds_s16x4;

__device__ __attribute__((aligned(64))) unsigned int g_ds_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct DsBranch {
  const char *w1, *w2, *wt;     // upa_pack_conv_weight layouts: 3x3 64 -> C, 3x3 C -> C, 1x1 C -> 16 NTT
  const float *b1, *b2, *bt;    // f32, padded to a multiple of 16
};
struct DsParams {
  const char* x;
  int N, H, W, ldx;
  DsBranch br[2];               // [0] box (C = 64), [1] class (C = 80)
  DetectEpi de;
  int strips, parts, L;         // workgroups per branch = N * parts * strips; L = output rows per part (even)
  int ncls_wg;                  // class workgroups come first in the grid (they run longer)
};

// profiling build (-DUPA_STAMP): every wave of the first 4 class and the first 4 box workgroups records s_memtime at the start of each step
// and before its barrier (tools/experiments/r06_dstream_stamps.py)
#ifdef UPA_STAMP
#define DS_STAMP_STEPS 48
__device__ unsigned long long g_ds_stamps[8 * 8 * DS_STAMP_STEPS * 2];
extern "C" int upa_debug_stamps_dstream(unsigned long long* out, int count) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ds_stamps), (size_t)count * 8) == hipSuccess ? 0 : -1;
}
#define DS_SLOT(p) ((int)blockIdx.x < 4 ? (int)blockIdx.x : ((int)blockIdx.x >= (p).ncls_wg && (int)blockIdx.x < (p).ncls_wg + 4 ? 4 + (int)blockIdx.x - (p).ncls_wg : -1))
#define DS_STAMP(slot, step, which)                                                                          \
  do {                                                                                                       \
    if ((slot) >= 0 && (step) < DS_STAMP_STEPS) {                                                            \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      if ((threadIdx.x & 63) == 0) g_ds_stamps[(((slot) * 8 + (threadIdx.x >> 6)) * DS_STAMP_STEPS + (step)) * 2 + (which)] = t_; \
    }                                                                                                        \
  } while (0)
#else
#define DS_SLOT(p) 0
#define DS_STAMP(slot, step, which) do {} while (0)
#endif

namespace dstream {
constexpr int WS = 20;            // output columns of a strip
constexpr int W1 = WS + 2;        // columns of t1
constexpr int XSLOTS = 32;        // slots (16 B) per x-ring row: 24 columns + 8 pad, so that a 2-row band of one plane is ONE LDS-DMA instruction
constexpr int XROWB = XSLOTS * 16;
constexpr int XROWS = 8;          // four bands: two being read, one landed, one landing
constexpr int XPLANE = XROWS * XROWB;
constexpr int XB = 0;
constexpr int XBYTES = 8 * XPLANE;
constexpr int TROWB = 24 * 16;    // t1 / t2 row pitch
constexpr int T1ROWS = 8, T2ROWS = 4;
constexpr int T1PLANE = T1ROWS * TROWB, T2PLANE = T2ROWS * TROWB;
constexpr int T1B = XBYTES;
template <int C> struct Geo {
  static constexpr int CP = C / 8;                  // 8-channel planes
  static constexpr int NT = C / 16;                 // n-tiles of the 3x3 stages
  static constexpr int KT32 = C / 32;               // whole 32-wide k-tiles of stage B / the tail
  static constexpr int K16 = (C % 32) ? 1 : 0;      // ... and a 16-wide remainder
  static constexpr int KTP = KT32 + K16;            // k-tiles in the packed weights
  static constexpr int T2B = T1B + CP * T1PLANE;
  static constexpr int DUMMY = T2B + CP * T2PLANE;  // 512 B nobody reads: where lanes outside a band store
  static constexpr int LDS = DUMMY + 512;
};
static_assert(XPLANE % 256 == 0 && T1PLANE % 256 == 0 && T2PLANE % 256 == 0, "planes keep the ds_read_b128 lane groups on disjoint banks");

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const ds_s16x4*>(&a), *reinterpret_cast<const ds_s16x4*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ void shape_fence() {  // between MFMA shapes on one accumulator chain (see c2f_stream.hip: f_role)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ u32x4 lds128(const char* sm, int off) { return *reinterpret_cast<const u32x4*>(sm + off); }
__device__ __forceinline__ u32x2 lds64(const char* sm, int off) { return *reinterpret_cast<const u32x2*>(sm + off); }

// Geometry of one 3x3 stage.  STG 0: x ring -> t1 (22 columns, rows {2s - 1, 2s} at step s);  STG 1: t1 ring -> t2 (20 columns, rows {2s - 4, 2s - 3}).
template <int C, int STG> struct StageGeo {
  using G = Geo<C>;
  static constexpr int SD = STG ? WS : W1;                       // output columns
  static constexpr int LAG = STG ? 4 : 1;                        // first output row of step s = 2 s - LAG
  static constexpr int IN_B = STG ? T1B : XB;
  static constexpr int IN_PLANE = STG ? T1PLANE : XPLANE;
  static constexpr int IN_ROWB = STG ? TROWB : XROWB;
  static constexpr int IN_MASK = (STG ? T1ROWS : XROWS) - 1;
  static constexpr int KT32 = STG ? G::KT32 : 2;                 // the level's input has 64 channels
  static constexpr int K16 = STG ? G::K16 : 0;
  static constexpr int KTP = KT32 + K16;
  static constexpr int OUT_B = STG ? G::T2B : T1B;
  static constexpr int OUT_PLANE = STG ? T2PLANE : T1PLANE;
  static constexpr int OUT_MASK = (STG ? T2ROWS : T1ROWS) - 1;
  static constexpr int LO = STG ? 2 : 1;                         // valid output rows [LO, LP - LO)
  static constexpr int NU = (2 * SD + 15) / 16;                  // 16-pixel units of the 2-row band
};

// ---- a 3x3 stage: this wave owns n-tiles [nt0, nt0 + NTW) of the stage and every unit of its band, weights in registers for the life of the workgroup
template <int C, int STG, int NTW>
__device__ __forceinline__ void conv_role(const DsParams& p, const DsBranch& br, char* sm, int nt0, int lane, int S, int py0, int sx0, int LP) {
  using G = Geo<C>;
  using SG = StageGeo<C, STG>;
  constexpr int NU = SG::NU, KT32 = SG::KT32, NF = 9 * KT32;  // NF = 32-wide fragments per unit
  constexpr int NBUF = SG::K16 ? 4 : 6;
  const int g = lane >> 4, r = lane & 15;
  const char* wp = STG ? br.w2 : br.w1;
  const float* bp = STG ? br.b2 : br.b1;

  u32x4 w32[9][KT32][NTW];
  u32x2 w16[SG::K16 ? 9 : 1][NTW];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int kt = 0; kt < KT32; ++kt)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        w32[tap][kt][j] = *reinterpret_cast<const u32x4*>(wp + ((size_t)((tap * SG::KTP + kt) * G::NT + nt0 + j) * 64 + lane) * 16);
    if constexpr (SG::K16 != 0) {
      // the 16-wide remainder: lane (g, r) needs W[co = 16 nt + r][ci = 32 KT32 + 4g .. + 3] = half (g & 1) of lane (g >> 1, r)'s 16 bytes of k-tile KT32
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        w16[tap][j] = *reinterpret_cast<const u32x2*>(wp + ((size_t)((tap * SG::KTP + KT32) * G::NT + nt0 + j) * 64 + (g >> 1) * 16 + r) * 16 + (g & 1) * 8);
    }
  }
  f32x4 bias[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bias[j] = *reinterpret_cast<const f32x4*>(bp + (nt0 + j) * 16 + 4 * g);

  // lane constants per unit
  int u_rr[NU], u_in[NU];
  unsigned u_colm[NU];
  bool u_act[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = u * 16 + r;
    u_act[u] = q < 2 * SG::SD;
    const int qq = u_act[u] ? q : 0;
    u_rr[u] = qq >= SG::SD ? 1 : 0;
    const int cc = qq - u_rr[u] * SG::SD;                       // output column; tap (dy, dx) reads input column cc + dx (both stages)
    u_in[u] = SG::IN_B + g * SG::IN_PLANE + cc * 16;
    const int gx = sx0 - 1 + cc;                                // (only stage A's output lies outside the strip's own columns)
    u_colm[u] = (STG || (gx >= 0 && gx < p.W)) ? 0xFFFFFFFFu : 0u;
  }
  const int in16 = (8 + (g >> 1) - g) * SG::IN_PLANE + (g & 1) * 8;   // 16-wide operand: plane 8 + (g >> 1), half (g & 1), relative to u_in
  // the lane's 8 output bytes sit at a lane-constant distance from its input address (same column, another ring)
  const int out_d = SG::OUT_B - SG::IN_B + (2 * nt0 + (g >> 1)) * SG::OUT_PLANE - g * SG::IN_PLANE + (g & 1) * 8;

  for (int s = 0; s < S; ++s) {
    DS_STAMP(DS_SLOT(p), s, 0);
    const int r0 = 2 * s - SG::LAG;
    if (r0 + 2 > SG::LO && r0 < LP - SG::LO) {  // wave-uniform
      int rb[NU][3];
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb[u][dy] = u_in[u] + ((r0 + u_rr[u] + dy - 1) & SG::IN_MASK) * SG::IN_ROWB;
      // fragment t of the flattened (unit, tap, k-tile) order
      auto rd = [&](int t) __attribute__((always_inline)) {
        const int u = t / NF, f = t % NF, tap = f / KT32, kt = f % KT32;
        return lds128(sm, rb[u][tap / 3] + (tap % 3) * 16 + kt * 4 * SG::IN_PLANE);
      };
      constexpr int NTOT = NU * NF;
      u32x4 buf[NBUF];
#pragma unroll
      for (int t = 0; t < NBUF; ++t) buf[t] = rd(t);
      // Software pipeline inside the wave: a wave issues in order, so the SiLU / pack / store epilogue of unit u - 1 is written BEFORE the
      // MFMA chain of unit u and the scheduler is told (sched_group_barrier) to deal it out between that chain's MFMAs - the matrix pipe
      // works on unit u while the vector unit finishes unit u - 1.  (Measured before: a 3x3 wave alone on its SIMD took 3.9 k cycles per
      // step for 1.7 k cycles of MFMA work - chain, then epilogue, strictly one after the other.)
      f32x4 acc[2][NTW];
      // the epilogue of one unit in 5 NTW slices: slice k < 4 NTW = SiLU of one accumulator value, the last NTW = pack + mask + store of an n-tile
      // (an unconditional store - lanes outside the band write a scratch slot - keeps the step one basic block)
      float sv[NTW][4];
      unsigned e_m = 0u;
      int e_oa = 0;
      bool e_ok = false;
      auto epi_slice = [&](int u, int k) __attribute__((always_inline)) {
        if (k == 0) {
          const int row = r0 + u_rr[u];
          const int gy = py0 - 2 + row;
          e_m = (STG || (gy >= 0 && gy < p.H)) ? u_colm[u] : 0u;  // t1 is ZERO outside the image (stage B's padding)
          e_ok = u_act[u] && row >= SG::LO && row < LP - SG::LO;
          e_oa = u_in[u] + out_d + (row & SG::OUT_MASK) * TROWB;
        }
        if (k < 4 * NTW) {
          sv[k >> 2][k & 3] = silu(acc[u & 1][k >> 2][k & 3]);
        } else if (k < 5 * NTW) {
          const int j = k - 4 * NTW;
          const u32x2 o = u32x2{pack_bf16x2(sv[j][0], sv[j][1]) & e_m, pack_bf16x2(sv[j][2], sv[j][3]) & e_m};
          *reinterpret_cast<u32x2*>(sm + (e_ok ? e_oa + 2 * j * SG::OUT_PLANE : G::DUMMY + lane * 8)) = o;
        }
      };
      static_assert(5 * NTW <= NF, "the epilogue slices of a unit fit under the next unit's fragments");
#pragma unroll
      for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[u & 1][j] = bias[j];
        u32x2 b16[3];  // the 16-wide operands, one tap row at a time (the first row's arrive under the 32-wide chain)
        if constexpr (SG::K16 != 0) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) b16[dx] = lds64(sm, rb[u][0] + dx * 16 + in16);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int t = u * NF + f;
          const u32x4 b = buf[t % NBUF];
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[u & 1][j] = mfma32(w32[f / KT32][f % KT32][j], b, acc[u & 1][j]);
          if (t + NBUF < NTOT) buf[t % NBUF] = rd(t + NBUF);
          // a wave issues in order: the previous unit's epilogue is dealt out between this unit's MFMAs, pinned, so that the matrix pipe works
          // on unit u while the vector unit finishes unit u - 1 (before: chain, then epilogue - 3.9 k cycles per step for 1.7 k of MFMA work)
          if (u > 0) epi_slice(u - 1, f);
          __builtin_amdgcn_sched_barrier(0);  // (unit 0 as well: left alone, the scheduler sinks the ring reads down to their MFMAs)
        }
        if constexpr (SG::K16 != 0) {
          shape_fence();
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            u32x2 nx[3];
            if (dy < 2) {
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) nx[dx] = lds64(sm, rb[u][dy + 1] + dx * 16 + in16);
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int j = 0; j < NTW; ++j) acc[u & 1][j] = mfma16(w16[dy * 3 + dx][j], b16[dx], acc[u & 1][j]);
            if (dy < 2) {
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) b16[dx] = nx[dx];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 5 * NTW; ++k) epi_slice(NU - 1, k);
    }
    DS_STAMP(DS_SLOT(p), s, 1);
    __syncthreads();
  }
}

// ---- the tail: 1x1 conv from the t2 ring (rows {2s - 6, 2s - 5} at step s) + the branch's half of the decode, all three units of the output band
// (this wave: units [U0, U0 + NU) of the band's three - the decode is vector work, so a branch deals its units to waves on different SIMDs)
template <int C, int KIND, int U0, int NU>
__device__ __forceinline__ void tail_role(const DsParams& p, const DsBranch& br, char* sm, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<C>;
  constexpr int NTT = KIND == 1 ? 4 : 5;    // output n-tiles: 4 sides x 16 bins | up to 80 classes
  static_assert(U0 + NU <= (2 * WS + 15) / 16, "unit range");
  const int g = lane >> 4, r = lane & 15;
  u32x4 w32[G::KT32][NTT];
  u32x2 w16[NTT];
#pragma unroll
  for (int j = 0; j < NTT; ++j) {
#pragma unroll
    for (int kt = 0; kt < G::KT32; ++kt) w32[kt][j] = *reinterpret_cast<const u32x4*>(br.wt + ((size_t)(kt * NTT + j) * 64 + lane) * 16);
    if constexpr (G::K16 != 0)
      w16[j] = *reinterpret_cast<const u32x2*>(br.wt + ((size_t)(G::KT32 * NTT + j) * 64 + (g >> 1) * 16 + r) * 16 + (g & 1) * 8);
  }
  f32x4 bias[NTT];
#pragma unroll
  for (int j = 0; j < NTT; ++j) bias[j] = *reinterpret_cast<const f32x4*>(br.bt + j * 16 + 4 * g);
  int u_rr[NU], u_in[NU], u_gx[NU];
  bool u_act[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = (U0 + u) * 16 + r;
    u_act[u] = q < 2 * WS;
    const int qq = u_act[u] ? q : 0;
    u_rr[u] = qq >= WS ? 1 : 0;
    const int cc = qq - u_rr[u] * WS;
    u_in[u] = G::T2B + g * T2PLANE + cc * 16;
    u_gx[u] = sx0 + cc;
    u_act[u] = u_act[u] && u_gx[u] < p.W;
  }
  const int in16 = (8 + (g >> 1) - g) * T2PLANE + (g & 1) * 8;

  for (int s = 0; s < S; ++s) {
    DS_STAMP(DS_SLOT(p), s, 0);
    const int r0 = 2 * s - 6;
    if (r0 + 2 > 2 && r0 < LP - 2) {  // wave-uniform
      u32x4 b32[NU][G::KT32];
      u32x2 b16[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int a = u_in[u] + ((r0 + u_rr[u]) & (T2ROWS - 1)) * TROWB;
#pragma unroll
        for (int kt = 0; kt < G::KT32; ++kt) b32[u][kt] = lds128(sm, a + kt * 4 * T2PLANE);
        if constexpr (G::K16 != 0) b16[u] = lds64(sm, a + in16);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x4 v[NTT];
#pragma unroll
        for (int j = 0; j < NTT; ++j) v[j] = bias[j];
#pragma unroll
        for (int kt = 0; kt < G::KT32; ++kt)
#pragma unroll
          for (int j = 0; j < NTT; ++j) v[j] = mfma32(w32[kt][j], b32[u][kt], v[j]);
        if constexpr (G::K16 != 0) {
          shape_fence();
#pragma unroll
          for (int j = 0; j < NTT; ++j) v[j] = mfma16(w16[j], b16[u], v[j]);
        }
        const int row = r0 + u_rr[u];
        const int gy = py0 - 2 + row;
        const bool ok = u_act[u] && row >= 2 && row < LP - 2 && gy < p.H;
        const int al = ok ? gy * p.W + u_gx[u] : 0;  // level-local anchor
        if constexpr (KIND == 1) {
          upa_detect_box_store(p.de, v, n, al, ok, g);
        } else {
          float best = -1.f;
          int bc = 0;
          if (p.de.keys_only) {  // uniform
            upa_detect_cls_keys_only<NTT>(p.de, v, ok, g, best, bc);
          } else {
#pragma unroll
            for (int j = 0; j < NTT; ++j) upa_detect_cls_store(p.de, v[j], j, n, al, ok, g, best, bc);
          }
          if (p.de.best_keys) upa_detect_best_key_store(p.de, best, bc, n, al, ok, lane);  // uniform
        }
      }
    }
    DS_STAMP(DS_SLOT(p), s, 1);
    __syncthreads();
  }
}

// ---- the input band two steps ahead: band b = x rows {2b, 2b + 1} (image rows py0 - 2 + ...), 8 planes x (2 rows x 32 slots) = 8 LDS-DMA instructions;
// lane = slot (row lane >> 5, column lane & 31); out-of-image slots (and the 8 pad slots of a row) read the zero page
__device__ __forceinline__ void dma_role(const DsParams& p, char* sm, int lane, int S, int n, int py0, int sx0, int LP) {
  const unsigned rowpitch = (unsigned)p.W * (unsigned)p.ldx * 2u;
  const char* ximg = p.x + (size_t)n * p.H * rowpitch;
  const int rr = lane >> 5, xc = lane & 31;
  const int gx = sx0 - 2 + xc;
  const bool colok = xc < WS + 4 && gx >= 0 && gx < p.W;
  const unsigned coloff = colok ? (unsigned)gx * (unsigned)p.ldx * 2u : 0u;
  const char* zp = reinterpret_cast<const char*>(g_ds_zero_page);
  auto stage_in = [&](int b) __attribute__((always_inline)) {
    if (2 * b >= LP) return 0;  // wave-uniform
    const int gy = py0 - 2 + 2 * b + rr;
    const bool ok = colok && gy >= 0 && gy < p.H;
    const char* src = ok ? ximg + ((unsigned)gy * rowpitch + coloff) : zp;
    const int dst = XB + ((2 * b) & (XROWS - 1)) * XROWB;  // + lane * 16 by the hardware
#pragma unroll
    for (int cg = 0; cg < 8; ++cg)
      __builtin_amdgcn_global_load_lds((ds_gptr_t)(src + (ok ? cg * 16 : 0)), (ds_lptr_t)(sm + dst + cg * XPLANE), 16, 0, 0);
    return 8;
  };
  stage_in(0);
  stage_in(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    DS_STAMP(DS_SLOT(p), s, 0);
    const int inflight = stage_in(s + 2);
    // band s + 1 has landed once everything but this step's requests is back (a DMA wave issues no other vector-memory operation)
    if (inflight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DS_STAMP(DS_SLOT(p), s, 1);
    __syncthreads();
  }
}

__device__ __forceinline__ void idle_role(int S) {
  for (int s = 0; s < S; ++s) __syncthreads();
}
}  // namespace dstream

__global__ __launch_bounds__(512) void detect_stream_kernel(const DsParams p) {
  using namespace dstream;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = (int)blockIdx.x;
  const bool cls = bid < p.ncls_wg;
  if (!cls) bid -= p.ncls_wg;
  const int n = bid / (p.parts * p.strips);
  bid -= n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  const int py0 = part * p.L, sx0 = strip * WS;
  int leff = p.H - py0 < p.L ? p.H - py0 : p.L;
  leff = (leff + 1) & ~1;
  const int LP = leff + 4;          // x rows of this part
  const int S = leff / 2 + 4;       // steps until the last output row has left

  if (wave != 5) __syncthreads();   // (the DMA wave arrives at this barrier with the first two bands landed)
  // waves w and w + 4 share a SIMD.  MFMAs (32-wide equivalents) per step:
  //   class:  SIMD 0  B(0-1) 135 + tail 38    SIMD 1  B(2-3) 135 + DMA    SIMD 2  A(0-1) 108 + A(4) 54    SIMD 3  A(2-3) 108 + B(4) 68
  //   box:    SIMD 0  A(0-1) 108 + tail unit 0    SIMD 1  A(2-3) 108 + DMA    SIMD 2  B(0-1) 108 + tail unit 1    SIMD 3  B(2-3) 108 + tail unit 2
  if (cls) {
    const DsBranch& br = p.br[1];
    switch (wave) {
      case 0: conv_role<80, 1, 2>(p, br, sm, 0, lane, S, py0, sx0, LP); break;
      case 1: conv_role<80, 1, 2>(p, br, sm, 2, lane, S, py0, sx0, LP); break;
      case 2: conv_role<80, 0, 2>(p, br, sm, 0, lane, S, py0, sx0, LP); break;
      case 3: conv_role<80, 0, 2>(p, br, sm, 2, lane, S, py0, sx0, LP); break;
      case 4: tail_role<80, 2, 0, 3>(p, br, sm, lane, S, n, py0, sx0, LP); break;
      case 5: dma_role(p, sm, lane, S, n, py0, sx0, LP); break;
      case 6: conv_role<80, 0, 1>(p, br, sm, 4, lane, S, py0, sx0, LP); break;
      default: conv_role<80, 1, 1>(p, br, sm, 4, lane, S, py0, sx0, LP); break;
    }
  } else {
    const DsBranch& br = p.br[0];
    switch (wave) {
      case 0: conv_role<64, 0, 2>(p, br, sm, 0, lane, S, py0, sx0, LP); break;
      case 1: conv_role<64, 0, 2>(p, br, sm, 2, lane, S, py0, sx0, LP); break;
      case 2: conv_role<64, 1, 2>(p, br, sm, 0, lane, S, py0, sx0, LP); break;
      case 3: conv_role<64, 1, 2>(p, br, sm, 2, lane, S, py0, sx0, LP); break;
      case 4: tail_role<64, 1, 0, 1>(p, br, sm, lane, S, n, py0, sx0, LP); break;
      case 5: dma_role(p, sm, lane, S, n, py0, sx0, LP); break;
      case 6: tail_role<64, 1, 1, 1>(p, br, sm, lane, S, n, py0, sx0, LP); break;
      default: tail_role<64, 1, 2, 1>(p, br, sm, lane, S, n, py0, sx0, LP); break;
    }
  }
}

// One level of a Detect head, both branches, in one launch (see the header of this file).  UPA_EUNSUPPORTED outside the form.
extern "C" int upa_detect_level_stream(const void* x, int n, int h, int w, int cin, int ldx, const upa_detect_branch* box,
                                       const upa_detect_branch* cls, int nc, float stride_px, float* y, int a_total, int a0,
                                       unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && box && cls && y, "detect_level_stream: null pointer");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && a0 >= 0 && a0 + h * w <= a_total, "detect_level_stream: level does not fit a_total");
  UPA_CHECK_ARG(box->w1 && box->w2 && box->wt && box->b1 && box->b2 && box->bt && cls->w1 && cls->w2 && cls->wt && cls->b1 && cls->b2 && cls->bt,
                "detect_level_stream: null weight pointer");
  const int off = UPA_OPT(opts, detect_stream);  // 2 = on; the default dispatch keeps the tile form (see the measurement in the header)
  if (off != 2 || dtype != UPA_BF16 || cin != 64 || box->c != 64 || cls->c != 80 || nc > 80 || nc < 1 || ldx % 8 != 0 || ((uintptr_t)x % 16) != 0 ||
      w < 8 || h < 2 || !upa_magic_exact((long)h * w - 1, w) || (long)h * w * ldx * 2 >= (1L << 31)) {
    upa_set_error("detect_level_stream: outside the line-buffer form (bf16, 64 input channels, box c = 64, class c = 80, nc <= 80)");
    return UPA_EUNSUPPORTED;
  }
  DsParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.N = n; p.H = h; p.W = w; p.ldx = ldx;
  const upa_detect_branch* src[2] = {box, cls};
  for (int k = 0; k < 2; ++k) {
    p.br[k].w1 = (const char*)src[k]->w1; p.br[k].w2 = (const char*)src[k]->w2; p.br[k].wt = (const char*)src[k]->wt;
    p.br[k].b1 = src[k]->b1; p.br[k].b2 = src[k]->b2; p.br[k].bt = src[k]->bt;
  }
  p.de.y = y; p.de.a_total = a_total; p.de.a0 = a0; p.de.HW = h * w; p.de.W = w;
  p.de.magicHW = upa_magic_div(h * w); p.de.magicW = upa_magic_div(w);
  p.de.nc = nc; p.de.stride_px = stride_px;
  if (best_keys && (long)a_total * nc < (1L << 31)) p.de.best_keys = best_keys;
  p.de.keys_only = (p.de.best_keys && UPA_OPT(opts, keys_only)) ? 1 : 0;
  p.strips = cdiv(w, dstream::WS);
  // rows per part: the whole height (fewest pipeline fills: least CU time) unless the caller asks for parts (`detect_stream_rows`: even >= 4)
  int rows = UPA_OPT(opts, detect_stream_rows);
  UPA_CHECK_ARG(rows == 0 || rows >= 4, "detect_stream_rows = %d: 0 (whole height) or >= 4", rows);
  int L = (h + 1) & ~1;
  if (rows >= 4) L = rows & ~1;
  if (L > ((h + 1) & ~1)) L = (h + 1) & ~1;
  p.L = L;
  p.parts = cdiv(h, L);
  p.ncls_wg = n * p.parts * p.strips;
  const int grid = 2 * p.ncls_wg;
  const size_t lds = dstream::Geo<80>::LDS;
  if (hipError_t e = upa_full_lds<detect_stream_kernel>(); e != hipSuccess) {
    upa_set_error("detect_level_stream: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL(detect_stream_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
