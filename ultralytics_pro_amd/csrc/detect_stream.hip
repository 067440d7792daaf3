// One level of the Detect head (yolov8n's 80 x 80 level: 64 input channels, box branch 64 -> 64 -> 64, class branch 64 -> 80 -> 80 -> nc) as
// ONE launch in LINE-BUFFER form: conv3x3 -> conv3x3 -> 1x1 -> DFL / dist2bbox / sigmoid -> decoded rows + best-class NMS keys.
//   Detect.__init__ / forward   ultralytics/nn/modules/head.py:94-100, 116-126   cv2[i] = Conv(x, c2, 3), Conv(c2, c2, 3), Conv2d(c2, 4 reg_max, 1)
//                                                                                cv3[i] = Conv(x, c3, 3), Conv(c3, c3, 3), Conv2d(c3, nc, 1)
//   Detect._inference           ultralytics/nn/modules/head.py:151-191           dfl, dist2bbox * stride, sigmoid
// The tile form of this level (conv_big.hip: a stacked 144-channel first conv, then conv_big_mix<..5,2 | ..4,1> = second 3x3 + 1x1 + decode)
// writes and re-reads the 144-channel intermediate (236 MB per batch of 32), DMAs every 3x3 weight slab into LDS once per 256-pixel tile and
// pays a halo wait + epilogue per tile: 49 + 70 us at 0.2-0.28 of the MFMA peak.  Here, as in c2f_stream.hip, a workgroup owns a vertical
// STRIP of one image (WS = 30 output columns, L rows) and walks down it two rows per step with ONE s_barrier per step:
//   * a workgroup runs ONE branch (box workgroups and class workgroups share the grid: 375 KB of weights do not fit one CU's registers,
//     156 / 220 KB do);  every wave has a fixed ROLE and loads its weights into registers ONCE:
//       stage A (first 3x3, from the x ring) and stage B (second 3x3, from the t1 ring) on v_mfma_f32_32x32x16_bf16: a wave owns 32 output
//       channels of its stage, a unit = one row of the band = 32 pixels (t1 is exactly 32 columns wide), one ds_read_b128 per MFMA, the epilogue of
//       row 0 dealt out between the MFMAs of row 1 (pinned); the class branch's odd 16 channels (80 = 32 + 32 + 16) as a 16x16x32-form wave;
//       tail waves: 1x1 from the t2 ring + the branch's half of the decode (csrc/detect_epi.h), the band's four 16-pixel units dealt to waves on
//       different SIMDs; one of them also stages the input band two steps ahead by global_load_lds straight into the PLANAR x ring (one
//       instruction = one row of one 8-channel plane; out-of-image slots read a zero page: the 3x3's zero padding).
//   * the intermediates t1 (32 columns) and t2 (30 columns) only exist as LDS rings of 8 / 4 rows, planar [8-channel group][row][column][16 B]:
//     a 3x3 tap is an immediate offset of one ds_read_b128.
//   * 80 input channels in the 16x16x32 form = 2 k-tiles of 32 + one of 16: the remainder runs on v_mfma_f32_16x16x16_bf16 (ds_read_b64 operands,
//     the low / high half of the packed third k-tile as A) AFTER the 32-wide chain of a unit (one 8-pass -> 4-pass transition per chain, fenced).
// Rounding points are those of the separate launches (bf16 t1, t2; f32 accumulation from the bias; f32 decode); the f32 summation order inside a
// convolution differs, so results equal the tile form's up to flipped bf16 ties.
//
// MEASURED SLOWER than the tile form in both forms built in round 6 (first: 20-column strips on 16x16x32 MFMAs; this one: 30-column strips on
// 32x32x16) - profiles/r06_detect_stream.txt holds the per-wave stamps: with two waves per SIMD every role is bound by its own in-order issue
// (a 32-wide 3x3 wave: 72 MFMAs of 32 cycles in 3.9-4.1 k cycles alone on its SIMD; a box tail: 1.8 k per 16-pixel unit alone; the class
// tails 1.85 k per unit), the class workgroup needs ~27 k issue cycles per step over four SIMDs.  Opt-in only: upa_opts.detect_stream = 2.
// -DDS_EXP=1 | 2 (timing experiments, wrong results): tails / 32-wide 3x3 stages do nothing.
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
constexpr int WS = 30;            // output columns of a strip
constexpr int W1 = WS + 2;        // columns of t1 = 32: one band row IS one 32-pixel unit of v_mfma_f32_32x32x16_bf16
constexpr int XW = WS + 4;        // input columns of a strip
constexpr int XSLOTS = 64;        // slots (16 B) per x-ring row: 34 columns + pad, so that a row of one plane is ONE LDS-DMA instruction
constexpr int XROWB = XSLOTS * 16;
constexpr int XROWS = 8;          // four bands: two being read, one landed, one landing
constexpr int XPLANE = XROWS * XROWB;
constexpr int XB = 0;
constexpr int XBYTES = 8 * XPLANE;
constexpr int TROWB = 32 * 16;    // t1 / t2 row pitch
constexpr int T1ROWS = 8, T2ROWS = 4;
constexpr int T1PLANE = T1ROWS * TROWB, T2PLANE = T2ROWS * TROWB;
constexpr int T1B = XBYTES;
template <int C> struct Geo {
  static constexpr int CP = C / 8;                  // 8-channel planes
  static constexpr int NT = C / 16;                 // 16-channel n-tiles in the packed 3x3 weights
  static constexpr int KT32 = C / 32;               // whole 32-wide k-tiles of stage B / the tail
  static constexpr int K16 = (C % 32) ? 1 : 0;      // ... and a 16-wide remainder
  static constexpr int KTP = KT32 + K16;            // k-tiles in the packed weights
  static constexpr int T2B = T1B + CP * T1PLANE;
  static constexpr int DUMMY = T2B + CP * T2PLANE;  // 1 KB nobody reads: where lanes outside a band store
  static constexpr int LDS = DUMMY + 1024;
};
static_assert(XPLANE % 256 == 0 && T1PLANE % 256 == 0 && T2PLANE % 256 == 0, "planes keep the ds_read_b128 lane groups on disjoint banks");

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const ds_s16x4*>(&a), *reinterpret_cast<const ds_s16x4*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma3232(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ void shape_fence() {  // between MFMA shapes on one accumulator chain (see c2f_stream.hip: f_role)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ u32x4 lds128(const char* sm, int off) { return *reinterpret_cast<const u32x4*>(sm + off); }
__device__ __forceinline__ u32x2 lds64(const char* sm, int off) { return *reinterpret_cast<const u32x2*>(sm + off); }

// Geometry of one 3x3 stage.  STG 0: x ring -> t1 (32 columns, rows {2s - 1, 2s} at step s);  STG 1: t1 ring -> t2 (30 columns, rows {2s - 4, 2s - 3}).
template <int C, int STG> struct StageGeo {
  using G = Geo<C>;
  static constexpr int SD = STG ? WS : W1;                       // output columns
  static constexpr int LAG = STG ? 4 : 1;                        // first output row of step s = 2 s - LAG
  static constexpr int IN_B = STG ? T1B : XB;
  static constexpr int IN_PLANE = STG ? T1PLANE : XPLANE;
  static constexpr int IN_ROWB = STG ? TROWB : XROWB;
  static constexpr int IN_MASK = (STG ? T1ROWS : XROWS) - 1;
  static constexpr int CIN = STG ? C : 64;                       // the level's input has 64 channels
  static constexpr int KS16 = CIN / 16;                          // 16-wide k-steps (32x32x16 form)
  static constexpr int KT32 = CIN / 32;                          // 32-wide k-tiles (16x16x32 form) ...
  static constexpr int K16 = (CIN % 32) ? 1 : 0;                 // ... and a 16-wide remainder
  static constexpr int KTP = KT32 + K16;                         // k-tiles in the packed weights
  static constexpr int OUT_B = STG ? G::T2B : T1B;
  static constexpr int OUT_PLANE = STG ? T2PLANE : T1PLANE;
  static constexpr int OUT_MASK = (STG ? T2ROWS : T1ROWS) - 1;
  static constexpr int LO = STG ? 2 : 1;                         // valid output rows [LO, LP - LO)
};

struct Ctx {  // what every role of a workgroup shares
  const DsParams* p;
  char* sm;
  int lane, n, py0, sx0, LP, S, slot;
};

// ---- a 3x3 stage on v_mfma_f32_32x32x16_bf16: this wave owns output channels [32 N, 32 N + 32) of the stage; a unit = one row of the band (32
// pixels), weights in registers for the life of the workgroup.  Lane (h = lane >> 5, c = lane & 31): B operand = 8 channels (plane 2 s + h) of
// pixel c, D = channels 32 N + 8 i + 4 h + j (i, j < 4) of pixel c.  The epilogue of row 0 is dealt out between the MFMAs of row 1 (PIPE).
template <int C, int STG, bool PIPE>
struct Conv32 {
  using G = Geo<C>;
  using SG = StageGeo<C, STG>;
  static constexpr int KS = SG::KS16, NF = 9 * KS, NBUF = PIPE ? 6 : 4;
  u32x4 w[9][KS];
  f32x4 bias[4];
  int in0, out_d, h, c;
  unsigned colm;
  bool act;
  __device__ __forceinline__ void init(const Ctx& x, const DsBranch& br, int N) {
    const int lane = x.lane;
    h = lane >> 5; c = lane & 31;
    const char* wp = STG ? br.w2 : br.w1;
    const float* bp = STG ? br.b2 : br.b1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int s = 0; s < KS; ++s)  // lane (h, c): W[co = 32 N + c][ci = 16 s + 8 h .. + 7] = lane ((s & 1) * 2 + h, c & 15) of n-tile 2 N + (c >> 4), k-tile s >> 1
        w[tap][s] = *reinterpret_cast<const u32x4*>(wp + ((size_t)((tap * SG::KTP + (s >> 1)) * G::NT + 2 * N + (c >> 4)) * 64 + ((s & 1) * 2 + h) * 16 + (c & 15)) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = *reinterpret_cast<const f32x4*>(bp + 32 * N + 8 * i + 4 * h);
    act = c < SG::SD;
    in0 = SG::IN_B + h * SG::IN_PLANE + c * 16;            // tap (dy, dx) reads input column c + dx (both stages)
    out_d = SG::OUT_B - SG::IN_B + 4 * N * SG::OUT_PLANE - h * SG::IN_PLANE + 8 * h;
    const int gx = x.sx0 - 1 + c;                          // (only stage A's output lies outside the strip's own columns)
    colm = (STG || (gx >= 0 && gx < x.p->W)) ? 0xFFFFFFFFu : 0u;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
#if defined(DS_EXP) && DS_EXP == 2  // timing experiment (wrong results): no 32-wide 3x3 work
    return;
#endif
    const int r0 = 2 * s - SG::LAG;
    if (!(r0 + 2 > SG::LO && r0 < x.LP - SG::LO)) return;  // wave-uniform
    char* sm = x.sm;
    int rb[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) rb[u][dy] = in0 + ((r0 + u + dy - 1) & SG::IN_MASK) * SG::IN_ROWB;
    auto rd = [&](int t) __attribute__((always_inline)) {
      const int u = t / NF, f = t % NF, tap = f / KS, ks = f % KS;
      return lds128(sm, rb[u][tap / 3] + (tap % 3) * 16 + ks * 2 * SG::IN_PLANE);
    };
    constexpr int NTOT = 2 * NF;
    u32x4 buf[NBUF];
#pragma unroll
    for (int t = 0; t < NBUF; ++t) buf[t] = rd(t);
    f32x16 acc[PIPE ? 2 : 1];
    float sv[4];
    unsigned e_m = 0u;
    int e_oa = 0;
    bool e_ok = false;
    // the epilogue of one row in 20 slices: slice k < 16 = SiLU of one accumulator value; every fourth one also packs + masks + stores the
    // four channels 8 i + 4 h .. + 3 (an unconditional store - lanes outside the band write a scratch slot - keeps the step one basic block)
    auto epi_slice = [&](int u, int k) __attribute__((always_inline)) {
      const f32x16& a = acc[PIPE ? (u & 1) : 0];
      if (k == 0) {
        const int row = r0 + u;
        const int gy = x.py0 - 2 + row;
        e_m = (STG || (gy >= 0 && gy < x.p->H)) ? colm : 0u;  // t1 is ZERO outside the image (stage B's padding)
        e_ok = act && row >= SG::LO && row < x.LP - SG::LO;
        e_oa = in0 + out_d + (row & SG::OUT_MASK) * TROWB;
      }
      if (k < 16) {
        sv[k & 3] = silu(a[k]);
        if ((k & 3) == 3) {
          const u32x2 o = u32x2{pack_bf16x2(sv[0], sv[1]) & e_m, pack_bf16x2(sv[2], sv[3]) & e_m};
          *reinterpret_cast<u32x2*>(sm + (e_ok ? e_oa + (k >> 2) * SG::OUT_PLANE : G::DUMMY + x.lane * 8)) = o;
        }
      }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16& a = acc[PIPE ? (u & 1) : 0];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * i + j] = bias[i][j];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = u * NF + f;
        a = mfma3232(w[f / KS][f % KS], buf[t % NBUF], a);
        if (t + NBUF < NTOT) buf[t % NBUF] = rd(t + NBUF);
        if (PIPE && u > 0) {  // row 0's sixteen epilogue slices, spread evenly over row 1's MFMAs (a slice is ~28 issue cycles, an MFMA 32 of pipe)
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if ((k * NF) / 16 == f) epi_slice(u - 1, k);
        }
        __builtin_amdgcn_sched_barrier(0);  // pinned: a wave issues in order, and left alone the scheduler sinks the ring reads to their MFMAs
      }
      if (!PIPE || u == 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) epi_slice(u, k);
      }
    }
  }
};

// ---- a 3x3 stage's ODD 16-channel n-tile (class branch: 80 = 32 + 32 + 16) in the 16x16x32 form: units of 16 pixels (two per band row), the 16-wide
// remainder of an 80-channel input on v_mfma_f32_16x16x16_bf16 after the 32-wide chain (fenced)
template <int C, int STG>
struct Conv16 {
  using G = Geo<C>;
  using SG = StageGeo<C, STG>;
  static constexpr int KT32 = SG::KT32, NF = 9 * KT32, NBUF = 3;
  u32x4 w32[9][KT32];
  u32x2 w16[SG::K16 ? 9 : 1];
  f32x4 bias;
  int in0, in16, out_d, g, r;
  unsigned colm[2];
  bool act[2];
  __device__ __forceinline__ void init(const Ctx& x, const DsBranch& br, int nt) {
    const int lane = x.lane;
    g = lane >> 4; r = lane & 15;
    const char* wp = STG ? br.w2 : br.w1;
    const float* bp = STG ? br.b2 : br.b1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int kt = 0; kt < KT32; ++kt) w32[tap][kt] = *reinterpret_cast<const u32x4*>(wp + ((size_t)((tap * SG::KTP + kt) * G::NT + nt) * 64 + lane) * 16);
      if constexpr (SG::K16 != 0)
        w16[tap] = *reinterpret_cast<const u32x2*>(wp + ((size_t)((tap * SG::KTP + KT32) * G::NT + nt) * 64 + (g >> 1) * 16 + r) * 16 + (g & 1) * 8);
    }
    bias = *reinterpret_cast<const f32x4*>(bp + nt * 16 + 4 * g);
    in0 = SG::IN_B + g * SG::IN_PLANE + r * 16;   // unit q of a row: columns 16 q + r = + 256 q bytes
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cc = 16 * q + r;
      act[q] = cc < SG::SD;
      const int gx = x.sx0 - 1 + cc;
      colm[q] = (STG || (gx >= 0 && gx < x.p->W)) ? 0xFFFFFFFFu : 0u;
    }
    in16 = (8 + (g >> 1) - g) * SG::IN_PLANE + (g & 1) * 8;   // 16-wide operand: plane 8 + (g >> 1), half (g & 1), relative to in0
    out_d = SG::OUT_B - SG::IN_B + (2 * nt + (g >> 1)) * SG::OUT_PLANE - g * SG::IN_PLANE + (g & 1) * 8;
  }
  // A band row = two 16-pixel units; both run as ONE pair with two independent accumulator chains (a single chain of dependent 16x16x32
  // MFMAs issues at half the matrix rate), the ring reads of both prefetched NBUF fragments ahead, and the SiLU / pack / store epilogue of
  // row 0 dealt out between the MFMAs of row 1 (pinned), as in Conv32.
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    const int r0 = 2 * s - SG::LAG;
    if (!(r0 + 2 > SG::LO && r0 < x.LP - SG::LO)) return;  // wave-uniform
    char* sm = x.sm;
    int rb[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) rb[u][dy] = in0 + ((r0 + u + dy - 1) & SG::IN_MASK) * SG::IN_ROWB;
    auto rd = [&](int t, int q) __attribute__((always_inline)) {
      const int u = t / NF, f = t % NF, tap = f / KT32, kt = f % KT32;
      return lds128(sm, rb[u][tap / 3] + (tap % 3) * 16 + kt * 4 * SG::IN_PLANE + 256 * q);
    };
    constexpr int NTOT = 2 * NF;
    u32x4 buf[NBUF][2];
#pragma unroll
    for (int t = 0; t < NBUF; ++t) { buf[t][0] = rd(t, 0); buf[t][1] = rd(t, 1); }
    f32x4 acc[2][2];  // [row][unit]
    float sv[4];
    unsigned e_m = 0u;
    int e_oa = 0;
    bool e_row = false;
    auto epi_slice = [&](int u, int k) __attribute__((always_inline)) {  // k < 8: SiLU of value k & 3 of unit k >> 2; every fourth also packs + stores
      if (k == 0) {
        const int row = r0 + u;
        const int gy = x.py0 - 2 + row;
        e_m = (STG || (gy >= 0 && gy < x.p->H)) ? 0xFFFFFFFFu : 0u;  // t1 is ZERO outside the image (stage B's padding)
        e_row = row >= SG::LO && row < x.LP - SG::LO;
        e_oa = in0 + out_d + (row & SG::OUT_MASK) * TROWB;
      }
      const int q = k >> 2;
      sv[k & 3] = silu(acc[u][q][k & 3]);
      if ((k & 3) == 3) {
        const unsigned m = e_m & colm[q];
        const u32x2 o = u32x2{pack_bf16x2(sv[0], sv[1]) & m, pack_bf16x2(sv[2], sv[3]) & m};
        *reinterpret_cast<u32x2*>(sm + ((e_row && act[q]) ? e_oa + 256 * q : G::DUMMY + x.lane * 8)) = o;
      }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      acc[u][0] = bias; acc[u][1] = bias;
      u32x2 b16[2][3];  // the 16-wide operands of both units, one tap row at a time
      if constexpr (SG::K16 != 0) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) { b16[0][dx] = lds64(sm, rb[u][0] + dx * 16 + in16); b16[1][dx] = lds64(sm, rb[u][0] + dx * 16 + in16 + 256); }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = u * NF + f;
        acc[u][0] = mfma32(w32[f / KT32][f % KT32], buf[t % NBUF][0], acc[u][0]);
        acc[u][1] = mfma32(w32[f / KT32][f % KT32], buf[t % NBUF][1], acc[u][1]);
        if (t + NBUF < NTOT) { buf[t % NBUF][0] = rd(t + NBUF, 0); buf[t % NBUF][1] = rd(t + NBUF, 1); }
        if (u > 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if ((k * NF) / 8 == f) epi_slice(u - 1, k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SG::K16 != 0) {
        shape_fence();
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          u32x2 nx[2][3];
          if (dy < 2) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { nx[0][dx] = lds64(sm, rb[u][dy + 1] + dx * 16 + in16); nx[1][dx] = lds64(sm, rb[u][dy + 1] + dx * 16 + in16 + 256); }
          }
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            acc[u][0] = mfma16(w16[dy * 3 + dx], b16[0][dx], acc[u][0]);
            acc[u][1] = mfma16(w16[dy * 3 + dx], b16[1][dx], acc[u][1]);
          }
          if (dy < 2) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { b16[0][dx] = nx[0][dx]; b16[1][dx] = nx[1][dx]; }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) epi_slice(1, k);
  }
};

// keys-only class decode of one 16-pixel tile, lean: top-2 of the lane's logits by v_max_f32 + v_med3_f32 (the median of {largest, second, x} is
// the new second largest), the certainty test of upa_detect_cls_keys_only, then the index of the largest by a descending compare-select scan.
// Same results as upa_detect_cls_keys_only (csrc/detect_epi.h); here the tail waves are vector-issue bound, so the instruction count matters.
template <int NTC>
__device__ __forceinline__ void cls_keys_lean(const DetectEpi& d, const f32x4 (&logit)[NTC], bool ok, int kg, float& best, int& bc) {
  constexpr float LOG2E = 1.44269504088896340736f;
  // (classes >= nc - the zero filters the 1x1 weights are padded with - arrive as -inf: the tail's bias holds -inf there, see Tail::init)
  const f32x4 (&v)[NTC] = logit;
  // two independent (largest, second) chains over the even / odd positions - one chain is 2 x 4 NTC dependent instructions deep - merged at the end
  float a1 = v[0][0], a2 = -INFINITY, b1 = v[0][1], b2 = -INFINITY;
#pragma unroll
  for (int j = 0; j < NTC; ++j)
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      if (j == 0 && q == 0) continue;
      a2 = __builtin_amdgcn_fmed3f(a1, a2, v[j][q]);
      a1 = fmaxf(a1, v[j][q]);
      b2 = __builtin_amdgcn_fmed3f(b1, b2, v[j][q + 1]);
      b1 = fmaxf(b1, v[j][q + 1]);
    }
  const float m1 = fmaxf(a1, b1);
  const float m2 = fmaxf(fminf(a1, b1), fmaxf(a2, b2));
  const float pm = upa_row_max4(m1);
  const bool mine = m1 == pm;
  const float ps = upa_row_max4(mine ? m2 : m1);
  const bool sure = !ok || (pm <= 4.0f && pm - ps >= 1e-4f && upa_row_sum4(mine ? 1.f : 0.f) == 1.f);
  if (__all(sure)) {
    int c1 = 0;
#pragma unroll
    for (int j = NTC - 1; j >= 0; --j)
#pragma unroll
      for (int q = 3; q >= 0; --q) c1 = v[j][q] == m1 ? 16 * j + q : c1;
    best = ok && mine ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m1 * -LOG2E)) : -1.f;
    bc = c1 + 4 * kg;
    return;
  }
  upa_detect_cls_keys_only<NTC>(d, logit, ok, kg, best, bc);
}

// ---- the tail: 1x1 conv from the t2 ring (rows {2s - 6, 2s - 5} at step s) + the branch's half of the decode on units [U0, U0 + NU) of the
// band's four 16-pixel units (unit U = row U >> 1, columns 16 (U & 1) + r); the decode is vector work, so a branch deals its units to waves on
// different SIMDs
template <int C, int KIND, int U0, int NU>
struct Tail {
  using G = Geo<C>;
  static constexpr int NTT = KIND == 1 ? 4 : 5;    // output n-tiles: 4 sides x 16 bins | up to 80 classes
  u32x4 w32[G::KT32][NTT];
  u32x2 w16[G::K16 ? NTT : 1];
  f32x4 bias[NTT];
  int u_in[NU], u_gx[NU], in16, g, r;
  bool u_act[NU];
  __device__ __forceinline__ void init(const Ctx& x, const DsBranch& br) {
    const int lane = x.lane;
    g = lane >> 4; r = lane & 15;
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
#pragma unroll
      for (int kt = 0; kt < G::KT32; ++kt) w32[kt][j] = *reinterpret_cast<const u32x4*>(br.wt + ((size_t)(kt * NTT + j) * 64 + lane) * 16);
      if constexpr (G::K16 != 0)
        w16[j] = *reinterpret_cast<const u32x2*>(br.wt + ((size_t)(G::KT32 * NTT + j) * 64 + (g >> 1) * 16 + r) * 16 + (g & 1) * 8);
      bias[j] = *reinterpret_cast<const f32x4*>(br.bt + j * 16 + 4 * g);
      if constexpr (KIND == 2) {  // a class beyond nc (zero filters) must never be the best one: its logit is -inf from the start
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (16 * j + 4 * g + q >= x.p->de.nc) bias[j][q] = -INFINITY;
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int cc = 16 * ((U0 + u) & 1) + r;
      u_in[u] = G::T2B + g * T2PLANE + cc * 16;
      u_gx[u] = x.sx0 + cc;
      u_act[u] = cc < WS && u_gx[u] < x.p->W;
    }
    in16 = (8 + (g >> 1) - g) * T2PLANE + (g & 1) * 8;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
#if defined(DS_EXP) && DS_EXP == 1  // timing experiment (wrong results): no tail work
    return;
#endif
    const int r0 = 2 * s - 6;
    if (!(r0 + 2 > 2 && r0 < x.LP - 2)) return;  // wave-uniform
    const DsParams& p = *x.p;
    const char* sm = x.sm;
    u32x4 b32[NU][G::KT32];
    u32x2 b16[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int a = u_in[u] + ((r0 + ((U0 + u) >> 1)) & (T2ROWS - 1)) * TROWB;
#pragma unroll
      for (int kt = 0; kt < G::KT32; ++kt) b32[u][kt] = lds128(sm, a + kt * 4 * T2PLANE);
      if constexpr (G::K16 != 0) b16[u] = lds64(sm, a + in16);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (s == 10 && u == 0) DS_STAMP(x.slot, 44, 0);
      f32x4 v[NTT];
#pragma unroll
      for (int j = 0; j < NTT; ++j) v[j] = bias[j];
#pragma unroll
      for (int kt = 0; kt < G::KT32; ++kt)
#pragma unroll
        for (int j = 0; j < NTT; ++j) v[j] = mfma32(w32[kt][j], b32[u][kt], v[j]);
      if (s == 10 && u == 0) DS_STAMP(x.slot, 45, 0);
      if constexpr (G::K16 != 0) {
        shape_fence();
#pragma unroll
        for (int j = 0; j < NTT; ++j) v[j] = mfma16(w16[j], b16[u], v[j]);
      }
      const int row = r0 + ((U0 + u) >> 1);
      const int gy = x.py0 - 2 + row;
      const bool ok = u_act[u] && row >= 2 && row < x.LP - 2 && gy < p.H;
      const int al = ok ? gy * p.W + u_gx[u] : 0;  // level-local anchor
      if constexpr (KIND == 1) {
        upa_detect_box_store(p.de, v, x.n, al, ok, g);
        if (s == 10 && u == 0) DS_STAMP(x.slot, 47, 0);
      } else {
        float best = -1.f;
        int bc = 0;
        if (p.de.keys_only) {  // uniform
          cls_keys_lean<NTT>(p.de, v, ok, g, best, bc);
          if (s == 10 && u == 0) DS_STAMP(x.slot, 46, 0);
        } else if (p.de.nc == 16 * NTT) {  // uniform: every filter is a class (nc = 80) - the score rows in ONE predicated region, no per-class test
          constexpr float LOG2E = 1.44269504088896340736f;
          float sg[NTT][4];
#pragma unroll
          for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[j][q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[j][q] * -LOG2E));
          if (ok) {
            float* yb = p.de.y + ((size_t)x.n * (4 + p.de.nc) + 4 + 4 * g) * p.de.a_total + p.de.a0 + al;
#pragma unroll
            for (int j = 0; j < NTT; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) yb[(size_t)(16 * j + q) * p.de.a_total] = sg[j][q];
          }
          best = ok ? sg[0][0] : -1.f;  // the lane's FIRST maximum (ascending class order: strict >), as upa_detect_cls_store keeps it
          bc = 4 * g;
#pragma unroll
          for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (j == 0 && q == 0) continue;
              const bool take = ok && sg[j][q] > best;
              best = take ? sg[j][q] : best;
              bc = take ? 16 * j + 4 * g + q : bc;
            }
        } else {
#pragma unroll
          for (int j = 0; j < NTT; ++j) upa_detect_cls_store(p.de, v[j], j, x.n, al, ok, g, best, bc);
        }
        if (p.de.best_keys) upa_detect_best_key_store(p.de, best, bc, x.n, al, ok, x.lane);  // uniform
        if (s == 10 && u == 0) DS_STAMP(x.slot, 47, 0);
      }
    }
  }
};

// ---- the input band two steps ahead: band b = x rows {2b, 2b + 1} (image rows py0 - 2 + ...), 8 planes x 2 rows = 16 LDS-DMA instructions, lane =
// column slot (34 of 64 used); out-of-image slots read the zero page (the 3x3's zero padding).  The wave that carries this role also runs a tail
// unit, so it cannot count its loads apart from its stores: it issues the band at the START of a step and waits for everything at the end - the
// band then has the whole step (>= 3 k cycles) to land, and a further step before anybody reads it.
struct Dma {
  const char* ximg;
  unsigned rowpitch, coloff;
  bool colok;
  __device__ __forceinline__ void init(const Ctx& x) {
    const DsParams& p = *x.p;
    rowpitch = (unsigned)p.W * (unsigned)p.ldx * 2u;
    ximg = p.x + (size_t)x.n * p.H * rowpitch;
    const int gx = x.sx0 - 2 + x.lane;
    colok = x.lane < XW && gx >= 0 && gx < p.W;
    coloff = colok ? (unsigned)gx * (unsigned)p.ldx * 2u : 0u;
  }
  __device__ __forceinline__ void band(const Ctx& x, int b) {
    if (2 * b >= x.LP) return;  // wave-uniform
    const char* zp = reinterpret_cast<const char*>(g_ds_zero_page);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int gy = x.py0 - 2 + 2 * b + rr;
      const bool ok = colok && gy >= 0 && gy < x.p->H;
      const char* src = ok ? ximg + ((unsigned)gy * rowpitch + coloff) : zp;
      const int dst = XB + ((2 * b + rr) & (XROWS - 1)) * XROWB;  // + lane * 16 by the hardware
#pragma unroll
      for (int cg = 0; cg < 8; ++cg)
        __builtin_amdgcn_global_load_lds((ds_gptr_t)(src + (ok ? cg * 16 : 0)), (ds_lptr_t)(x.sm + dst + cg * XPLANE), 16, 0, 0);
    }
  }
};

template <typename R>
__device__ __forceinline__ void run_role(const Ctx& x, R& role) {
  __syncthreads();  // (the DMA wave arrives here with the first two bands landed)
  for (int s = 0; s < x.S; ++s) {
    DS_STAMP(x.slot, s, 0);
    role.step(x, s);
    DS_STAMP(x.slot, s, 1);
    __syncthreads();
  }
}
template <typename RA, typename RB>
__device__ __forceinline__ void run_roles2(const Ctx& x, RA& ra, RB& rb) {
  __syncthreads();
  for (int s = 0; s < x.S; ++s) {
    DS_STAMP(x.slot, s, 0);
    ra.step(x, s);
    rb.step(x, s);
    DS_STAMP(x.slot, s, 1);
    __syncthreads();
  }
}
template <typename T>
__device__ __forceinline__ void run_tail_dma(const Ctx& x, T& tail, Dma& dma) {
  dma.band(x, 0);
  dma.band(x, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < x.S; ++s) {
    DS_STAMP(x.slot, s, 0);
    dma.band(x, s + 2);
    tail.step(x, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DS_STAMP(x.slot, s, 1);
    __syncthreads();
  }
}
}  // namespace dstream

__global__ __launch_bounds__(512) void detect_stream_kernel(const DsParams p) {
  using namespace dstream;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = (int)blockIdx.x;
  const bool cls = bid < p.ncls_wg;
  if (!cls) bid -= p.ncls_wg;
  Ctx x;
  x.p = &p; x.sm = sm; x.lane = tid & 63; x.slot = DS_SLOT(p);
  x.n = bid / (p.parts * p.strips);
  bid -= x.n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  x.py0 = part * p.L; x.sx0 = strip * WS;
  int leff = p.H - x.py0 < p.L ? p.H - x.py0 : p.L;
  leff = (leff + 1) & ~1;
  x.LP = leff + 4;          // x rows of this part
  x.S = leff / 2 + 4;       // steps until the last output row has left

  // waves w and w + 4 share a SIMD.  Issue cycles per step (MFMA x 8 + vector x ~5 + LDS), estimated:
  //   box:    SIMD 0  A32(0) 1.9 k + tail unit 0 1.4 k    SIMD 1  A32(1) + tail unit 1    SIMD 2  B32(0) + tail unit 2    SIMD 3  B32(1) + tail unit 3 + DMA
  //   class:  SIMD 0  B32(0) 2.2 k + A16 1.4 k            SIMD 1  B32(1) + tail units 0-1 1.8 k    SIMD 2  A32(0) 1.9 k + B16 1.9 k    SIMD 3  A32(1) + tail units 2-3 + DMA
  if (cls) {
    const DsBranch& br = p.br[1];
    switch (wave) {
      case 0: { Conv32<80, 1, false> r; r.init(x, br, 0); run_role(x, r); break; }  // (80 input channels: 180 weight registers leave no room for a second accumulator set)
      case 1: { Conv32<80, 1, false> r; r.init(x, br, 1); run_role(x, r); break; }
      case 2: { Conv32<80, 0, true> r; r.init(x, br, 0); run_role(x, r); break; }
      case 3: { Conv32<80, 0, true> r; r.init(x, br, 1); run_role(x, r); break; }
      case 4: { Conv16<80, 0> r; r.init(x, br, 4); run_role(x, r); break; }
      case 5: { Tail<80, 2, 0, 2> r; r.init(x, br); run_role(x, r); break; }
      case 6: { Conv16<80, 1> r; r.init(x, br, 4); run_role(x, r); break; }
      default: { Tail<80, 2, 2, 2> r; Dma d; r.init(x, br); d.init(x); run_tail_dma(x, r, d); break; }
    }
  } else {
    const DsBranch& br = p.br[0];
    switch (wave) {
      case 0: { Conv32<64, 0, true> r; r.init(x, br, 0); run_role(x, r); break; }
      case 1: { Conv32<64, 0, true> r; r.init(x, br, 1); run_role(x, r); break; }
      case 2: { Conv32<64, 1, true> r; r.init(x, br, 0); run_role(x, r); break; }
      case 3: { Conv32<64, 1, true> r; r.init(x, br, 1); run_role(x, r); break; }
      case 4: { Tail<64, 1, 0, 1> r; r.init(x, br); run_role(x, r); break; }
      case 5: { Tail<64, 1, 1, 1> r; r.init(x, br); run_role(x, r); break; }
      case 6: { Tail<64, 1, 2, 1> r; r.init(x, br); run_role(x, r); break; }
      default: { Tail<64, 1, 3, 1> r; Dma d; r.init(x, br); d.init(x); run_tail_dma(x, r, d); break; }
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
