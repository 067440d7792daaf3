// C2f(64 -> 64, n = 2 Bottlenecks of 32 channels, bf16) as ONE kernel in its LINE-BUFFER form: model.4 of yolov8n at 80 x 80.
//   C2f.forward        ultralytics/nn/modules/block.py:457-488   y = list(cv1(x).chunk(2, 1)); y.extend(m(y[-1]) ...); cv2(cat(y, 1))
//   Bottleneck.forward ultralytics/nn/modules/block.py:644-668   x + cv2(cv1(x))
// The tile form of this block (c2f32_fused_kernel, c2f_fused.hip) recomputes a 4-pixel halo ring per 16 x 16 tile (cv1 2.25x, the 3x3
// convs 1.9 .. 1.3x), reloads the 3x3 weights of every stage through L1 in every tile, waits for its whole 73 KB input tile before
// the first MFMA, runs 3.125 one-per-CU rounds as four, and issues ~13 VALU instructions per MFMA (PMC, profiles/r04_pmc_step_budget.txt).
// Here a workgroup owns a vertical STRIP of the image (WS = 20 output columns, L output rows) and streams down it RS = 2 rows at
// a time; the five intermediates (y1, t1, b1, t2, b2) only ever exist as a few rows each in LDS ring buffers, and every wave has
// a fixed ROLE for the life of the workgroup, so weights are loaded into registers exactly once:
//   - 13 "3x3" waves: stage k (k = 0..3: t1, b1, t2, b2) x unit u, a unit = 16 consecutive pixels of the stage's RS-row band and ALL 32
//     output channels (both n-tiles): 9 ds_read_b128 feed 18 v_mfma_f32_16x16x32_bf16 - half the LDS reads and address arithmetic per
//     MFMA of the tile form; LDS tiles are PLANAR ([8-channel group][row][column][16 B]), so a tap is an immediate offset;
//   - 3 "X" waves: cv1 on the band the LDS-DMA of the previous step brought in (each wave stages exactly the pixels it consumes: no
//     barrier on the input path), cv2 on the band that left the last Bottleneck, y0 recomputed there from a second (L2-hot) read of
//     x so it never occupies LDS; output rows leave as 16-byte NHWC stores.
// One s_barrier per step; stage k runs (k + 1)(RS + 1) rows behind cv1, cv2 another RS behind.  Halo recompute: 1.4x on cv1 in x,
// nothing in y except the 8 + 14 rows of pipeline fill per strip.  The grid is sized to ONE round (32 images x 4 strips x 2 parts =
// 256 workgroups for the 80 x 80 maps at batch 32; the host picks L).
// Rounding points (bf16 y0, y1, t1, b1, t2, b2, out; f32 accumulation from the bias, taps in order, f32 residual add) are those of
// the separate launches and of the tile form.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((address_space(1))) const void* cgptr_t;
typedef __attribute__((address_space(3))) void* clptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct C2fsParams {
  const char* x; char* y;
  const char *w1, *w2;
  const char* wm[4];     // m[0].cv1, m[0].cv2, m[1].cv1, m[1].cv2
  const float *b1, *b2;
  const float* bm[4];
  int N, H, W, ldx, ldy, strips, parts, L, shortcut, xcd;
  // n = 1 form: cv1 reads c1 = 64 * NCH input channels; the first upC of them come from pixel (y / 2, x / 2) of the half-resolution tensor
  // `up` (a virtual nn.Upsample + Concat), the rest from x itself (whose pixel record still starts at channel 0 of the concat)
  const char* up; int c1, upC, up_ld;
};

// profiling build (-DUPA_STAMP): every wave of the first 8 workgroups records s_memtime at the start of each step and before its barrier
#ifdef UPA_STAMP
#define C2FS_STAMP_STEPS 48
__device__ unsigned long long g_c2fs_stamps[8 * 16 * C2FS_STAMP_STEPS * 2];
extern "C" int upa_debug_stamps_c2fs(unsigned long long* out, int count) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c2fs_stamps), (size_t)count * 8) == hipSuccess ? 0 : -1;
}
#define C2FS_STAMP(step, which)                                                                              \
  do {                                                                                                       \
    if (blockIdx.x < 8 && (step) < C2FS_STAMP_STEPS) {                                                       \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      if ((threadIdx.x & 63) == 0) g_c2fs_stamps[((blockIdx.x * 16 + (threadIdx.x >> 6)) * C2FS_STAMP_STEPS + (step)) * 2 + (which)] = t_; \
    }                                                                                                        \
  } while (0)
#else
#define C2FS_STAMP(step, which) do {} while (0)
#endif

namespace c2fs {
constexpr int WS = 20;  // output columns of a strip
constexpr int npow2(int v) { int q = 1; while (q < v) q <<= 1; return q; }

template <int NB>
struct Geo {
  static constexpr int R = 2 * NB;          // halo columns / rows each side
  static constexpr int XW = WS + 2 * R;     // columns of y1 (and the pitch of every ring)
  static constexpr int RS = 2;              // rows per step
  static constexpr int NST = 2 * NB;        // 3x3 stages
  static constexpr int NTEN = NST + 1;      // ring tensors y1, t1, b1 (, t2, b2)
  static constexpr int ROWB = XW * 16;      // bytes of one row of one 8-channel plane
  static constexpr int lag(int k) { return (k + 1) * (RS + 1); }  // stage k writes rows [RS s - lag, + RS) at step s
  static constexpr int LAGF = lag(NST - 1) + RS;                   // cv2
  static constexpr int sd(int k) { return XW - 2 * (k + 1); }      // valid columns of stage k's output
  static constexpr int units(int k) { return (RS * sd(k) + 15) / 16; }
  // rows of tensor j alive at once: written at lag(j - 1) (y1: 0), read down to LAGF (even j: cv2) or lag(j) + 1 (odd j)
  static constexpr int ring(int j) { return npow2(((j & 1) ? lag(j) + 1 : LAGF) - (j ? lag(j - 1) : 0) + RS); }
  static constexpr int plane(int j) { return ring(j) * ROWB; }
  static constexpr int base(int j) { int o = 0; for (int i = 0; i < j; ++i) o += 4 * plane(i); return o; }
  static constexpr int NY1 = (RS * XW + 15) / 16, NF = (RS * WS + 15) / 16;  // cv1 / cv2 units per step
  static constexpr int PY1 = (RS * XW + 7) / 8, PF = (RS * WS + 7) / 8;      // 8-pixel DMA pieces of the two input bands
  static constexpr int NSLOT = 3;                        // input bands in flight: the DMA runs two steps ahead of its readers
  static constexpr int XS = base(NTEN);                  // cv1 input staging: NSLOT slots of RS x XW pixels x 128 B
  static constexpr int XSLOT = NY1 * 16 * 128;
  static constexpr int XF = XS + NSLOT * XSLOT;          // y0 input staging: NSLOT slots of RS x WS pixels
  static constexpr int FSLOT = NF * 16 * 128;
  static constexpr int BIAS = XF + NSLOT * FSLOT;        // f32: bm[NST][32]
  static constexpr int LDS = BIAS + NST * 32 * 4;
};

#if defined(C2FS_EXP) && (C2FS_EXP == 11 || C2FS_EXP == 15)   // timing experiments only (tools/experiments/r05_c2fs_variants.sh): not SiLU
__device__ __forceinline__ float silu(float v) { return v * 0.5f; }
#elif defined(C2FS_EXP) && C2FS_EXP == 12
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + v * v); }
#else
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
#endif
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
#if defined(C2FS_EXP) && (C2FS_EXP == 13 || C2FS_EXP == 15)  // timing experiment: no matrix instructions (operands still loaded)
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(&a), *reinterpret_cast<const s16x4*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 silu_pack(const f32x4& a) {
  return u32x2{pack_bf16x2(silu(a[0]), silu(a[1])), pack_bf16x2(silu(a[2]), silu(a[3]))};
}
#if defined(C2FS_EXP) && C2FS_EXP == 14  // timing experiment: no LDS reads of the 3x3 taps / cv2 operands
__device__ __forceinline__ u32x4 lds128(const char* sm, int off) { return u32x4{(unsigned)off, 1u, 2u, 3u}; }
#else
__device__ __forceinline__ u32x4 lds128(const char* sm, int off) { return *reinterpret_cast<const u32x4*>(sm + off); }
#endif

// ---- a 3x3 stage K: the wave owns unit `unit` (and unit + 1 if TWO) of the stage's RS-row band for the life of the workgroup.
// A unit = 16 consecutive pixels of the band (row-major over the stage's SD valid columns) x all 32 output channels.
template <int NB, int K, bool TWO>
__device__ __forceinline__ void stage_role(const C2fsParams& p, char* sm, int unit, int lane, int S, int py0, int sx0, int LP, int bias_base) {
  using G = Geo<NB>;
  constexpr int SD = G::sd(K), C0 = K + 1, LAG = G::lag(K);
  constexpr int RIN = G::ring(K), ROUT = G::ring(K + 1);
  constexpr bool HAS_RES = (K & 1) != 0;
  constexpr int RRES = HAS_RES ? G::ring(K - 1) : 1;
  constexpr int PRES = HAS_RES ? G::plane(K - 1) : 0;
  constexpr int NU = TWO ? 2 : 1;
  constexpr int NBUF = 5;  // B fragments in flight
  static_assert(G::RS == 2, "the lane -> row map below assumes two rows per step");
  const int g = lane >> 4, r = lane & 15;

  u32x4 w[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w[tap][nt] = *reinterpret_cast<const u32x4*>(p.wm[K] + ((size_t)(tap * 2 + nt) * 64 + lane) * 16);

  // lane constants per unit: row within the band, byte offsets of the lane's pixel in the input / output / shortcut rings, validity
  int u_rr[NU], u_in[NU];
  unsigned u_colm[NU];
  bool u_act[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = (unit + u) * 16 + r;
    u_act[u] = q < G::RS * SD;
    const int qq = u_act[u] ? q : 0;
    u_rr[u] = qq >= SD ? 1 : 0;
    const int cc = C0 + qq - u_rr[u] * SD;
    u_in[u] = G::base(K) + g * G::plane(K) + (cc - 1) * 16;
    const int gx = sx0 - G::R + cc;
    u_colm[u] = (gx >= 0 && gx < p.W) ? 0xFFFFFFFFu : 0u;
  }
  // the lane's 8 bytes in the output / shortcut rings sit at a lane-constant distance from its input address (same pixel, another ring)
  const int out_d = G::base(K + 1) - G::base(K) + (g >> 1) * G::plane(K + 1) - g * G::plane(K) + 16 + (g & 1) * 8;
  const int res_d = HAS_RES ? G::base(K - 1) - G::base(K) + (g >> 1) * PRES - g * G::plane(K) + 16 + (g & 1) * 8 : 0;
  const int bias_off = bias_base + (K * 32 + 4 * g) * 4;
  const int lo = K + 1, hi = LP - (K + 1);
  const bool use_res = HAS_RES && p.shortcut;

  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    const int r0 = G::RS * s - LAG;
    if (r0 + G::RS > lo && r0 < hi) {  // wave-uniform
      int rb[NU][3];
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb[u][dy] = u_in[u] + ((r0 + u_rr[u] + dy - 1) & (RIN - 1)) * G::ROWB;
      auto rd = [&](int t) __attribute__((always_inline)) { return lds128(sm, rb[t / 9][(t % 9) / 3] + (t % 3) * 16); };
      constexpr int NTAP = 9 * NU;
      u32x4 buf[NBUF];
#pragma unroll
      for (int t = 0; t < NBUF; ++t) buf[t] = rd(t);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int row = r0 + u_rr[u];
        u32x2 rs0 = {0u, 0u}, rs1 = {0u, 0u};
        f32x4 acc0 = *reinterpret_cast<const f32x4*>(sm + bias_off);
        f32x4 acc1 = *reinterpret_cast<const f32x4*>(sm + bias_off + 64);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int t = u * 9 + tap;
          const u32x4 b = buf[t % NBUF];
          acc0 = mfma32(w[tap][0], b, acc0);
          acc1 = mfma32(w[tap][1], b, acc1);
          if (t + NBUF < NTAP) buf[t % NBUF] = rd(t + NBUF);
          if (tap == 4 && use_res) {  // the shortcut operand arrives under the remaining taps
            const int ra = u_in[u] + res_d + (row & (RRES - 1)) * G::ROWB;
            rs0 = *reinterpret_cast<const u32x2*>(sm + ra);
            rs1 = *reinterpret_cast<const u32x2*>(sm + ra + 2 * PRES);
          }
        }
        const int gy = py0 - G::R + row;
        const unsigned m = (gy >= 0 && gy < p.H) ? u_colm[u] : 0u;  // the tensor is ZERO outside the image (the next conv's padding)
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = silu(acc0[e]); v1[e] = silu(acc1[e]); }
        if (use_res) {
          v0[0] += __uint_as_float(rs0[0] << 16); v0[1] += __uint_as_float(rs0[0] & 0xFFFF0000u);
          v0[2] += __uint_as_float(rs0[1] << 16); v0[3] += __uint_as_float(rs0[1] & 0xFFFF0000u);
          v1[0] += __uint_as_float(rs1[0] << 16); v1[1] += __uint_as_float(rs1[0] & 0xFFFF0000u);
          v1[2] += __uint_as_float(rs1[1] << 16); v1[3] += __uint_as_float(rs1[1] & 0xFFFF0000u);
        }
        const u32x2 o0 = u32x2{pack_bf16x2(v0[0], v0[1]) & m, pack_bf16x2(v0[2], v0[3]) & m};
        const u32x2 o1 = u32x2{pack_bf16x2(v1[0], v1[1]) & m, pack_bf16x2(v1[2], v1[3]) & m};
        if (u_act[u] && row >= lo && row < hi) {
          const int oa = u_in[u] + out_d + (row & (ROUT - 1)) * G::ROWB;
          *reinterpret_cast<u32x2*>(sm + oa) = o0;
          *reinterpret_cast<u32x2*>(sm + oa + 2 * G::plane(K + 1)) = o1;
        }
      }
    }
    C2FS_STAMP(s, 1);
    __syncthreads();
  }
}

// ---- cv1's upper half (y1 into its ring) on two units of the band, and the LDS-DMA of the NEXT step's input rows for everybody:
// Y wave yi of 2.  Pieces of 8 pixels x 128 B; this wave's share lands before its barrier (s_waitcnt vmcnt(0): a Y wave has no stores
// in flight), the readers run after it.
template <int NB>
__device__ __forceinline__ void y_role(const C2fsParams& p, char* sm, int yi, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  static_assert(G::NY1 <= 4 && G::RS == 2, "two Y waves x two units; lane -> row maps assume two rows per step");
  constexpr int PCS = (G::PY1 + 1) / 2 + (G::PF + 1) / 2;  // DMA pieces per Y wave
  static_assert(PCS <= 7, "the vmcnt switch below covers up to seven pieces in flight");
  const int g = lane >> 4, r = lane & 15;
  const unsigned rowpitch = (unsigned)p.W * (unsigned)p.ldx * 2u;
  const char* ximg = p.x + (size_t)n * p.H * rowpitch;

  u32x4 w1f[2][2];  // [k-tile][n-tile 2, 3]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w1f[kt][nt] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(kt * 4 + 2 + nt) * 64 + lane) * 16);
  const f32x4 bias0 = *reinterpret_cast<const f32x4*>(p.b1 + 32 + 4 * g), bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 48 + 4 * g);

  int u_q[2], u_rr[2], u_out[2];
  unsigned u_colm[2];
  bool u_act[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = (2 * yi + u) * 16 + r;
    u_q[u] = q;
    u_act[u] = (2 * yi + u) < G::NY1 && q < G::RS * G::XW;
    const int qq = u_act[u] ? q : 0;
    u_rr[u] = qq >= G::XW ? 1 : 0;
    const int col = qq - u_rr[u] * G::XW;
    u_out[u] = G::base(0) + (g >> 1) * G::plane(0) + col * 16 + (g & 1) * 8;
    const int gx = sx0 - G::R + col;
    u_colm[u] = (gx >= 0 && gx < p.W) ? 0xFFFFFFFFu : 0u;
  }
  // DMA pieces: cv1 band pieces [yi * ceil(PY1 / 2), ...), cv2 band pieces likewise; lane (lane >> 3) = pixel of the piece, (lane & 7) = 16-byte slot
  unsigned d_col[PCS];
  int d_rr[PCS], d_dst[PCS];
  bool d_ok[PCS], d_isf[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const bool isf = i >= (G::PY1 + 1) / 2;
    const int pc = isf ? yi * ((G::PF + 1) / 2) + (i - (G::PY1 + 1) / 2) : yi * ((G::PY1 + 1) / 2) + i;
    const int npc = isf ? G::PF : G::PY1, width = isf ? WS : G::XW, npx = G::RS * width;
    d_isf[i] = isf;
    d_ok[i] = pc < npc && (isf ? i - (G::PY1 + 1) / 2 < (G::PF + 1) / 2 : true);
    const int pd = pc * 8 + (lane >> 3);
    const int pq = pd < npx ? pd : 0;
    d_rr[i] = pq >= width ? 1 : 0;
    const int dcol = pq - d_rr[i] * width;
    int dgx = sx0 + dcol - (isf ? 0 : G::R);
    dgx = dgx < 0 ? 0 : (dgx >= p.W ? p.W - 1 : dgx);
    const int cg = (lane & 7) ^ (pd & 7);
    d_col[i] = (unsigned)dgx * (unsigned)p.ldx * 2u + (unsigned)cg * 16u;
    d_dst[i] = (isf ? G::XF : G::XS) + pc * 1024;
  }
  auto stage_in = [&](int st) __attribute__((always_inline)) {  // returns the number of pieces issued
    const int slot = st % G::NSLOT;
    int issued = 0;
    const int y0r = py0 - G::R + G::RS * st;             // image row of the cv1 band's first row
    const int f0r = y0r - G::LAGF;                       // ... of the cv2 band's
    const bool yon = G::RS * st < LP;
    const int o0 = G::RS * st - G::LAGF;
    const bool fon = o0 + G::RS > G::R && o0 < LP - G::R;
    unsigned yro[2], fro[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int a = y0r + k, b = f0r + k;
      a = a < 0 ? 0 : (a >= p.H ? p.H - 1 : a);
      b = b < 0 ? 0 : (b >= p.H ? p.H - 1 : b);
      yro[k] = (unsigned)a * rowpitch;
      fro[k] = (unsigned)b * rowpitch;
    }
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      if (!d_ok[i] || !(d_isf[i] ? fon : yon)) continue;  // wave-uniform
      const unsigned off = d_col[i] + (d_isf[i] ? (d_rr[i] ? fro[1] : fro[0]) : (d_rr[i] ? yro[1] : yro[0]));
      __builtin_amdgcn_global_load_lds((cgptr_t)(ximg + off), (clptr_t)(sm + d_dst[i] + slot * (d_isf[i] ? G::FSLOT : G::XSLOT)), 16, 0, 0);
      ++issued;
    }
    return issued;
  };

  stage_in(0);
  stage_in(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    const int inflight = stage_in(s + 2);  // lands during the NEXT step; the band of step s + 1 went out a step ago
    if (s == 12) C2FS_STAMP(32, 0);
    if (G::RS * s < LP) {
      const char* xb = sm + G::XS + (s % G::NSLOT) * G::XSLOT;
      u32x4 bx[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) bx[u][kt] = *reinterpret_cast<const u32x4*>(xb + u_q[u] * 128 + (((kt * 4 + g) ^ (u_q[u] & 7)) << 4));
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (2 * yi + u >= G::NY1) break;  // wave-uniform
        f32x4 a0 = bias0, a1 = bias1;
        if (s == 12 && u == 0) C2FS_STAMP(32, 1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          a0 = mfma32(w1f[kt][0], bx[u][kt], a0);
          a1 = mfma32(w1f[kt][1], bx[u][kt], a1);
        }
        const int row = G::RS * s + u_rr[u];
        const int gy = py0 - G::R + row;
        const unsigned m = (gy >= 0 && gy < p.H) ? u_colm[u] : 0u;
        u32x2 o0 = silu_pack(a0), o1 = silu_pack(a1);
        o0[0] &= m; o0[1] &= m; o1[0] &= m; o1[1] &= m;
        if (u_act[u] && row < LP) {
          const int oa = u_out[u] + (row & (G::ring(0) - 1)) * G::ROWB;
          *reinterpret_cast<u32x2*>(sm + oa) = o0;
          *reinterpret_cast<u32x2*>(sm + oa + 2 * G::plane(0)) = o1;
        }
        if (s == 12 && u == 0) C2FS_STAMP(33, 0);
      }
    }
    C2FS_STAMP(s, 1);
    // the NEXT step's input pieces of this wave have landed: everything but the `inflight` youngest requests (a Y wave issues no other
    // vector-memory operation, so the count is exact)
    switch (inflight) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
    __syncthreads();
  }
}

// ---- y0 + cv2 on unit fu of the output band, output n-tiles 2 fj, 2 fj + 1: every weight in registers, no LDS-DMA, stores never waited for
template <int NB>
__device__ __forceinline__ void f_role(const C2fsParams& p, char* sm, int fu, int fj, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  static_assert(G::RS == 2, "lane -> row maps assume two rows per step");
  const int g = lane >> 4, r = lane & 15;

  u32x4 w1f[2][2];  // cv1 [k-tile][n-tile 0, 1] (y0)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w1f[kt][nt] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(kt * 4 + nt) * 64 + lane) * 16);
  // cv2's k-tile 0 (y0) as two 16-wide k-steps against the D layout of the y0 accumulators: lane (g, r) = W2[co = 16 nt + r][ci = 16 ks + 4g .. + 3]
  u32x2 w2y0[2][2];
  u32x4 w2f[1 + NB][2];  // k-tiles 1.. (y1, b1 (, b2)) x this wave's two n-tiles
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c0 = 16 * ks + 4 * g;
      const int gg = (c0 & 31) >> 3, half = (c0 & 7) >> 2;
      w2y0[ks][nt] = *reinterpret_cast<const u32x2*>(p.w2 + ((size_t)((2 * fj + nt) * 64 + gg * 16 + r)) * 16 + half * 8);
    }
#pragma unroll
    for (int k = 0; k < 1 + NB; ++k) w2f[k][nt] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)((k + 1) * 4 + 2 * fj + nt) * 64 + lane) * 16);
  }
  const f32x4 by0a = *reinterpret_cast<const f32x4*>(p.b1 + 4 * g), by0b = *reinterpret_cast<const f32x4*>(p.b1 + 16 + 4 * g);
  const f32x4 b2a = *reinterpret_cast<const f32x4*>(p.b2 + (2 * fj) * 16 + 4 * g), b2b = *reinterpret_cast<const f32x4*>(p.b2 + (2 * fj + 1) * 16 + 4 * g);

  const int fq = fu * 16 + r;
  const bool f_act = fq < G::RS * WS;
  const int fqq = f_act ? fq : 0;
  const int f_rr = fqq >= WS ? 1 : 0, f_oc = fqq - f_rr * WS;
  const bool f_colok = sx0 + f_oc < p.W;
  int f_in[1 + NB];
#pragma unroll
  for (int k = 0; k < 1 + NB; ++k) f_in[k] = G::base(2 * k) + g * G::plane(2 * k) + (G::R + f_oc) * 16;
  const int cb = 16 * (2 * fj + (g & 1)) + 8 * (g >> 1);
  char* const ybase = p.y + (((size_t)n * p.H * p.W) + (size_t)(sx0 + f_oc)) * (size_t)p.ldy * 2 + cb * 2;
  const size_t yrow = (size_t)p.W * p.ldy * 2;

  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    const int o0r = G::RS * s - G::LAGF;
    if (o0r + G::RS > G::R && o0r < LP - G::R) {
      const int row = o0r + f_rr;
      const char* xb = sm + G::XF + (s % G::NSLOT) * G::FSLOT + fq * 128;
      u32x4 bx[2], opnd[1 + NB];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) bx[kt] = *reinterpret_cast<const u32x4*>(xb + (((kt * 4 + g) ^ (fq & 7)) << 4));
#pragma unroll
      for (int k = 0; k < 1 + NB; ++k) opnd[k] = lds128(sm, f_in[k] + (row & (G::ring(2 * k) - 1)) * G::ROWB);
      f32x4 y0a = by0a, y0b = by0b;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        y0a = mfma32(w1f[kt][0], bx[kt], y0a);
        y0b = mfma32(w1f[kt][1], bx[kt], y0b);
      }
      if (s == 12) C2FS_STAMP(32, 0);
      const u32x2 y0B[2] = {silu_pack(y0a), silu_pack(y0b)};
      if (s == 12) C2FS_STAMP(32, 1);
      f32x4 o0 = b2a, o1 = b2b;
      o0 = mfma16(w2y0[0][0], y0B[0], o0);
      o1 = mfma16(w2y0[0][1], y0B[0], o1);
      o0 = mfma16(w2y0[1][0], y0B[1], o0);
      o1 = mfma16(w2y0[1][1], y0B[1], o1);
      // A 4-pass 16x16x16 MFMA whose result is the NEXT instruction's srcC of an 8-pass 16x16x32 MFMA came out wrong in rows 2, 3
      // of each lane's four (measured, ROCm 7.2 / gfx950: hipcc inserts no wait states between the two shapes on one accumulator):
      // finish the 16-wide chains first, then wait out the short pipeline before the 32-wide chains start.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 1 + NB; ++k) {
        o0 = mfma32(w2f[k][0], opnd[k], o0);
        o1 = mfma32(w2f[k][1], opnd[k], o1);
      }
      const int gy = py0 - G::R + row;
      if (s == 12) C2FS_STAMP(33, 0);
      const u32x2 a = silu_pack(o0), b = silu_pack(o1);
      auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
      if (f_act && f_colok && row >= G::R && row < LP - G::R && gy < p.H)
        *reinterpret_cast<u32x4*>(ybase + (size_t)gy * yrow) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    }
    C2FS_STAMP(s, 1);
    __syncthreads();
  }
}

// =====================================================================================================================
// The n = 1 block: C2f(64 NCH -> 64, one Bottleneck of 32 channels) - yolov8n model.15 (NCH = 3: 128 upsampled + 64 skip channels, read
// through the virtual nn.Upsample + Concat) and yolov8s model.2 (NCH = 1).  Same strips, rings and 3x3 roles (halo 2: y1 24 columns,
// t1 22, b1 20); cv1 is 1.7x - 5x the work of the n = 2 block's, so it gets six waves (unit x half: y0 | y1, weights 16 NCH registers) and
// y0 goes through a ring of its own instead of being recomputed by cv2; the input band (NCH 64-channel chunks) is staged by three waves that
// do nothing else; cv2 is three waves with all 12 fragments in registers.
// =====================================================================================================================
template <int NB, int NCH, int ND>
struct GeoS : Geo<NB> {
  using G = Geo<NB>;
  static constexpr int Y0B = G::base(G::NTEN);            // y0 ring: as y1 (same rows alive, same pitch)
  static constexpr int XS1 = Y0B + 4 * G::plane(0);       // input staging: NSLOT slots x NCH chunks x (RS x XW pixels x 128 B)
  static constexpr int XCH = G::NY1 * 16 * 128;
  static constexpr int XSLOT1 = NCH * XCH;
  static constexpr int BIAS1 = XS1 + G::NSLOT * XSLOT1;   // f32: bm[NST][32]
  static constexpr int LDS1 = BIAS1 + G::NST * 32 * 4;
  static constexpr int PCS = NCH * G::PY1;                // DMA pieces per step
  static constexpr int PPW = (PCS + ND - 1) / ND;         // ... per DMA wave
};
template <int NCH> using Geo1 = GeoS<1, NCH, 3>;

// cv1 on unit u of the band, output half h (0: y0 = n-tiles 0, 1 -> the y0 ring; 1: y1 = n-tiles 2, 3 -> the y1 ring, ZERO outside the image)
// (NH = 2: both halves by one wave - the n = 2 block, whose cv1 is a quarter of the work)
template <int NB, int NCH, int ND, int NH>
__device__ __forceinline__ void cv1_role(const C2fsParams& p, char* sm, int u, int h, int lane, int S, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  using G1 = GeoS<NB, NCH, ND>;
  const int g = lane >> 4, r = lane & 15;
  u32x4 w1f[2 * NCH][2 * NH];
#pragma unroll
  for (int kt = 0; kt < 2 * NCH; ++kt)
#pragma unroll
    for (int nt = 0; nt < 2 * NH; ++nt) w1f[kt][nt] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(kt * 4 + 2 * h + nt) * 64 + lane) * 16);
  f32x4 bias[2 * NH];
#pragma unroll
  for (int nt = 0; nt < 2 * NH; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(p.b1 + (2 * h + nt) * 16 + 4 * g);
  const int q = u * 16 + r;
  const bool act = q < G::RS * G::XW;
  const int qq = act ? q : 0;
  const int rr = qq >= G::XW ? 1 : 0, col = qq - rr * G::XW;
  const int out_rel = (g >> 1) * G::plane(0) + col * 16 + (g & 1) * 8;  // + the ring's base: y0 (half 0) | y1 (half 1)
  const int gx = sx0 - G::R + col;
  const unsigned colm = (gx >= 0 && gx < p.W) ? 0xFFFFFFFFu : 0u;
  const int x_c = q * 128;
  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    if (G::RS * s < LP) {
      const char* xb = sm + G1::XS1 + (s % G::NSLOT) * G1::XSLOT1 + x_c;
      u32x4 bx[2 * NCH];
#pragma unroll
      for (int kt = 0; kt < 2 * NCH; ++kt) bx[kt] = *reinterpret_cast<const u32x4*>(xb + (kt >> 1) * G1::XCH + ((((kt & 1) * 4 + g) ^ (q & 7)) << 4));
      f32x4 a[2 * NH];
#pragma unroll
      for (int nt = 0; nt < 2 * NH; ++nt) a[nt] = bias[nt];
#pragma unroll
      for (int kt = 0; kt < 2 * NCH; ++kt)
#pragma unroll
        for (int nt = 0; nt < 2 * NH; ++nt) a[nt] = mfma32(w1f[kt][nt], bx[kt], a[nt]);
      const int row = G::RS * s + rr;
      const int gy = py0 - G::R + row;
      const unsigned m = (gy >= 0 && gy < p.H) ? colm : 0u;
      const int orow = out_rel + (row & (G::ring(0) - 1)) * G::ROWB;
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        u32x2 o0 = silu_pack(a[2 * hh]), o1 = silu_pack(a[2 * hh + 1]);
        o0[0] &= m; o0[1] &= m; o1[0] &= m; o1[1] &= m;
        if (act && row < LP) {
          const int oa = orow + ((h + hh) ? G::base(0) : G1::Y0B);
          *reinterpret_cast<u32x2*>(sm + oa) = o0;
          *reinterpret_cast<u32x2*>(sm + oa + 2 * G::plane(0)) = o1;
        }
      }
    }
    C2FS_STAMP(s, 1);
    __syncthreads();
  }
}

// cv2 over [y0 | y1 | b1] on unit fu of the output band: 12 fragments in registers, operands from the three rings
template <int NB, int NCH, int ND>
__device__ __forceinline__ void cv2_role(const C2fsParams& p, char* sm, int fu, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  using G1 = GeoS<NB, NCH, ND>;
  constexpr int K2 = 2 + NB;  // operands y0, y1, b1 (, b2)
  const int g = lane >> 4, r = lane & 15;
  u32x4 w2f[K2][4];
#pragma unroll
  for (int k = 0; k < K2; ++k)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w2f[k][nt] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)(k * 4 + nt) * 64 + lane) * 16);
  f32x4 b2v[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b2v[nt] = *reinterpret_cast<const f32x4*>(p.b2 + nt * 16 + 4 * g);
  const int fq = fu * 16 + r;
  const bool f_act = fq < G::RS * WS;
  const int fqq = f_act ? fq : 0;
  const int f_rr = fqq >= WS ? 1 : 0, f_oc = fqq - f_rr * WS;
  const bool f_colok = sx0 + f_oc < p.W;
  int f_in[K2];
  f_in[0] = G1::Y0B + g * G::plane(0) + (G::R + f_oc) * 16;
#pragma unroll
  for (int k = 1; k < K2; ++k) f_in[k] = G::base(2 * (k - 1)) + g * G::plane(2 * (k - 1)) + (G::R + f_oc) * 16;
  char* const ybase = p.y + (((size_t)n * p.H * p.W) + (size_t)(sx0 + f_oc)) * (size_t)p.ldy * 2 + (16 * (g & 1) + 8 * (g >> 1)) * 2;
  const size_t yrow = (size_t)p.W * p.ldy * 2;
  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    const int o0r = G::RS * s - G::LAGF;
    if (o0r + G::RS > G::R && o0r < LP - G::R) {
      const int row = o0r + f_rr;
      u32x4 opnd[K2];
      opnd[0] = lds128(sm, f_in[0] + (row & (G::ring(0) - 1)) * G::ROWB);
#pragma unroll
      for (int k = 1; k < K2; ++k) opnd[k] = lds128(sm, f_in[k] + (row & (G::ring(2 * (k - 1)) - 1)) * G::ROWB);
      f32x4 o[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) o[nt] = b2v[nt];
#pragma unroll
      for (int k = 0; k < K2; ++k)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = mfma32(w2f[k][nt], opnd[k], o[nt]);
      const int gy = py0 - G::R + row;
      const bool ok = f_act && f_colok && row >= G::R && row < LP - G::R && gy < p.H;
      char* dst = ybase + (size_t)gy * yrow;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x2 a = silu_pack(o[2 * j]), b = silu_pack(o[2 * j + 1]);
        auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
        if (ok) *reinterpret_cast<u32x4*>(dst + j * 64) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
    C2FS_STAMP(s, 1);
    __syncthreads();
  }
}

// the LDS-DMA of the input band two steps ahead: wave di of 3 owns pieces [di PPW, (di + 1) PPW) of the NCH x PY1 pieces of a band
template <int NB, int NCH, int ND>
__device__ __forceinline__ void dma_role(const C2fsParams& p, char* sm, int di, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  using G1 = GeoS<NB, NCH, ND>;
  static_assert(G1::PPW <= 7, "the vmcnt switch covers up to seven pieces in flight");
  const unsigned rowpitch = (unsigned)p.W * (unsigned)p.ldx * 2u;
  const unsigned up_rowpitch = (unsigned)(p.W >> 1) * (unsigned)p.up_ld * 2u;
  const char* ximg = p.x + (size_t)n * p.H * rowpitch;
  const char* uimg = p.up ? p.up + (size_t)n * (p.H >> 1) * up_rowpitch : nullptr;
  unsigned d_col[G1::PPW];
  int d_rr[G1::PPW], d_dst[G1::PPW];
  bool d_ok[G1::PPW], d_up[G1::PPW];
#pragma unroll
  for (int i = 0; i < G1::PPW; ++i) {
    const int pi = di * G1::PPW + i;
    d_ok[i] = pi < G1::PCS;
    const int c = d_ok[i] ? pi / G::PY1 : 0, pc = d_ok[i] ? pi % G::PY1 : 0;
    const int pd = pc * 8 + (lane >> 3);
    const int pq = pd < G::RS * G::XW ? pd : 0;
    d_rr[i] = pq >= G::XW ? 1 : 0;
    const int dcol = pq - d_rr[i] * G::XW;
    int dgx = sx0 + dcol - G::R;
    dgx = dgx < 0 ? 0 : (dgx >= p.W ? p.W - 1 : dgx);
    const int cg = (lane & 7) ^ (pd & 7);
    d_up[i] = p.up != nullptr && c * 64 < p.upC;
    d_col[i] = d_up[i] ? (unsigned)(dgx >> 1) * (unsigned)p.up_ld * 2u + (unsigned)c * 128u + (unsigned)cg * 16u
                       : (unsigned)dgx * (unsigned)p.ldx * 2u + (unsigned)c * 128u + (unsigned)cg * 16u;
    d_dst[i] = G1::XS1 + c * G1::XCH + pc * 1024;
  }
  auto stage_in = [&](int st) __attribute__((always_inline)) {
    int issued = 0;
    if (G::RS * st < LP) {
      const int slot = st % G::NSLOT;
      const int y0r = py0 - G::R + G::RS * st;
      unsigned xro[2], uro[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int a = y0r + k;
        a = a < 0 ? 0 : (a >= p.H ? p.H - 1 : a);
        xro[k] = (unsigned)a * rowpitch;
        uro[k] = (unsigned)(a >> 1) * up_rowpitch;
      }
#pragma unroll
      for (int i = 0; i < G1::PPW; ++i) {
        if (!d_ok[i]) continue;  // wave-uniform
        const char* src = d_up[i] ? uimg + (d_col[i] + (d_rr[i] ? uro[1] : uro[0])) : ximg + (d_col[i] + (d_rr[i] ? xro[1] : xro[0]));
        __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(sm + d_dst[i] + slot * G1::XSLOT1), 16, 0, 0);
        ++issued;
      }
    }
    return issued;
  };
  stage_in(0);
  stage_in(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    C2FS_STAMP(s, 0);
    const int inflight = stage_in(s + 2);
    C2FS_STAMP(s, 1);
    switch (inflight) {  // everything but this step's requests has landed (a DMA wave issues no other vector-memory operation)
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
    __syncthreads();
  }
}
}  // namespace c2fs

template <int NB>
__global__ __launch_bounds__(1024) void c2f32_stream_kernel(const C2fsParams p) {
  using G = c2fs::Geo<NB>;
  static_assert(NB == 2, "wave roles below are those of the n = 2 block");
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.N * p.parts * p.strips;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, total) : (int)blockIdx.x;
  const int n = bid / (p.parts * p.strips);
  bid -= n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  const int py0 = part * p.L, sx0 = strip * c2fs::WS;
  int leff = p.H - py0 < p.L ? p.H - py0 : p.L;
  leff = (leff + G::RS - 1) / G::RS * G::RS;
  const int LP = leff + 2 * G::R;                              // rows of y1 this workgroup produces
  const int S = (leff + G::R + G::LAGF - 1) / G::RS + 1;       // steps until the last output row has left

  // the 3x3 stages' biases -> LDS (their waves have no registers to spare)
  if (tid < G::NST * 32) reinterpret_cast<float*>(sm + G::BIAS)[tid] = p.bm[tid >> 5][tid & 31];
  if (wave != 11 && wave != 15) __syncthreads();  // (the Y waves arrive at this barrier with the first input band landed)

  // wave -> role.  Waves w, w + 4, w + 8, w + 12 share a SIMD; MFMAs per step and SIMD: 82 / 82 / 82 / 88.
  //   SIMD 0: t1 units 0-1, b1 unit 2, cv2 (0, 0), cv2 (1, 1)     SIMD 1: t1 units 2-3, t2 unit 2, cv2 (0, 1), cv2 (2, 0)
  //   SIMD 2: b1 units 0-1, b2 unit 2, cv2 (1, 0), cv2 (2, 1)     SIMD 3: t2 units 0-1, b2 units 0-1, the two cv1 + DMA waves
  switch (wave) {
    case 0: c2fs::stage_role<NB, 0, true>(p, sm, 0, lane, S, py0, sx0, LP, G::BIAS); break;
    case 1: c2fs::stage_role<NB, 0, true>(p, sm, 2, lane, S, py0, sx0, LP, G::BIAS); break;
    case 2: c2fs::stage_role<NB, 1, true>(p, sm, 0, lane, S, py0, sx0, LP, G::BIAS); break;
    case 3: c2fs::stage_role<NB, 2, true>(p, sm, 0, lane, S, py0, sx0, LP, G::BIAS); break;
    case 4: c2fs::stage_role<NB, 1, false>(p, sm, 2, lane, S, py0, sx0, LP, G::BIAS); break;
    case 5: c2fs::stage_role<NB, 2, false>(p, sm, 2, lane, S, py0, sx0, LP, G::BIAS); break;
    case 6: c2fs::stage_role<NB, 3, false>(p, sm, 2, lane, S, py0, sx0, LP, G::BIAS); break;
    case 7: c2fs::stage_role<NB, 3, true>(p, sm, 0, lane, S, py0, sx0, LP, G::BIAS); break;
    case 8: c2fs::f_role<NB>(p, sm, 0, 0, lane, S, n, py0, sx0, LP); break;
    case 9: c2fs::f_role<NB>(p, sm, 0, 1, lane, S, n, py0, sx0, LP); break;
    case 10: c2fs::f_role<NB>(p, sm, 1, 0, lane, S, n, py0, sx0, LP); break;
    case 11: c2fs::y_role<NB>(p, sm, 0, lane, S, n, py0, sx0, LP); break;
    case 12: c2fs::f_role<NB>(p, sm, 1, 1, lane, S, n, py0, sx0, LP); break;
    case 13: c2fs::f_role<NB>(p, sm, 2, 0, lane, S, n, py0, sx0, LP); break;
    case 14: c2fs::f_role<NB>(p, sm, 2, 1, lane, S, n, py0, sx0, LP); break;
    default: c2fs::y_role<NB>(p, sm, 1, lane, S, n, py0, sx0, LP); break;
  }
}

template <int NCH>
__global__ __launch_bounds__(1024) void c2f32_stream1_kernel(const C2fsParams p) {
  using G = c2fs::Geo<1>;
  using G1 = c2fs::Geo1<NCH>;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.N * p.parts * p.strips;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, total) : (int)blockIdx.x;
  const int n = bid / (p.parts * p.strips);
  bid -= n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  const int py0 = part * p.L, sx0 = strip * c2fs::WS;
  int leff = p.H - py0 < p.L ? p.H - py0 : p.L;
  leff = (leff + G::RS - 1) / G::RS * G::RS;
  const int LP = leff + 2 * G::R;
  const int S = (leff + G::R + G::LAGF - 1) / G::RS + 1;
  if (tid < 64) reinterpret_cast<float*>(sm + G1::BIAS1)[tid] = p.bm[tid >> 5][tid & 31];
  if (wave != 12 && wave != 13 && wave != 15) __syncthreads();  // (the DMA waves arrive at this barrier with the first two bands landed)
  // wave -> role.  Waves w, w + 4, w + 8, w + 12 share a SIMD.
  //   SIMD 0: t1 units 0-1, cv1 (0, y0), cv1 (0, y1), DMA 0      SIMD 1: b1 units 0-1, cv1 (1, y0), cv1 (1, y1), DMA 1
  //   SIMD 2: t1 unit 2, b1 unit 2, cv1 (2, y0), cv2 unit 0      SIMD 3: cv1 (2, y1), cv2 units 1, 2, DMA 2
  switch (wave) {
    case 0: c2fs::stage_role<1, 0, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 1: c2fs::stage_role<1, 1, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 2: c2fs::stage_role<1, 0, false>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 6: c2fs::stage_role<1, 1, false>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 4: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 0, 0, lane, S, py0, sx0, LP); break;
    case 8: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 0, 1, lane, S, py0, sx0, LP); break;
    case 5: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 1, 0, lane, S, py0, sx0, LP); break;
    case 9: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 1, 1, lane, S, py0, sx0, LP); break;
    case 10: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 2, 0, lane, S, py0, sx0, LP); break;
    case 3: c2fs::cv1_role<1, NCH, 3, 1>(p, sm, 2, 1, lane, S, py0, sx0, LP); break;
    case 14: c2fs::cv2_role<1, NCH, 3>(p, sm, 0, lane, S, n, py0, sx0, LP); break;
    case 7: c2fs::cv2_role<1, NCH, 3>(p, sm, 1, lane, S, n, py0, sx0, LP); break;
    case 11: c2fs::cv2_role<1, NCH, 3>(p, sm, 2, lane, S, n, py0, sx0, LP); break;
    case 12: c2fs::dma_role<1, NCH, 3>(p, sm, 0, lane, S, n, py0, sx0, LP); break;
    case 13: c2fs::dma_role<1, NCH, 3>(p, sm, 1, lane, S, n, py0, sx0, LP); break;
    default: c2fs::dma_role<1, NCH, 3>(p, sm, 2, lane, S, n, py0, sx0, LP); break;
  }
}

// The n = 2 block on the role set of the n = 1 kernel: cv1 (both halves, one unit per wave) writes y0 into a ring of its own instead of
// each cv2 wave recomputing it from a second read of x; cv2 = three waves with all 16 fragments in registers; ONE wave stages the
// input band.  Per step 216 SiLU wave-values and 318 MFMAs instead of 232 and 334, no cv1 wave issues LDS-DMA, 7 pieces instead of 12.
__global__ __launch_bounds__(1024) void c2f32_stream2_kernel(const C2fsParams p) {
  using G = c2fs::Geo<2>;
  using G1 = c2fs::GeoS<2, 1, 1>;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.N * p.parts * p.strips;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, total) : (int)blockIdx.x;
  const int n = bid / (p.parts * p.strips);
  bid -= n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  const int py0 = part * p.L, sx0 = strip * c2fs::WS;
  int leff = p.H - py0 < p.L ? p.H - py0 : p.L;
  leff = (leff + G::RS - 1) / G::RS * G::RS;
  const int LP = leff + 2 * G::R;
  const int S = (leff + G::R + G::LAGF - 1) / G::RS + 1;
  if (tid < G::NST * 32) reinterpret_cast<float*>(sm + G1::BIAS1)[tid] = p.bm[tid >> 5][tid & 31];
  if (wave != 15) __syncthreads();  // (the DMA wave arrives at this barrier with the first two bands landed)
  // wave -> role.  Waves w, w + 4, w + 8, w + 12 share a SIMD; MFMAs / SiLU wave-values per step and SIMD: 78 / 56, 78 / 56, 78 / 56, 80 / 48.
  //   SIMD 0: t1 units 0-1, b1 unit 2, cv1 unit 0, cv2 unit 0     SIMD 1: t1 units 2-3, t2 unit 2, cv1 unit 1, cv2 unit 1
  //   SIMD 2: b1 units 0-1, b2 unit 2, cv1 unit 2, cv2 unit 2     SIMD 3: t2 units 0-1, b2 units 0-1, cv1 unit 3, the DMA wave
  switch (wave) {
    case 0: c2fs::stage_role<2, 0, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 1: c2fs::stage_role<2, 0, true>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 2: c2fs::stage_role<2, 1, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 3: c2fs::stage_role<2, 2, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 4: c2fs::stage_role<2, 1, false>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 5: c2fs::stage_role<2, 2, false>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 6: c2fs::stage_role<2, 3, false>(p, sm, 2, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 7: c2fs::stage_role<2, 3, true>(p, sm, 0, lane, S, py0, sx0, LP, G1::BIAS1); break;
    case 8: c2fs::cv1_role<2, 1, 1, 2>(p, sm, 0, 0, lane, S, py0, sx0, LP); break;
    case 9: c2fs::cv1_role<2, 1, 1, 2>(p, sm, 1, 0, lane, S, py0, sx0, LP); break;
    case 10: c2fs::cv1_role<2, 1, 1, 2>(p, sm, 2, 0, lane, S, py0, sx0, LP); break;
    case 11: c2fs::cv1_role<2, 1, 1, 2>(p, sm, 3, 0, lane, S, py0, sx0, LP); break;
    case 12: c2fs::cv2_role<2, 1, 1>(p, sm, 0, lane, S, n, py0, sx0, LP); break;
    case 13: c2fs::cv2_role<2, 1, 1>(p, sm, 1, lane, S, n, py0, sx0, LP); break;
    case 14: c2fs::cv2_role<2, 1, 1>(p, sm, 2, lane, S, n, py0, sx0, LP); break;
    default: c2fs::dma_role<2, 1, 1>(p, sm, 0, lane, S, n, py0, sx0, LP); break;
  }
}

// CUs of the current device (C++11 static initialisation: thread safe; 0 = the query failed)
static int c2fs_cus() {
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
    return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }();
  return cus;
}

// rows per part for an (n, h, w) problem: one round of workgroups if possible, as few steps as possible
static int c2fs_pick_rows(int n, int h, int w, int cus) {
  const int strips = cdiv(w, c2fs::WS);
  long best_cost = -1;
  int best = (h + 1) & ~1;
  for (int parts = 1; parts <= cdiv(h, 4); ++parts) {
    int L = cdiv(cdiv(h, parts), 2) * 2;
    if (L < 4) break;
    const long wgs = (long)n * strips * cdiv(h, L);
    const long rounds = (wgs + cus - 1) / cus;
    const long cost = rounds * (L / 2 + 10);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = L; }
  }
  return best;
}

// Called by upa_c2f_fused (c2f_fused.hip) for the C2f(64, 64, n = 2) form: UPA_EUNSUPPORTED = the caller runs the tile form.
int upa_c2f32_stream_launch(const void* x, int n, int h, int w, int ldx, int shortcut, const void* w1, const float* b1,
                            const void* const* wm, const float* const* bm, const void* w2, const float* b2, void* y, int ldy,
                            const upa_opts* opts, hipStream_t s) {
  if ((long)n * h * w * (long)(ldx > ldy ? ldx : ldy) * 2 >= (1L << 31) || (long)w * ldx * 2 >= (1L << 24)) return UPA_EUNSUPPORTED;
  const int cus = c2fs_cus();
  if (!cus) return UPA_ELAUNCH;
  C2fsParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 4; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.shortcut = shortcut ? 1 : 0;
  p.strips = cdiv(w, c2fs::WS);
  const int rows = UPA_OPT(opts, c2f_stream_rows);
  UPA_CHECK_ARG(rows <= 0 || rows >= 4, "c2f_stream_rows = %d: 0 (auto), -1 (whole height) or >= 4", rows);
  p.L = rows >= 4 ? (rows + 1) & ~1 : rows < 0 ? (h + 1) & ~1 : c2fs_pick_rows(n, h, w, cus);
  p.parts = cdiv(h, p.L);
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long wgs = (long)n * p.strips * p.parts;
  if (wgs >= (1L << 31) / 2) return UPA_EUNSUPPORTED;
  if (UPA_OPT(opts, c2f_stream) == 2) {  // the first role set (cv2 recomputes y0): kept for A/B
    if (upa_full_lds<c2f32_stream_kernel<2>>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL((c2f32_stream_kernel<2>), dim3((unsigned)wgs), dim3(1024), c2fs::Geo<2>::LDS, s, p);
    return UPA_OK;
  }
  if (upa_full_lds<c2f32_stream2_kernel>() != hipSuccess) return UPA_ELAUNCH;
  constexpr size_t lds2 = c2fs::GeoS<2, 1, 1>::LDS1;
  hipLaunchKernelGGL(c2f32_stream2_kernel, dim3((unsigned)wgs), dim3(1024), lds2, s, p);
  return UPA_OK;
}

template <int NCH>
static int c2fs1_launch(const C2fsParams& p, long wgs, hipStream_t s) {
  if (upa_full_lds<c2f32_stream1_kernel<NCH>>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL((c2f32_stream1_kernel<NCH>), dim3((unsigned)wgs), dim3(1024), c2fs::Geo1<NCH>::LDS1, s, p);
  return UPA_OK;
}

// The n = 1 form: called by upa_c2f_fused (c1 = 64) and upa_c2f32_up_fused (c1 = 64 k, optional virtual Upsample + Concat).
// UPA_EUNSUPPORTED = the caller runs the tile form.
int upa_c2f32_stream1_launch(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int shortcut,
                             const void* w1, const float* b1, const void* const* wm, const float* const* bm, const void* w2, const float* b2,
                             void* y, int ldy, const upa_opts* opts, hipStream_t s) {
  const int nch = c1 / 64;
  if (c1 % 64 != 0 || nch < 1 || nch > 3 || (up && (up_c % 64 != 0 || (h & 1) || (w & 1)))) return UPA_EUNSUPPORTED;
  const long ldm = ldx > ldy ? (ldx > up_ld ? ldx : up_ld) : (ldy > up_ld ? ldy : up_ld);
  if ((long)n * h * w * ldm * 2 >= (1L << 31) || (long)w * ldm * 2 >= (1L << 24)) return UPA_EUNSUPPORTED;
  const int cus = c2fs_cus();
  if (!cus) return UPA_ELAUNCH;
  C2fsParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 2; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.shortcut = shortcut ? 1 : 0;
  p.up = (const char*)up; p.c1 = c1; p.upC = up ? up_c : 0; p.up_ld = up_ld;
  p.strips = cdiv(w, c2fs::WS);
  const int rows = UPA_OPT(opts, c2f_stream_rows);
  UPA_CHECK_ARG(rows <= 0 || rows >= 4, "c2f_stream_rows = %d: 0 (auto), -1 (whole height) or >= 4", rows);
  p.L = rows >= 4 ? (rows + 1) & ~1 : rows < 0 ? (h + 1) & ~1 : c2fs_pick_rows(n, h, w, cus);
  p.parts = cdiv(h, p.L);
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long wgs = (long)n * p.strips * p.parts;
  if (wgs >= (1L << 31) / 2) return UPA_EUNSUPPORTED;
  return nch == 1 ? c2fs1_launch<1>(p, wgs, s) : nch == 2 ? c2fs1_launch<2>(p, wgs, s) : c2fs1_launch<3>(p, wgs, s);
}
