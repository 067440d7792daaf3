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
};

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
  static constexpr int W2S = base(NTEN);                 // cv2 A fragments [k-tile][4 n-tiles][lane][16 B]
  static constexpr int W2B = (2 + NB) * 4 * 1024;
  static constexpr int XS = W2S + W2B;                   // cv1 input staging: 2 slots of RS x XW pixels x 128 B
  static constexpr int XSLOT = NY1 * 16 * 128;
  static constexpr int XF = XS + 2 * XSLOT;              // y0 input staging: 2 slots of RS x WS pixels
  static constexpr int FSLOT = NF * 16 * 128;
  static constexpr int BIAS = XF + 2 * FSLOT;            // f32: b1[64], b2[64], bm[NST][32]
  static constexpr int LDS = BIAS + (128 + NST * 32) * 4;
};

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(&a), *reinterpret_cast<const s16x4*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 silu_pack(const f32x4& a) {
  return u32x2{pack_bf16x2(silu(a[0]), silu(a[1])), pack_bf16x2(silu(a[2]), silu(a[3]))};
}

// ---- one 3x3 stage, one unit: the wave's whole life
template <int NB, int K>
__device__ __forceinline__ void stage_role(const C2fsParams& p, char* sm, int unit, int lane, int S, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  constexpr int SD = G::sd(K), C0 = K + 1, LAG = G::lag(K);
  constexpr int RIN = G::ring(K), ROUT = G::ring(K + 1);
  constexpr bool HAS_RES = (K & 1) != 0;
  constexpr int RRES = HAS_RES ? G::ring(K - 1) : 1;
  const int g = lane >> 4, r = lane & 15;
  const int q = unit * 16 + r;
  const bool act = q < G::RS * SD;
  const int qq = act ? q : 0;
  const int rr = qq >= SD ? 1 : 0, cc = C0 + qq - rr * SD;
  static_assert(G::RS == 2, "the lane -> row map above assumes two rows per step");

  u32x4 w[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w[tap][nt] = *reinterpret_cast<const u32x4*>(p.wm[K] + ((size_t)(tap * 2 + nt) * 64 + lane) * 16);

  const int in_const = G::base(K) + g * G::plane(K) + (cc - 1) * 16;
  const int out_const = G::base(K + 1) + (g >> 1) * G::plane(K + 1) + cc * 16 + (g & 1) * 8;
  const int res_const = HAS_RES ? G::base(K - 1) + (g >> 1) * G::plane(K - 1) + cc * 16 + (g & 1) * 8 : 0;
  const int bias_off = G::BIAS + (128 + K * 32 + 4 * g) * 4;
  const int gx = sx0 - G::R + cc;
  const bool colok = gx >= 0 && gx < p.W;
  const int lo = K + 1, hi = LP - (K + 1);
  const bool use_res = HAS_RES && p.shortcut;

  for (int s = 0; s < S; ++s) {
    const int r0 = G::RS * s - LAG;
    if (r0 + G::RS > lo && r0 < hi) {  // wave-uniform
      const int row = r0 + rr;
      int rb[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) rb[dy] = in_const + ((row + dy - 1) & (RIN - 1)) * G::ROWB;
      f32x4 acc0 = *reinterpret_cast<const f32x4*>(sm + bias_off);
      f32x4 acc1 = *reinterpret_cast<const f32x4*>(sm + bias_off + 64);
      u32x2 rs0 = {0u, 0u}, rs1 = {0u, 0u};
      if (use_res) {
        const int ra = res_const + (row & (RRES - 1)) * G::ROWB;
        rs0 = *reinterpret_cast<const u32x2*>(sm + ra);
        rs1 = *reinterpret_cast<const u32x2*>(sm + ra + 2 * G::plane(HAS_RES ? K - 1 : 0));
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const u32x4 b = *reinterpret_cast<const u32x4*>(sm + rb[tap / 3] + (tap % 3) * 16);
        acc0 = mfma32(w[tap][0], b, acc0);
        acc1 = mfma32(w[tap][1], b, acc1);
      }
      const int gy = py0 - G::R + row;
      const unsigned m = (colok && gy >= 0 && gy < p.H) ? 0xFFFFFFFFu : 0u;  // the tensor is ZERO outside the image (the next conv's padding)
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
      if (act && row >= lo && row < hi) {
        const int oa = out_const + (row & (ROUT - 1)) * G::ROWB;
        *reinterpret_cast<u32x2*>(sm + oa) = o0;
        *reinterpret_cast<u32x2*>(sm + oa + 2 * G::plane(K + 1)) = o1;
      }
    }
    __syncthreads();
  }
}

// ---- cv1 (y1 into its ring), y0 + cv2 (output rows): X wave xi of NX
template <int NB>
__device__ __forceinline__ void x_role(const C2fsParams& p, char* sm, int xi, int lane, int S, int n, int py0, int sx0, int LP) {
  using G = Geo<NB>;
  constexpr int NX = 16 - (G::units(0) + G::units(1) + (NB == 2 ? G::units(2) + G::units(3) : 0));
  static_assert(G::NY1 <= 2 * NX && G::NF <= NX, "cv1 / cv2 units must fit the X waves");
  static_assert(G::RS == 2, "lane -> row maps assume two rows per step");
  const int g = lane >> 4, r = lane & 15;
  const size_t rowpitch = (size_t)p.W * p.ldx * 2;
  const char* ximg = p.x + (size_t)n * p.H * rowpitch;

  // cv1 fragments: [k-tile 0..1][n-tile 0..3] (n-tiles 0, 1 = y0, 2, 3 = y1)
  u32x4 w1f[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w1f[kt][nt] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(kt * 4 + nt) * 64 + lane) * 16);
  // cv2's k-tile 0 (y0) as two 16-wide k-steps against the D layout of the y0 accumulators: lane (g, r) = W2[co = 16 nt + r][ci = 16 ks + 4g .. + 3]
  u32x2 w2y0[2][4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int c0 = 16 * ks + 4 * g;
      const int gg = (c0 & 31) >> 3, half = (c0 & 7) >> 2;
      w2y0[ks][nt] = *reinterpret_cast<const u32x2*>(p.w2 + ((size_t)(nt * 64 + gg * 16 + r)) * 16 + half * 8);
    }

  // ---- lane constants of the (up to two) cv1 units and of the cv2 unit
  int y1_q[2], y1_out[2], y1_row[2];
  bool y1_act[2], y1_col[2];
  unsigned y1_dma[2][2];  // per DMA piece: column part of the global offset
  int y1_drow[2][2];
  bool y1_dok[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = xi + t * NX;
    const int q = u * 16 + r;
    y1_q[t] = q;
    y1_act[t] = u < G::NY1 && q < G::RS * G::XW;
    const int qq = y1_act[t] ? q : 0;
    const int rr = qq >= G::XW ? 1 : 0, col = qq - rr * G::XW;
    y1_row[t] = rr;
    y1_out[t] = G::base(0) + (g >> 1) * G::plane(0) + col * 16 + (g & 1) * 8;
    const int gx = sx0 - G::R + col;
    y1_col[t] = gx >= 0 && gx < p.W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pd = u * 16 + 8 * i + (lane >> 3);
      y1_dok[t][i] = u < G::NY1 && (u * 16 + 8 * i) < G::RS * G::XW;  // wave-uniform: the piece holds at least one pixel
      const int pq = pd < G::RS * G::XW ? pd : 0;
      const int drr = pq >= G::XW ? 1 : 0, dcol = pq - drr * G::XW;
      int dgx = sx0 - G::R + dcol;
      dgx = dgx < 0 ? 0 : (dgx >= p.W ? p.W - 1 : dgx);
      const int cg = (lane & 7) ^ (pd & 7);
      y1_drow[t][i] = drr;
      y1_dma[t][i] = (unsigned)dgx * (unsigned)p.ldx * 2u + (unsigned)cg * 16u;
    }
  }
  const int fu = xi;  // the cv2 unit
  const int fq = fu * 16 + r;
  const bool f_act = fu < G::NF && fq < G::RS * WS;
  const int fqq = f_act ? fq : 0;
  const int f_rr = fqq >= WS ? 1 : 0, f_oc = fqq - f_rr * WS;
  const int f_col = G::R + f_oc;
  const bool f_colok = sx0 + f_oc < p.W;
  unsigned f_dma[2];
  int f_drow[2];
  bool f_dok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pd = fu * 16 + 8 * i + (lane >> 3);
    f_dok[i] = fu < G::NF && (fu * 16 + 8 * i) < G::RS * WS;
    const int pq = pd < G::RS * WS ? pd : 0;
    const int drr = pq >= WS ? 1 : 0, doc = pq - drr * WS;
    int dgx = sx0 + doc;
    dgx = dgx >= p.W ? p.W - 1 : dgx;
    const int cg = (lane & 7) ^ (pd & 7);
    f_drow[i] = drr;
    f_dma[i] = (unsigned)dgx * (unsigned)p.ldx * 2u + (unsigned)cg * 16u;
  }
  int f_in[3];  // B fragments of y1, b1 (, b2) at the output pixel
#pragma unroll
  for (int k = 0; k < 1 + NB; ++k) f_in[k] = G::base(2 * k) + g * G::plane(2 * k) + f_col * 16;
  const size_t f_store = ((size_t)n * p.H * p.W) * (size_t)p.ldy * 2 + (size_t)(sx0 + f_oc) * p.ldy * 2;

  // DMA of the rows step `st` consumes (cv1: y1 rows RS st ..; cv2: output rows RS st - LAGF ..) into staging slot st & 1
  auto stage_in = [&](int st) __attribute__((always_inline)) {
    const int slot = st & 1;
    if (G::RS * st < LP) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (y1_dok[t][i]) {
            int gy = py0 - G::R + G::RS * st + y1_drow[t][i];
            gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
            const char* src = ximg + (size_t)gy * rowpitch + y1_dma[t][i];
            __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(sm + G::XS + slot * G::XSLOT + ((xi + t * NX) * 16 + 8 * i) * 128), 16, 0, 0);
          }
    }
    const int o0 = G::RS * st - G::LAGF;
    if (o0 + G::RS > G::R && o0 < LP - G::R) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (f_dok[i]) {
          int gy = py0 - G::R + o0 + f_drow[i];
          gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
          const char* src = ximg + (size_t)gy * rowpitch + f_dma[i];
          __builtin_amdgcn_global_load_lds((cgptr_t)src, (clptr_t)(sm + G::XF + slot * G::FSLOT + (fu * 16 + 8 * i) * 128), 16, 0, 0);
        }
    }
  };

  stage_in(0);
  for (int s = 0; s < S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own pieces of step s have landed (nobody else reads them)
    stage_in(s + 1);
    const int slot = s & 1;
    // ---- cv1 upper half -> y1 rows RS s, RS s + 1
    if (G::RS * s < LP) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (xi + t * NX >= G::NY1) break;  // wave-uniform
        const int q = y1_q[t];
        const char* xb = sm + G::XS + slot * G::XSLOT + q * 128;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(sm + G::BIAS + (32 + 4 * g) * 4);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(sm + G::BIAS + (48 + 4 * g) * 4);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(xb + (((kt * 4 + g) ^ (q & 7)) << 4));
          a0 = mfma32(w1f[kt][2], b, a0);
          a1 = mfma32(w1f[kt][3], b, a1);
        }
        const int row = G::RS * s + y1_row[t];
        const int gy = py0 - G::R + row;
        const unsigned m = (y1_col[t] && gy >= 0 && gy < p.H) ? 0xFFFFFFFFu : 0u;
        u32x2 o0 = silu_pack(a0), o1 = silu_pack(a1);
        o0[0] &= m; o0[1] &= m; o1[0] &= m; o1[1] &= m;
        if (y1_act[t] && row < LP) {
          const int oa = y1_out[t] + (row & (G::ring(0) - 1)) * G::ROWB;
          *reinterpret_cast<u32x2*>(sm + oa) = o0;
          *reinterpret_cast<u32x2*>(sm + oa + 2 * G::plane(0)) = o1;
        }
      }
    }
    // ---- y0 and cv2 on output rows RS s - LAGF ..
    const int o0r = G::RS * s - G::LAGF;
    if (fu < G::NF && o0r + G::RS > G::R && o0r < LP - G::R) {
      const int row = o0r + f_rr;
      const char* xb = sm + G::XF + slot * G::FSLOT + fq * 128;
      f32x4 y0a = *reinterpret_cast<const f32x4*>(sm + G::BIAS + (0 + 4 * g) * 4);
      f32x4 y0b = *reinterpret_cast<const f32x4*>(sm + G::BIAS + (16 + 4 * g) * 4);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const u32x4 b = *reinterpret_cast<const u32x4*>(xb + (((kt * 4 + g) ^ (fq & 7)) << 4));
        y0a = mfma32(w1f[kt][0], b, y0a);
        y0b = mfma32(w1f[kt][1], b, y0b);
      }
      const u32x2 y0B[2] = {silu_pack(y0a), silu_pack(y0b)};
      u32x4 opnd[1 + NB];
#pragma unroll
      for (int k = 0; k < 1 + NB; ++k) opnd[k] = *reinterpret_cast<const u32x4*>(sm + f_in[k] + (row & (G::ring(2 * k) - 1)) * G::ROWB);
      f32x4 o[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        o[nt] = *reinterpret_cast<const f32x4*>(sm + G::BIAS + (64 + nt * 16 + 4 * g) * 4);
        o[nt] = mfma16(w2y0[0][nt], y0B[0], o[nt]);
        o[nt] = mfma16(w2y0[1][nt], y0B[1], o[nt]);
      }
      // A 4-pass 16x16x16 MFMA whose result is the NEXT instruction's srcC of an 8-pass 16x16x32 MFMA came out wrong in rows 2, 3
      // of each lane's four (measured, ROCm 7.2 / gfx950: hipcc inserts no wait states between the two shapes on one accumulator):
      // finish every 16-wide chain first, then wait out the short pipeline before the 32-wide chains start.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int k = 0; k < 1 + NB; ++k) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(sm + G::W2S + (((k + 1) * 4 + nt) * 64 + lane) * 16);
          o[nt] = mfma32(a, opnd[k], o[nt]);
        }
      }
      const int gy = py0 - G::R + row;
      const bool st_ok = f_act && f_colok && row >= G::R && row < LP - G::R && gy < p.H;
      char* dst = p.y + f_store + (size_t)gy * p.W * p.ldy * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x2 a = silu_pack(o[2 * j]), b = silu_pack(o[2 * j + 1]);
        auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
        const int cb = 16 * (2 * j + (g & 1)) + 8 * (g >> 1);
        if (st_ok) *reinterpret_cast<u32x4*>(dst + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
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

  // cv2 fragments and every bias -> LDS (read by the X waves each step: they have no registers to spare for them)
  for (int i = tid; i < G::W2B / 16; i += 1024) *reinterpret_cast<u32x4*>(sm + G::W2S + i * 16) = *reinterpret_cast<const u32x4*>(p.w2 + (size_t)i * 16);
  if (tid < 64) reinterpret_cast<float*>(sm + G::BIAS)[tid] = p.b1[tid];
  else if (tid < 128) reinterpret_cast<float*>(sm + G::BIAS)[tid] = p.b2[tid - 64];
  else if (tid < 128 + G::NST * 32) reinterpret_cast<float*>(sm + G::BIAS)[tid] = p.bm[(tid - 128) >> 5][(tid - 128) & 31];
  __syncthreads();

  // wave -> role.  Waves w, w + 4, w + 8, w + 12 share a SIMD: each SIMD gets at most one X wave (the VALU-heavy role).
  switch (wave) {
    case 0: case 1: case 2: case 3: c2fs::stage_role<NB, 0>(p, sm, wave, lane, S, py0, sx0, LP); break;
    case 4: case 5: case 6: c2fs::stage_role<NB, 1>(p, sm, wave - 4, lane, S, py0, sx0, LP); break;
    case 7: case 8: case 9: c2fs::stage_role<NB, 2>(p, sm, wave - 7, lane, S, py0, sx0, LP); break;
    case 10: case 11: c2fs::stage_role<NB, 3>(p, sm, wave - 10, lane, S, py0, sx0, LP); break;
    case 15: c2fs::stage_role<NB, 3>(p, sm, 2, lane, S, py0, sx0, LP); break;
    default: c2fs::x_role<NB>(p, sm, wave - 12, lane, S, n, py0, sx0, LP); break;
  }
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
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return UPA_ELAUNCH;
    cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  C2fsParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.b1 = b1; p.b2 = b2;
  for (int i = 0; i < 4; ++i) { p.wm[i] = (const char*)wm[i]; p.bm[i] = bm[i]; }
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.shortcut = shortcut ? 1 : 0;
  p.strips = cdiv(w, c2fs::WS);
  const int rows = UPA_OPT(opts, c2f_stream_rows);
  p.L = rows >= 4 ? (rows + 1) & ~1 : c2fs_pick_rows(n, h, w, cus);
  p.parts = cdiv(h, p.L);
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long wgs = (long)n * p.strips * p.parts;
  if (wgs >= (1L << 31) / 2) return UPA_EUNSUPPORTED;
  if (upa_full_lds<c2f32_stream_kernel<2>>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL((c2f32_stream_kernel<2>), dim3((unsigned)wgs), dim3(1024), c2fs::Geo<2>::LDS, s, p);
  return UPA_OK;
}
