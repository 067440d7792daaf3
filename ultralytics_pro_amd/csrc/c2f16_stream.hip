// C2f(32 -> 32, n = 1 Bottleneck of 16 channels, shortcut, bf16) as ONE kernel in LINE-BUFFER form: model.2 of yolov8n at 160 x 160.
//   C2f.forward        ultralytics/nn/modules/block.py:457-488   y = list(cv1(x).chunk(2, 1)); y.extend(m(y[-1]) ...); cv2(cat(y, 1))
//   Bottleneck.forward ultralytics/nn/modules/block.py:644-668   x + cv2(cv1(x))
// The tile form (c2f16_fused_kernel, c2f_fused.hip: one 16 x 16 output tile per workgroup, every intermediate an LDS tile) recomputes a
// 4-pixel halo ring per tile (cv1 2.25x, the 3x3 convs 1.9 / 1.3x) and spends 16.2 M vector instructions per launch where the block's SiLU
// evaluations need 6.1 M: 50 us one step at a time, 62 us with four steps in flight - chip-filling, so a tenth of a step's CU time.
// Here (the form of c2f_stream.hip / detect_stream.hip, sized for this block's tiny per-row work): a workgroup of FOUR waves owns a 20-column strip
// of one image and L output rows and walks down it two rows per step with one s_barrier; y = cv1(x) (24 columns), t (22), b (20) only exist
// as planar LDS rings [8-channel group][row][column][16 B] of 16 / 8 / 4 rows; one wave per stage, weights in registers:
//   wave 0  cv1 (1x1, 32 -> 32) on the band the LDS-DMA brought in + that DMA itself, FOUR bands ahead (a step is ~1 k cycles, a memory round
//           trip 2-4 k: counted s_waitcnt vmcnt, this wave issues no other vector-memory operation);
//   wave 1  t = SiLU(m.cv1(y1)): 3x3 with 16 input channels = HALF a 32-wide MFMA k-step, so a k-step pairs two taps (lane groups 0-1 take
//           tap 2s, groups 2-3 tap 2s + 1): 5 k-steps instead of 9 (as the fused stem's second conv, stem.hip);
//   wave 2  b = y1 + SiLU(m.cv2(t)), f32 shortcut add;
//   wave 3  out = SiLU(cv2([y0 | y1 | b])): one 32-wide k-step straight from the y ring (its four planes ARE y0 | y1) + a 16-wide one from the
//           b ring (v_mfma_f32_16x16x16_bf16, after the 32-wide chain, fenced), 16-byte NHWC stores (n-tile pairs through v_permlane16_swap).
// 73 KB of LDS and 4 waves: two workgroups per CU.  Halo recompute only across the strip (cv1 1.2x, t 1.1x), none down it.
// Rounding points (bf16 y, t, b, out; f32 accumulation from the bias; f32 shortcut add) are those of the separate launches and of the tile form.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((address_space(1))) const void* c16_gptr_t;
typedef __attribute__((address_space(3))) void* c16_lptr_t;
typedef __attribute__((ext_vector_type(4))) short c16_s16x4;

__device__ __attribute__((aligned(64))) unsigned int g_c16_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct C16Params {
  const char* x; char* y;
  const char *w1, *wa, *wb, *w2;   // cv1 (1x1 32 -> 32), m.cv1 / m.cv2 (3x3 16 -> 16), cv2 (1x1 48 -> 32): upa_pack_conv_weight(bf16) layouts
  const float *b1, *ba, *bb, *b2;
  const char* wd; const float* bd; // DOWN form: the following Conv(32, 64, 3, 2) (3x3 stride 2 pad 1), its output is what y receives
  int N, H, W, ldx, ldy, strips, parts, L, xcd;
  int OH, OW;                      // DOWN form: size of the stride-2 output
};

// profiling build (-DUPA_STAMP): every wave of workgroups 0-3 records s_memtime at the start of each step and before its barrier
#ifdef UPA_STAMP
#define C16_STAMP_STEPS 64
__device__ unsigned long long g_c16_stamps[4 * 12 * C16_STAMP_STEPS * 2];
extern "C" int upa_debug_stamps_c2f16s(unsigned long long* out, int count) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c16_stamps), (size_t)count * 8) == hipSuccess ? 0 : -1;
}
#define C16_STAMP(step, which)                                                                               \
  do {                                                                                                       \
    if (blockIdx.x < 4 && (step) < C16_STAMP_STEPS) {                                                        \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      if ((threadIdx.x & 63) == 0) g_c16_stamps[((blockIdx.x * 12 + (threadIdx.x >> 6)) * C16_STAMP_STEPS + (step)) * 2 + (which)] = t_; \
    }                                                                                                        \
  } while (0)
#else
#define C16_STAMP(step, which) do {} while (0)
#endif

namespace c16s {
constexpr int PREF = 2;            // input bands in flight ahead of the one cv1 reads (the other workgroups of the CU hide the rest of the round trip)
// Geometry.  Plain form: a strip = 20 output columns.  DOWN form (the block + the stride-2 Conv behind it): a strip = 10 columns of the stride-2
// output = 21 columns of the block's output (2 ox - 1 .. 2 ox + 1), which then only exists as a ring `o` as well.
template <bool DOWN> struct Geo {
  static constexpr int WO = DOWN ? 21 : 20;           // columns of b / out
  static constexpr int WT = WO + 2, XW = WO + 4;      // columns of t, of x / y
  static constexpr int SW = DOWN ? 10 : 20;           // columns of the strip in the tensor the kernel writes
  static constexpr int XSLOTS = 32, XROWB = XSLOTS * 16, XROWS = 8, XPLANE = XROWS * XROWB;   // x ring: 4 planes, PREF + 2 bands alive
  static constexpr int YROWB = (DOWN ? 32 : 24) * 16, YROWS = 16, YPLANE = YROWS * YROWB;      // y ring: 4 planes (y0 | y1), rows 2s - 8 .. 2s + 1 alive
  static constexpr int TROWB = 24 * 16;                                                       // t / b / o rings: <= 23 columns
  static constexpr int TROWS = 8, TPLANE = TROWS * TROWB;                                     // t: 2 planes
  static constexpr int BROWS = 4, BPLANE = BROWS * TROWB;                                     // b: 2 planes
  static constexpr int OROWS = 8, OPLANE = OROWS * TROWB;                                     // o (DOWN): 4 planes, rows 2s - 12 .. 2s - 7 alive
  static constexpr int XB = 0, YB = XB + 4 * XPLANE, TB = YB + 4 * YPLANE, BB = TB + 2 * TPLANE, OB = BB + 2 * BPLANE;
  static constexpr int DUMMY = OB + (DOWN ? 4 * OPLANE : 0), LDS = DUMMY + 512;
  static constexpr int NUC = (2 * XW + 15) / 16;      // cv1 units per band: 3 | 4
  static_assert(XPLANE % 256 == 0 && YPLANE % 256 == 0 && TPLANE % 256 == 0 && BPLANE % 256 == 0 && OPLANE % 256 == 0,
                "planes keep the ds_read_b128 lane groups on disjoint banks");
};
// step s: cv1 -> y rows {2s, 2s + 1};  t rows {2s - 3, 2s - 2};  b rows {2s - 6, 2s - 5};  out rows {2s - 8, 2s - 7};  DOWN: stride-2 row s - 7
// (row coordinate i = image row ry0 + i, column coordinate of y = image column cx0 + column; x / y rows [0, LP), t rows [1, LP - 1),
//  b / out rows [2, LP - 2); plain: ry0 = py0 - 2, cx0 = sx0 - 2; DOWN: ry0 = 2 dy0 - 3, cx0 = 2 ox0 - 3)

struct Ctx {
  const C16Params* p;
  char* sm;
  int lane, n, ry0, cx0, LP, S;
  int oy0, ox0;   // DOWN: first stride-2 output row / column of this workgroup
};

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const u32x2& a, const u32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const c16_s16x4*>(&a), *reinterpret_cast<const c16_s16x4*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const char* sm, int off) { return *reinterpret_cast<const u32x4*>(sm + off); }
__device__ __forceinline__ u32x2 lds64(const char* sm, int off) { return *reinterpret_cast<const u32x2*>(sm + off); }
__device__ __forceinline__ u32x2 silu_pack(const f32x4& a, unsigned m) {
  return u32x2{pack_bf16x2(silu(a[0]), silu(a[1])) & m, pack_bf16x2(silu(a[2]), silu(a[3])) & m};
}

// ---- cv1 (1x1, 32 -> 32) on units [U0, U0 + NU) of the band that has landed; with DMA: also the input bands by LDS-DMA, PREF ahead.  Band b = x rows
// {2b, 2b + 1}; one instruction = one 8-channel plane of a band (2 rows x 32 slots, 24 | 25 used); lane = slot (row lane >> 5, column lane & 31)
template <bool DOWN, int U0, int NU, bool DMA>
struct Cv1 {
  using G = Geo<DOWN>;
  u32x4 w[2];
  f32x4 bias[2];
  const char* ximg;
  unsigned rowpitch, coloff;
  bool colok;
  int u_rr[NU], u_in[NU], u_out[NU];
  unsigned u_colm[NU];
  bool u_act[NU];
  __device__ __forceinline__ void init(const Ctx& x) {
    const C16Params& p = *x.p;
    const int lane = x.lane, g = lane >> 4, r = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      w[nt] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)nt * 64 + lane) * 16);
      bias[nt] = *reinterpret_cast<const f32x4*>(p.b1 + nt * 16 + 4 * g);
    }
    rowpitch = (unsigned)p.W * (unsigned)p.ldx * 2u;
    ximg = p.x + (size_t)x.n * p.H * rowpitch;
    const int xc = lane & 31, gx = x.cx0 + xc;
    colok = xc < G::XW && gx >= 0 && gx < p.W;
    coloff = colok ? (unsigned)gx * (unsigned)p.ldx * 2u : 0u;
#pragma unroll
    for (int u = 0; u < NU; ++u) {  // the band's 48 | 50 pixels in 16-pixel units; this wave: units [U0, U0 + NU)
      const int q = 16 * (U0 + u) + r;
      u_act[u] = q < 2 * G::XW;
      const int qq = u_act[u] ? q : 0;
      u_rr[u] = qq >= G::XW ? 1 : 0;
      const int col = qq - u_rr[u] * G::XW;
      u_in[u] = G::XB + g * G::XPLANE + u_rr[u] * G::XROWB + col * 16;
      // (two ds_write_b64 per unit at a 16-byte pitch: 2-way bank conflicts; pairing the n-tiles with v_permlane16_swap into ONE conflict-free
      // ds_write_b128 - here and for the o ring - was measured slower: 60.0 vs 56.6 us, same box)
      u_out[u] = G::YB + (g >> 1) * G::YPLANE + col * 16 + (g & 1) * 8;   // n-tile nt: + 2 nt planes
      const int gxx = x.cx0 + col;
      u_colm[u] = (gxx >= 0 && gxx < p.W) ? 0xFFFFFFFFu : 0u;
    }
  }
  __device__ __forceinline__ int band(const Ctx& x, int b) {
    if (2 * b >= x.LP) return 0;  // wave-uniform
    const int gy = x.ry0 + 2 * b + (x.lane >> 5);
    const bool ok = colok && gy >= 0 && gy < x.p->H;
    const char* src = ok ? ximg + ((unsigned)gy * rowpitch + coloff) : reinterpret_cast<const char*>(g_c16_zero_page);
    const int dst = G::XB + ((2 * b) & (G::XROWS - 1)) * G::XROWB;  // + lane * 16 by the hardware
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
      __builtin_amdgcn_global_load_lds((c16_gptr_t)(src + (ok ? cg * 16 : 0)), (c16_lptr_t)(x.sm + dst + cg * G::XPLANE), 16, 0, 0);
    return 4;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    char* sm = x.sm;
    if (2 * s < x.LP) {
      const int r0 = 2 * s;
      u32x4 bx[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) bx[u] = lds128(sm, u_in[u] + (r0 & (G::XROWS - 1)) * G::XROWB);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const f32x4 a0 = mfma32(w[0], bx[u], bias[0]), a1 = mfma32(w[1], bx[u], bias[1]);
        const int row = r0 + u_rr[u];
        const int gy = x.ry0 + row;
        const unsigned m = (gy >= 0 && gy < x.p->H) ? u_colm[u] : 0u;  // y is ZERO outside the image (the 3x3's padding)
        const int oa = u_act[u] ? u_out[u] + (row & (G::YROWS - 1)) * G::YROWB : G::DUMMY + x.lane * 8;
        *reinterpret_cast<u32x2*>(sm + oa) = silu_pack(a0, m);
        *reinterpret_cast<u32x2*>(sm + (u_act[u] ? oa + 2 * G::YPLANE : oa)) = silu_pack(a1, m);
      }
    }
  }
};

// ---- a 3x3 stage with 16 input and 16 output channels on units [U0, U0 + NU) of its band.  STG 0: t from y1 (y planes 2, 3; rows {2s - 3, 2s - 2});
// STG 1: b = y1 + SiLU(conv(t)) (rows {2s - 6, 2s - 5}).  k-step ks pairs taps 2 ks (lane groups 0-1) and 2 ks + 1 (groups 2-3).
template <bool DOWN, int STG, int U0, int NU>
struct Conv3 {
  using G = Geo<DOWN>;
  static constexpr int SD = STG ? G::WO : G::WT, LAG = STG ? 6 : 3, LO = STG ? 2 : 1;
  static constexpr int IN_B = STG ? G::TB : G::YB + 2 * G::YPLANE, IN_PLANE = STG ? G::TPLANE : G::YPLANE;
  static constexpr int IN_ROWB = STG ? G::TROWB : G::YROWB, IN_MASK = (STG ? G::TROWS : G::YROWS) - 1;
  static constexpr int OUT_B = STG ? G::BB : G::TB, OUT_PLANE = STG ? G::BPLANE : G::TPLANE, OUT_MASK = (STG ? G::BROWS : G::TROWS) - 1;
  u32x4 w[5];
  f32x4 bias;
  int u_rr[NU], u_col[NU], tdy[5], tdx16[5], g, r;
  unsigned u_colm[NU];
  bool u_act[NU];
  __device__ __forceinline__ void init(const Ctx& x) {
    const C16Params& p = *x.p;
    const int lane = x.lane;
    g = lane >> 4; r = lane & 15;
    const char* wp = STG ? p.wb : p.wa;
    const float* bp = STG ? p.bb : p.ba;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int tap = 2 * ks + (g >> 1);  // lane (g, r): W[co = r][ci = 8 (g & 1) .. + 7][tap] = packed lane ((g & 1) * 16 + r) of that tap's k-tile
      w[ks] = tap < 9 ? *reinterpret_cast<const u32x4*>(wp + ((size_t)tap * 64 + (g & 1) * 16 + r) * 16) : u32x4{0u, 0u, 0u, 0u};
      const int tt = tap < 9 ? tap : 8;    // (the tenth half-step multiplies zero weights: read tap 8's pixel again)
      tdy[ks] = tt / 3;
      tdx16[ks] = (tt % 3) * 16;
    }
    bias = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = 16 * (U0 + u) + r;
      u_act[u] = q < 2 * SD;
      const int qq = u_act[u] ? q : 0;
      u_rr[u] = qq >= SD ? 1 : 0;
      u_col[u] = qq - u_rr[u] * SD;                 // output column; tap (dy, dx) reads input column col + dx (both stages)
      const int gx = x.cx0 + (STG ? 2 : 1) + u_col[u];
      u_colm[u] = (gx >= 0 && gx < p.W) ? 0xFFFFFFFFu : 0u;   // (t: the next conv's padding; b: the stride-2 conv's, DOWN form)
    }
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    const int r0 = 2 * s - LAG;
    if (!(r0 + 2 > LO && r0 < x.LP - LO)) return;  // wave-uniform
    char* sm = x.sm;
    const int lane_in = IN_B + (g & 1) * IN_PLANE;
    u32x4 b[NU][5];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
        b[u][ks] = lds128(sm, lane_in + ((r0 + u_rr[u] + tdy[ks] - 1) & IN_MASK) * IN_ROWB + u_col[u] * 16 + tdx16[ks]);
    f32x4 acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = bias;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[u] = mfma32(w[ks], b[u][ks], acc[u]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int row = r0 + u_rr[u];
      const int gy = x.ry0 + row;
      const bool ok = u_act[u] && row >= LO && row < x.LP - LO;
      const int oa = OUT_B + (g >> 1) * OUT_PLANE + (row & OUT_MASK) * G::TROWB + u_col[u] * 16 + (g & 1) * 8;
      u32x2 o;
      if constexpr (STG == 0) {
        const unsigned m = (gy >= 0 && gy < x.p->H) ? u_colm[u] : 0u;  // t is ZERO outside the image (m.cv2's padding)
        o = silu_pack(acc[u], m);
      } else {
        // the shortcut: y1 (y planes 2, 3) at the same pixel = y column col + 2; f32 add, then the bf16 rounding of the separate launches
        const u32x2 rs = lds64(sm, G::YB + (2 + (g >> 1)) * G::YPLANE + (row & (G::YROWS - 1)) * G::YROWB + (u_col[u] + 2) * 16 + (g & 1) * 8);
        const float v0 = silu(acc[u][0]) + __uint_as_float(rs[0] << 16), v1 = silu(acc[u][1]) + __uint_as_float(rs[0] & 0xFFFF0000u);
        const float v2 = silu(acc[u][2]) + __uint_as_float(rs[1] << 16), v3 = silu(acc[u][3]) + __uint_as_float(rs[1] & 0xFFFF0000u);
        o = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
      }
      *reinterpret_cast<u32x2*>(sm + (ok ? oa : G::DUMMY + x.lane * 8)) = o;
    }
  }
};

// ---- cv2 over [y0 | y1 | b] on units [U0, U0 + NU) of the output band (rows {2s - 8, 2s - 7}).  Plain: 16-byte NHWC stores.  DOWN: into the o ring
// (4 planes), ZERO outside the image - the stride-2 conv's padding.
template <bool DOWN, int U0, int NU>
struct Cv2 {
  using G = Geo<DOWN>;
  u32x4 w32[2];
  u32x2 w16[2];
  f32x4 bias[2];
  int u_rr[NU], u_col[NU], g, r;
  unsigned u_colm[NU];
  bool u_act[NU];
  char* ybase;
  size_t yrow;
  __device__ __forceinline__ void init(const Ctx& x) {
    const C16Params& p = *x.p;
    const int lane = x.lane;
    g = lane >> 4; r = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      w32[nt] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)(0 * 2 + nt) * 64 + lane) * 16);   // k-tile 0: input channels 0 .. 31 = y0 | y1
      // k-tile 1 holds b (input channels 32 .. 47): lane (g, r) of the 16-wide step needs W[r][32 + 4g .. + 3] = half (g & 1) of packed lane (g >> 1, r)
      w16[nt] = *reinterpret_cast<const u32x2*>(p.w2 + ((size_t)(1 * 2 + nt) * 64 + (g >> 1) * 16 + r) * 16 + (g & 1) * 8);
      bias[nt] = *reinterpret_cast<const f32x4*>(p.b2 + nt * 16 + 4 * g);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = 16 * (U0 + u) + r;
      u_act[u] = q < 2 * G::WO;
      const int qq = u_act[u] ? q : 0;
      u_rr[u] = qq >= G::WO ? 1 : 0;
      u_col[u] = qq - u_rr[u] * G::WO;
      const int gx = x.cx0 + 2 + u_col[u];
      u_colm[u] = (gx >= 0 && gx < p.W) ? 0xFFFFFFFFu : 0u;
      if (!DOWN) u_act[u] = u_act[u] && gx < p.W;
    }
    if constexpr (!DOWN) {
      yrow = (size_t)p.W * p.ldy * 2;
      // after the n-tile pairing (v_permlane16_swap) lane (g, r) holds channels 16 (g & 1) + 8 (g >> 1) .. + 7 of its pixel
      ybase = p.y + ((size_t)x.n * p.H * p.W + (x.cx0 + 2)) * (size_t)p.ldy * 2 + (16 * (g & 1) + 8 * (g >> 1)) * 2;
    }
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    const int r0 = 2 * s - 8;
    if (!(r0 + 2 > 2 && r0 < x.LP - 2)) return;  // wave-uniform
    char* sm = x.sm;
    u32x4 by[NU];
    u32x2 bb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int row = r0 + u_rr[u];
      by[u] = lds128(sm, G::YB + g * G::YPLANE + (row & (G::YROWS - 1)) * G::YROWB + (u_col[u] + 2) * 16);
      bb[u] = lds64(sm, G::BB + (g >> 1) * G::BPLANE + (row & (G::BROWS - 1)) * G::TROWB + u_col[u] * 16 + (g & 1) * 8);
    }
    f32x4 o[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) o[u][nt] = mfma32(w32[nt], by[u], bias[nt]);
    // a 4-pass MFMA must not take the result of an 8-pass one as the NEXT instruction's srcC without wait states (c2f_stream.hip: f_role)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) o[u][nt] = mfma16(w16[nt], bb[u], o[u][nt]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int row = r0 + u_rr[u];
      const int gy = x.ry0 + row;
      if constexpr (DOWN) {
        const unsigned m = (gy >= 0 && gy < x.p->H) ? u_colm[u] : 0u;
        const bool ok = u_act[u] && row >= 2 && row < x.LP - 2;
        // the o ring holds a row DE-INTERLEAVED - even columns in slots 0-10, odd ones in 12-21 - so that the stride-2 conv's ten pixels read
        // consecutive slots (column 2 r + dx): with the columns in order every ds_read_b128 there was a 2-3-way bank conflict, 36 reads per step
        // lane (g, r): channels 16 nt + 4g .. + 3 -> plane 2 nt + (g >> 1), half (g & 1)
        const int oa = G::OB + (g >> 1) * G::OPLANE + (row & (G::OROWS - 1)) * G::TROWB + ((u_col[u] >> 1) + (u_col[u] & 1) * 12) * 16 + (g & 1) * 8;
        *reinterpret_cast<u32x2*>(sm + (ok ? oa : G::DUMMY + x.lane * 8)) = silu_pack(o[u][0], m);
        *reinterpret_cast<u32x2*>(sm + (ok ? oa + 2 * G::OPLANE : G::DUMMY + x.lane * 8)) = silu_pack(o[u][1], m);
      } else {
        const u32x2 a = silu_pack(o[u][0], 0xFFFFFFFFu), b = silu_pack(o[u][1], 0xFFFFFFFFu);
        auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
        if (u_act[u] && row >= 2 && row < x.LP - 2 && gy < x.p->H)
          *reinterpret_cast<u32x4*>(ybase + (size_t)gy * yrow + (size_t)u_col[u] * x.p->ldy * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
  }
};

// ---- DOWN form only: the stride-2 Conv(32, 64, 3, 2) on the o ring, output n-tiles [T0, T0 + NT): at step s the ONE stride-2 row d = s - 7 (o rows
// 2d + 2 .. 2d + 4), ten pixels (lanes r < 10; o column 2 r + dx), one 32-wide k-step per tap and n-tile, weights in registers; NHWC stores of the
// (n, OH, OW, 64) output: 16 bytes per lane when the wave holds an n-tile PAIR (v_permlane16_swap), 8 bytes with a single n-tile
template <int T0, int NT>
struct Down {
  using G = Geo<true>;
  u32x4 w[9][NT];
  f32x4 bias[NT];
  int in0, g, r;
  bool act;
  char* ybase;
  size_t yrow;
  __device__ __forceinline__ void init(const Ctx& x) {
    const C16Params& p = *x.p;
    const int lane = x.lane;
    g = lane >> 4; r = lane & 15;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) w[tap][nt] = *reinterpret_cast<const u32x4*>(p.wd + ((size_t)(tap * 4 + T0 + nt) * 64 + lane) * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(p.bd + (T0 + nt) * 16 + 4 * g);
    const int rr = r < 10 ? r : 9;
    act = r < 10 && x.ox0 + r < p.OW;
    // o column 2 r + dx = slot r (dx 0), 12 + r (dx 1), r + 1 (dx 2) of the de-interleaved row: the sixteen lanes of a lane group read sixteen
    // consecutive slots (lanes 10-15: slots nobody wrote - or the first bytes behind the ring, inside the allocation - for outputs nobody stores)
    in0 = G::OB + g * G::OPLANE + r * 16;
    yrow = (size_t)p.OW * p.ldy * 2;
    // NT = 2: after the n-tile pairing lane (g, r) holds channels 16 T0 + 16 (g & 1) + 8 (g >> 1) .. + 7; NT = 1: channels 16 T0 + 4 g .. + 3
    const int ch = NT == 2 ? 16 * T0 + 16 * (g & 1) + 8 * (g >> 1) : 16 * T0 + 4 * g;
    ybase = p.y + ((size_t)x.n * p.OH * p.OW + (x.ox0 + rr)) * (size_t)p.ldy * 2 + ch * 2;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    const int d = s - 7;
    if (d < 0 || x.oy0 + d >= x.p->OH || 2 * d + 4 >= x.LP - 2) return;  // wave-uniform
    const char* sm = x.sm;
    u32x4 b[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) b[tap] = lds128(sm, in0 + ((2 * d + 2 + tap / 3) & (G::OROWS - 1)) * G::TROWB + (tap % 3 == 1 ? 12 * 16 : (tap % 3) * 8));
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = bias[nt];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] = mfma32(w[tap][nt], b[tap], o[nt]);
    char* dst = ybase + (size_t)(x.oy0 + d) * yrow;
    if constexpr (NT == 2) {
      const u32x2 a = silu_pack(o[0], 0xFFFFFFFFu), c = silu_pack(o[1], 0xFFFFFFFFu);
      auto lo = __builtin_amdgcn_permlane16_swap(a[0], c[0], false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(a[1], c[1], false, false);
      if (act) *reinterpret_cast<u32x4*>(dst) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    } else {
      const u32x2 a = silu_pack(o[0], 0xFFFFFFFFu);
      if (act) *reinterpret_cast<u32x2*>(dst) = a;
    }
  }
};

template <typename R>
__device__ __forceinline__ void run(const Ctx& x, R& r) {
  r.init(x);
  __syncthreads();
  for (int s = 0; s < x.S; ++s) { C16_STAMP(s, 0); r.step(x, s); C16_STAMP(s, 1); __syncthreads(); }
}
// the wave that also stages the input: band 0 has landed once everything but the PREF younger bands is back (it issues no other vector-memory
// operation); at the end of step s band s + 1 must have - at most PREF bands (the youngest) may still be in flight, fewer near the end of the part
template <typename R>
__device__ __forceinline__ void run_dma(const Ctx& x, R& r) {
  r.init(x);
  for (int b = 0; b <= PREF; ++b) r.band(x, b);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PREF) : "memory");
  if (2 * PREF >= x.LP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (short parts issue fewer bands than that)
  __syncthreads();
  for (int s = 0; s < x.S; ++s) {
    C16_STAMP(s, 0);
    const int issued = r.band(x, s + 1 + PREF);
    r.step(x, s);
    C16_STAMP(s, 1);
    if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PREF) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}
}  // namespace c16s

// EIGHT waves: every stage's three units as {0, 1} | {2} on two waves (with one stage per wave - four waves, two workgroups per CU - the CU ran
// two waves per SIMD and ~9 cycles per instruction: no faster than the tile form)
__global__ __launch_bounds__(512) void c2f16_stream_kernel(const C16Params p) {
  using namespace c16s;
  using G = Geo<false>;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.N * p.parts * p.strips;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, total) : (int)blockIdx.x;
  Ctx x;
  x.p = &p; x.sm = sm; x.lane = tid & 63; x.oy0 = x.ox0 = 0;
  x.n = bid / (p.parts * p.strips);
  bid -= x.n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  const int py0 = part * p.L;
  x.ry0 = py0 - 2; x.cx0 = strip * G::SW - 2;
  int leff = p.H - py0 < p.L ? p.H - py0 : p.L;
  leff = (leff + 1) & ~1;
  x.LP = leff + 4;
  x.S = leff / 2 + 5;  // steps until the last output row has left (cv2 runs 8 rows behind cv1)
  switch (wave) {
    case 0: { Cv1<false, 2, 1, true> r; run_dma(x, r); break; }
    case 1: { Conv3<false, 0, 0, 2> r; run(x, r); break; }
    case 2: { Conv3<false, 1, 0, 2> r; run(x, r); break; }
    case 3: { Cv2<false, 0, 2> r; run(x, r); break; }
    case 4: { Cv1<false, 0, 2, false> r; run(x, r); break; }
    case 5: { Conv3<false, 0, 2, 1> r; run(x, r); break; }
    case 6: { Conv3<false, 1, 2, 1> r; run(x, r); break; }
    default: { Cv2<false, 2, 1> r; run(x, r); break; }
  }
}

// The block AND the Conv(32, 64, 3, 2) behind it (yolov8n rows 2-3): TWELVE waves - the eight above (cv1 on four units) + one per n-tile of the
// stride-2 conv, a row's ten pixels per step (68 registers, 69.5 KB of LDS: two workgroups per CU.  An n-tile PAIR per wave - 10 waves - needs 72
// weight registers in those waves: at the 96 that two workgroups per CU allow it spills, 85-117 us against 64-68); the block's 160 x 160 x 32
// output (52 MB written and read back at batch 32, and a chip-filling launch of its own for the stride-2 conv) only ever exists as eight rows of a
// strip in LDS.
__global__ __launch_bounds__(768, 6) void c2f16_down_kernel(const C16Params p) {
  using namespace c16s;
  using G = Geo<true>;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.N * p.parts * p.strips;
  int bid = p.xcd ? upa_xcd_tile((int)blockIdx.x, total) : (int)blockIdx.x;
  Ctx x;
  x.p = &p; x.sm = sm; x.lane = tid & 63;
  x.n = bid / (p.parts * p.strips);
  bid -= x.n * (p.parts * p.strips);
  const int part = bid / p.strips, strip = bid - part * p.strips;
  x.oy0 = part * p.L; x.ox0 = strip * G::SW;       // L = stride-2 output rows per workgroup
  x.ry0 = 2 * x.oy0 - 3; x.cx0 = 2 * x.ox0 - 3;
  const int ld = p.OH - x.oy0 < p.L ? p.OH - x.oy0 : p.L;
  x.LP = 2 * ld + 5;        // y rows: the 2 ld + 1 rows of the block's output the stride-2 rows read, + 2 on either side
  x.S = ld + 7;             // the stride-2 row d leaves at step d + 7
  switch (wave) {
    case 0: { Cv1<true, 2, 2, true> r; run_dma(x, r); break; }
    case 1: { Conv3<true, 0, 0, 2> r; run(x, r); break; }
    case 2: { Conv3<true, 1, 0, 2> r; run(x, r); break; }
    case 3: { Cv2<true, 0, 2> r; run(x, r); break; }
    case 4: { Cv1<true, 0, 2, false> r; run(x, r); break; }
    case 5: { Conv3<true, 0, 2, 1> r; run(x, r); break; }
    case 6: { Conv3<true, 1, 2, 1> r; run(x, r); break; }
    case 7: { Cv2<true, 2, 1> r; run(x, r); break; }
    case 8: { Down<0, 1> r; run(x, r); break; }
    case 9: { Down<1, 1> r; run(x, r); break; }
    case 10: { Down<2, 1> r; run(x, r); break; }
    default: { Down<3, 1> r; run(x, r); break; }
  }
}

static int c16s_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

// Called by upa_c2f_fused (c2f_fused.hip) for the C2f(32, 32, n = 1, shortcut) form: UPA_EUNSUPPORTED = the caller runs the tile form.
int upa_c2f16_stream_launch(const void* x, int n, int h, int w, int ldx, const void* w1, const float* b1, const void* wa, const float* ba,
                            const void* wb, const float* bb, const void* w2, const float* b2, void* y, int ldy, const upa_opts* opts,
                            hipStream_t s) {
  if ((long)n * h * w * (long)(ldx > ldy ? ldx : ldy) * 2 >= (1L << 31) || (long)w * ldx * 2 >= (1L << 24) || h < 2 || w < 2) return UPA_EUNSUPPORTED;
  C16Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y;
  p.w1 = (const char*)w1; p.wa = (const char*)wa; p.wb = (const char*)wb; p.w2 = (const char*)w2;
  p.b1 = b1; p.ba = ba; p.bb = bb; p.b2 = b2;
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy;
  p.strips = cdiv(w, c16s::Geo<false>::SW);
  // rows per workgroup: the grid aims at one round of three workgroups per CU; `c2f_stream_rows`: even >= 4 (0 / -1: auto)
  const int rows = UPA_OPT(opts, c2f_stream_rows);
  UPA_CHECK_ARG(rows <= 0 || rows >= 4, "c2f_stream_rows = %d: 0 (auto), -1 (whole height) or >= 4", rows);
  int L = (h + 1) & ~1;
  if (rows >= 4) L = (rows + 1) & ~1;
  else {  // (-1, "the whole height", is the in-flight choice for the 16-wave kernels of c2f_stream.hip: their workgroups own a CU.  Three of these share one, so the one-round grid is the right size in flight as well)
    const long slots = 3L * c16s_cus();  // three workgroups fit a CU (49.5 KB of LDS, 8 waves each)
    long best = -1;
    for (int parts = 1; parts <= cdiv(h, 8); ++parts) {
      const int l = cdiv(cdiv(h, parts), 2) * 2;
      const long wgs = (long)n * p.strips * cdiv(h, l);
      const long cost = ((wgs + slots - 1) / slots) * (l / 2 + 5);
      if (best < 0 || cost < best) { best = cost; L = l; }
    }
  }
  if (L > ((h + 1) & ~1)) L = (h + 1) & ~1;
  p.L = L;
  p.parts = cdiv(h, L);
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long wgs = (long)n * p.strips * p.parts;
  if (wgs >= (1L << 31) / 2) return UPA_EUNSUPPORTED;
  if (upa_full_lds<c2f16_stream_kernel>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL(c2f16_stream_kernel, dim3((unsigned)wgs), dim3(512), c16s::Geo<false>::LDS, s, p);
  return UPA_OK;
}

// C2f(32, 32, n = 1, shortcut) AND the Conv(32, 64, 3, 2) that follows it as ONE launch (yolov8n rows 2-3: cfg/models/v8/yolov8.yaml, block.py:457-488,
// conv.py:188-197): x (n, h, w, 32) -> y (n, ceil(h / 2), ceil(w / 2), 64).  UPA_EUNSUPPORTED outside the form (callers run upa_c2f_fused, then
// upa_conv2d_bias_act).
extern "C" int upa_c2f16_down_fused(const void* x, int n, int h, int w, int ldx, const void* w1, const float* b1, const void* const* wm,
                                    const float* const* bm, const void* w2, const float* b2, const void* wd, const float* bd, void* y, int ldy,
                                    int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && y && w1 && b1 && wm && bm && wm[0] && wm[1] && bm[0] && bm[1] && w2 && b2 && wd && bd && n > 0 && h > 0 && w > 0,
                "c2f16_down_fused: bad args");
  const int oh = (h + 1) / 2, ow = (w + 1) / 2;
  if (UPA_OPT(opts, c2f) == 1 || UPA_OPT(opts, c2f) == 2 || UPA_OPT(opts, c2f_stream) == 1 || UPA_OPT(opts, c2f16_waves) != 0 || UPA_OPT(opts, no_c2f16_down) || dtype != UPA_BF16 ||
      ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 || h < 2 || w < 2 ||
      (long)n * h * w * (long)ldx * 2 >= (1L << 31) || (long)n * oh * ow * (long)ldy * 2 >= (1L << 31) || (long)w * ldx * 2 >= (1L << 24)) {
    upa_set_error("c2f16_down_fused: outside the fused form (bf16; C2f(32, 32, n=1, shortcut) + Conv(32, 64, 3, 2))");
    return UPA_EUNSUPPORTED;
  }
  C16Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y;
  p.w1 = (const char*)w1; p.wa = (const char*)wm[0]; p.wb = (const char*)wm[1]; p.w2 = (const char*)w2; p.wd = (const char*)wd;
  p.b1 = b1; p.ba = bm[0]; p.bb = bm[1]; p.b2 = b2; p.bd = bd;
  p.N = n; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.OH = oh; p.OW = ow;
  p.strips = cdiv(ow, c16s::Geo<true>::SW);
  // stride-2 rows per workgroup: one round of two workgroups per CU (69.5 KB of LDS, 12 waves each); `c2f_stream_rows` >= 4: that many INPUT rows
  const int rows = UPA_OPT(opts, c2f_stream_rows);
  UPA_CHECK_ARG(rows <= 0 || rows >= 4, "c2f_stream_rows = %d: 0 (auto), -1 (whole height) or >= 4", rows);
  int L = oh;
  if (rows >= 4) L = rows / 2;
  else {
    const long slots = 2L * c16s_cus();
    long best = -1;
    for (int parts = 1; parts <= cdiv(oh, 4); ++parts) {
      const int l = cdiv(oh, parts);
      const long wgs = (long)n * p.strips * cdiv(oh, l);
      const long cost = ((wgs + slots - 1) / slots) * (l + 7);
      if (best < 0 || cost < best) { best = cost; L = l; }
    }
  }
  if (L > oh) L = oh;
  if (L < 1) L = 1;
  p.L = L;
  p.parts = cdiv(oh, L);
  p.xcd = UPA_OPT(opts, no_xcd) ? 0 : 1;
  const long wgs = (long)n * p.strips * p.parts;
  if (wgs >= (1L << 31) / 2) return UPA_EUNSUPPORTED;
  if (upa_full_lds<c2f16_down_kernel>() != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL(c2f16_down_kernel, dim3((unsigned)wgs), dim3(768), c16s::Geo<true>::LDS, (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
