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
  int N, H, W, ldx, ldy, strips, parts, L, xcd;
};

// profiling build (-DUPA_STAMP): every wave of workgroups 0-3 records s_memtime at the start of each step and before its barrier
#ifdef UPA_STAMP
#define C16_STAMP_STEPS 64
__device__ unsigned long long g_c16_stamps[4 * 8 * C16_STAMP_STEPS * 2];
extern "C" int upa_debug_stamps_c2f16s(unsigned long long* out, int count) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c16_stamps), (size_t)count * 8) == hipSuccess ? 0 : -1;
}
#define C16_STAMP(step, which)                                                                               \
  do {                                                                                                       \
    if (blockIdx.x < 4 && (step) < C16_STAMP_STEPS) {                                                        \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      if ((threadIdx.x & 63) == 0) g_c16_stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * C16_STAMP_STEPS + (step)) * 2 + (which)] = t_; \
    }                                                                                                        \
  } while (0)
#else
#define C16_STAMP(step, which) do {} while (0)
#endif

namespace c16s {
constexpr int WS = 20;             // output columns of a strip
constexpr int XW = WS + 4;         // columns of x / y
constexpr int PREF = 2;            // input bands in flight ahead of the one cv1 reads (three workgroups per CU hide the rest of the round trip)
constexpr int XSLOTS = 32, XROWB = XSLOTS * 16, XROWS = 8, XPLANE = XROWS * XROWB;    // x ring: 4 planes, PREF + 2 bands alive
constexpr int YROWB = 24 * 16, YROWS = 16, YPLANE = YROWS * YROWB;                     // y ring: 4 planes (y0 | y1), rows 2s - 8 .. 2s + 1 alive
constexpr int TROWS = 8, TPLANE = TROWS * YROWB;                                       // t ring: 2 planes
constexpr int BROWS = 4, BPLANE = BROWS * YROWB;                                       // b ring: 2 planes
constexpr int XB = 0, YB = XB + 4 * XPLANE, TB = YB + 4 * YPLANE, BB = TB + 2 * TPLANE, DUMMY = BB + 2 * BPLANE, LDS = DUMMY + 512;
static_assert(XPLANE % 256 == 0 && YPLANE % 256 == 0 && TPLANE % 256 == 0 && BPLANE % 256 == 0, "planes keep the ds_read_b128 lane groups on disjoint banks");
// step s: cv1 -> y rows {2s, 2s + 1};  t rows {2s - 3, 2s - 2};  b rows {2s - 6, 2s - 5};  out rows {2s - 8, 2s - 7}
// (row coordinate i = image row py0 - 2 + i; x / y rows [0, LP), t rows [1, LP - 1), b / out rows [2, LP - 2))

struct Ctx {
  const C16Params* p;
  char* sm;
  int lane, n, py0, sx0, LP, S;
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

// ---- wave 0: the input bands by LDS-DMA (PREF ahead) and cv1 on the band that has landed.  Band b = x rows {2b, 2b + 1}; one instruction = one
// 8-channel plane of a band (2 rows x 32 slots, 24 used); lane = slot (row lane >> 5, column lane & 31)
template <int U0, int NU, bool DMA>
struct Cv1 {
  u32x4 w[2];
  f32x4 bias[2];
  const char* ximg;
  unsigned rowpitch, coloff;
  bool colok;
  int u_rr[NU], u_in[NU], u_out[NU];
  unsigned u_colm[NU];
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
    const int xc = lane & 31, gx = x.sx0 - 2 + xc;
    colok = xc < XW && gx >= 0 && gx < p.W;
    coloff = colok ? (unsigned)gx * (unsigned)p.ldx * 2u : 0u;
#pragma unroll
    for (int u = 0; u < NU; ++u) {  // the band's 48 pixels = three 16-pixel units exactly; this wave: units [U0, U0 + NU)
      const int q = 16 * (U0 + u) + r;
      u_rr[u] = q >= XW ? 1 : 0;
      const int col = q - u_rr[u] * XW;
      u_in[u] = XB + g * XPLANE + u_rr[u] * XROWB + col * 16;
      u_out[u] = YB + (g >> 1) * YPLANE + col * 16 + (g & 1) * 8;   // n-tile nt: + 2 nt planes
      const int gxx = x.sx0 - 2 + col;
      u_colm[u] = (gxx >= 0 && gxx < p.W) ? 0xFFFFFFFFu : 0u;
    }
  }
  __device__ __forceinline__ int band(const Ctx& x, int b) {
    if (2 * b >= x.LP) return 0;  // wave-uniform
    const int gy = x.py0 - 2 + 2 * b + (x.lane >> 5);
    const bool ok = colok && gy >= 0 && gy < x.p->H;
    const char* src = ok ? ximg + ((unsigned)gy * rowpitch + coloff) : reinterpret_cast<const char*>(g_c16_zero_page);
    const int dst = XB + ((2 * b) & (XROWS - 1)) * XROWB;  // + lane * 16 by the hardware
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
      __builtin_amdgcn_global_load_lds((c16_gptr_t)(src + (ok ? cg * 16 : 0)), (c16_lptr_t)(x.sm + dst + cg * XPLANE), 16, 0, 0);
    return 4;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    char* sm = x.sm;
    if (2 * s < x.LP) {
      const int r0 = 2 * s;
      u32x4 bx[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) bx[u] = lds128(sm, u_in[u] + (r0 & (XROWS - 1)) * XROWB);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const f32x4 a0 = mfma32(w[0], bx[u], bias[0]), a1 = mfma32(w[1], bx[u], bias[1]);
        const int row = r0 + u_rr[u];
        const int gy = x.py0 - 2 + row;
        const unsigned m = (gy >= 0 && gy < x.p->H) ? u_colm[u] : 0u;  // y is ZERO outside the image (the 3x3's padding)
        const int oa = u_out[u] + (row & (YROWS - 1)) * YROWB;
        *reinterpret_cast<u32x2*>(sm + oa) = silu_pack(a0, m);
        *reinterpret_cast<u32x2*>(sm + oa + 2 * YPLANE) = silu_pack(a1, m);
      }
    }
  }
};

// ---- waves 1, 2: a 3x3 stage with 16 input and 16 output channels.  STG 0: t from y1 (y planes 2, 3; 22 columns, rows {2s - 3, 2s - 2});
// STG 1: b = y1 + SiLU(conv(t)) (20 columns, rows {2s - 6, 2s - 5}).  k-step ks pairs taps 2 ks (lane groups 0-1) and 2 ks + 1 (groups 2-3).
template <int STG, int U0, int NU>
struct Conv3 {
  static constexpr int SD = STG ? WS : WS + 2, LAG = STG ? 6 : 3, LO = STG ? 2 : 1;
  static constexpr int IN_B = STG ? TB : YB + 2 * YPLANE, IN_PLANE = STG ? TPLANE : YPLANE, IN_MASK = (STG ? TROWS : YROWS) - 1;
  static constexpr int OUT_B = STG ? BB : TB, OUT_PLANE = STG ? BPLANE : TPLANE, OUT_MASK = (STG ? BROWS : TROWS) - 1;
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
      u_col[u] = qq - u_rr[u] * SD;           // output column; tap (dy, dx) reads input column col + dx (both stages)
      const int gx = x.sx0 - 1 + u_col[u];    // (only t lies outside the strip's own columns)
      u_colm[u] = (STG || (gx >= 0 && gx < p.W)) ? 0xFFFFFFFFu : 0u;
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
        b[u][ks] = lds128(sm, lane_in + ((r0 + u_rr[u] + tdy[ks] - 1) & IN_MASK) * YROWB + u_col[u] * 16 + tdx16[ks]);
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
      const int gy = x.py0 - 2 + row;
      const bool ok = u_act[u] && row >= LO && row < x.LP - LO;
      const int oa = OUT_B + (g >> 1) * OUT_PLANE + (row & OUT_MASK) * YROWB + u_col[u] * 16 + (g & 1) * 8;
      u32x2 o;
      if constexpr (STG == 0) {
        const unsigned m = (gy >= 0 && gy < x.p->H) ? u_colm[u] : 0u;  // t is ZERO outside the image (m.cv2's padding)
        o = silu_pack(acc[u], m);
      } else {
        // the shortcut: y1 (y planes 2, 3) at the same pixel = y column col + 2; f32 add, then the bf16 rounding of the separate launches
        const u32x2 rs = lds64(sm, YB + (2 + (g >> 1)) * YPLANE + (row & (YROWS - 1)) * YROWB + (u_col[u] + 2) * 16 + (g & 1) * 8);
        const float v0 = silu(acc[u][0]) + __uint_as_float(rs[0] << 16), v1 = silu(acc[u][1]) + __uint_as_float(rs[0] & 0xFFFF0000u);
        const float v2 = silu(acc[u][2]) + __uint_as_float(rs[1] << 16), v3 = silu(acc[u][3]) + __uint_as_float(rs[1] & 0xFFFF0000u);
        o = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
      }
      *reinterpret_cast<u32x2*>(sm + (ok ? oa : DUMMY + x.lane * 8)) = o;
    }
  }
};

// ---- wave 3: cv2 over [y0 | y1 | b] on the output band (rows {2s - 8, 2s - 7}, 20 columns = 40 pixels: three units), stores
template <int U0, int NU>
struct Cv2 {
  u32x4 w32[2];
  u32x2 w16[2];
  f32x4 bias[2];
  int u_rr[NU], u_col[NU], g, r;
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
      u_act[u] = q < 2 * WS;
      const int qq = u_act[u] ? q : 0;
      u_rr[u] = qq >= WS ? 1 : 0;
      u_col[u] = qq - u_rr[u] * WS;
      u_act[u] = u_act[u] && x.sx0 + u_col[u] < p.W;
    }
    yrow = (size_t)p.W * p.ldy * 2;
    // after the n-tile pairing (v_permlane16_swap) lane (g, r) holds channels 16 (g & 1) + 8 (g >> 1) .. + 7 of its pixel
    ybase = p.y + ((size_t)x.n * p.H * p.W + x.sx0) * (size_t)p.ldy * 2 + (16 * (g & 1) + 8 * (g >> 1)) * 2;
  }
  __device__ __forceinline__ void step(const Ctx& x, int s) {
    const int r0 = 2 * s - 8;
    if (!(r0 + 2 > 2 && r0 < x.LP - 2)) return;  // wave-uniform
    const char* sm = x.sm;
    u32x4 by[NU];
    u32x2 bb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int row = r0 + u_rr[u];
      by[u] = lds128(sm, YB + g * YPLANE + (row & (YROWS - 1)) * YROWB + (u_col[u] + 2) * 16);
      bb[u] = lds64(sm, BB + (g >> 1) * BPLANE + (row & (BROWS - 1)) * YROWB + u_col[u] * 16 + (g & 1) * 8);
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
      const int gy = x.py0 - 2 + row;
      const u32x2 a = silu_pack(o[u][0], 0xFFFFFFFFu), b = silu_pack(o[u][1], 0xFFFFFFFFu);
      auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
      if (u_act[u] && row >= 2 && row < x.LP - 2 && gy < x.p->H)
        *reinterpret_cast<u32x4*>(ybase + (size_t)gy * yrow + (size_t)u_col[u] * x.p->ldy * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    }
  }
};
}  // namespace c16s

template <typename R>
__device__ __forceinline__ void c16_run(const c16s::Ctx& x, R& r) {
  r.init(x);
  __syncthreads();
  for (int s = 0; s < x.S; ++s) { C16_STAMP(s, 0); r.step(x, s); C16_STAMP(s, 1); __syncthreads(); }
}

// EIGHT waves: every stage's three units as {0, 1} | {2} on two waves (with one stage per wave - four waves, two workgroups per CU - the CU ran
// two waves per SIMD and ~9 cycles per instruction: 60 us, no faster than the tile form)
__global__ __launch_bounds__(512) void c2f16_stream_kernel(const C16Params p) {
  using namespace c16s;
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
  x.py0 = part * p.L; x.sx0 = strip * WS;
  int leff = p.H - x.py0 < p.L ? p.H - x.py0 : p.L;
  leff = (leff + 1) & ~1;
  x.LP = leff + 4;
  x.S = leff / 2 + 5;  // steps until the last output row has left (cv2 runs 8 rows behind cv1)
  switch (wave) {
    case 0: {  // cv1 unit 2 + the input bands
      Cv1<2, 1, true> r;
      r.init(x);
      for (int b = 0; b <= PREF; ++b) r.band(x, b);
      // band 0 has landed once everything but the PREF younger bands is back (this wave issues no other vector-memory operation)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PREF) : "memory");
      if (2 * PREF >= x.LP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (short parts issue fewer bands than that)
      __syncthreads();
      for (int s = 0; s < x.S; ++s) {
        C16_STAMP(s, 0);
        const int issued = r.band(x, s + 1 + PREF);
        r.step(x, s);
        C16_STAMP(s, 1);
        // band s + 1 has landed once at most PREF bands (the youngest) are still in flight; near the end of the part fewer are issued at all
        if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PREF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      break;
    }
    case 1: { Conv3<0, 0, 2> r; c16_run(x, r); break; }
    case 2: { Conv3<1, 0, 2> r; c16_run(x, r); break; }
    case 3: { Cv2<0, 2> r; c16_run(x, r); break; }
    case 4: { Cv1<0, 2, false> r; c16_run(x, r); break; }
    case 5: { Conv3<0, 2, 1> r; c16_run(x, r); break; }
    case 6: { Conv3<1, 2, 1> r; c16_run(x, r); break; }
    default: { Cv2<2, 1> r; c16_run(x, r); break; }
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
  p.strips = cdiv(w, c16s::WS);
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
  hipLaunchKernelGGL(c2f16_stream_kernel, dim3((unsigned)wgs), dim3(512), c16s::LDS, s, p);
  return UPA_OK;
}
