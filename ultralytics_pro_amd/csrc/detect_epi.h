// Detect decode as a convolution EPILOGUE (bf16 perf mode): the last 1x1 conv of a Detect branch leaves its logits in MFMA
// accumulators - D[row = channel][col = pixel], lane (kg = lane >> 4, p16 = lane & 15) holds channels 16j + 4kg + q
// (q = 0..3) of pixel p16 for n-tile j - and this header turns them into the reference's decoded output rows without the
// (B, H, W, 4*reg_max + nc) round trip through HBM that a separate decode kernel needs:
//   * box branch (4 n-tiles = 4 sides x reg_max 16 bins): DFL softmax expectation over the 16 bins of a side
//     (nn/modules/block.py:250-253) = 4 values in the lane x 4 lane rows, reduced with v_permlane16_swap / v_permlane32_swap
//     (no LDS); dist2bbox around the cell-centre anchor and x stride (utils/tal.py:352-376, nn/modules/head.py:151-169,
//     184-191); lane row kg writes output channel kg (x, y, w, h) of its pixel;
//   * class branch: sigmoid of every logit (head.py:169), written channel-major.
// Output layout = the reference's (B, 4 + nc, A) float32; along the anchor axis a 16-lane row writes 64 contiguous bytes.
// Arithmetic as csrc/detect.hip in bf16 mode (v_exp_f32 / v_rcp_f32); the f32 parity mode keeps the separate kernel.
#pragma once
#include "common.h"

struct DetectEpi {
  float* y;        // (B, 4 + nc, a_total) f32
  int a_total;     // anchors of all levels
  int a0;          // first anchor of this level
  int HW, W;       // pixels per image / row width of this level
  unsigned magicHW;  // floor(2^32 / HW) + 1: pix / HW == umulhi(pix, magicHW) while pix * (magicHW * HW - 2^32) < 2^32 (host-checked)
  unsigned magicW;
  int nc;
  float stride_px;
  // NMS prefilter (class branch only; nullptr: off): the NMS sort key of every anchor's best class,
  // (~bits(best score) << 32) | (anchor * nc + best class), best class = FIRST maximum (torch.max order, nms.py:109), as a dense
  // (B, a_total) array - upa_nms_batched_hot takes its candidates from it instead of re-reading the (B, nc, A) score block
  unsigned long long* best_keys;
  // 1 = the class rows of y are NOT written (upa_opts.keys_only, needs best_keys): single-label non_max_suppression reads only the
  // boxes and the best-class keys (upa_nms_batched_hot), so the 4 * nc * A bytes per image of scores and all but one sigmoid per
  // anchor are dead work on the predict path.  The keys are bit-identical to the ones computed next to the full score rows.
  int keys_only;
};

__device__ __forceinline__ float upa_row_sum4(float v) {  // sum over the 4 lane rows (lanes l, l^16, l^32, l^48)
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float upa_row_max4(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Box branch of one 16-pixel tile.  v[j] = logits (bias added) of side j: bins 4kg + q of pixel `pix`.
// (b, a) = image and level-local anchor of the lane's pixel (upa_detect_split, or straight from the tile coordinates).
__device__ __forceinline__ void upa_detect_box_store(const DetectEpi& d, const f32x4 (&v)[4], int b, int a, bool ok, int kg) {
  constexpr float LOG2E = 1.44269504088896340736f;
  float dist[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float m = upa_row_max4(fmaxf(fmaxf(v[s][0], v[s][1]), fmaxf(v[s][2], v[s][3])));
    float sum = 0.f, e = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ex = __builtin_amdgcn_exp2f((v[s][q] - m) * LOG2E);
      sum += ex;
      e += ex * (float)(4 * kg + q);
    }
    sum = upa_row_sum4(sum);
    e = upa_row_sum4(e);
    dist[s] = e * __builtin_amdgcn_rcpf(sum);
  }
  const int ay = (int)__umulhi((unsigned)a, d.magicW);
  const int ax = a - ay * d.W;
  const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f;
  const float x1 = cx - dist[0], y1 = cy - dist[1], x2 = cx + dist[2], y2 = cy + dist[3];
  const float o = kg == 0 ? (x1 + x2) / 2.f : kg == 1 ? (y1 + y2) / 2.f : kg == 2 ? (x2 - x1) : (y2 - y1);
  if (ok) d.y[((size_t)b * (4 + d.nc) + kg) * d.a_total + d.a0 + a] = o * d.stride_px;
}

__device__ __forceinline__ int upa_row_min4i(int v) {  // min over the 4 lane rows
  auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  const int s = min((int)a[0], (int)a[1]);
  auto b = __builtin_amdgcn_permlane32_swap((unsigned)s, (unsigned)s, false, false);
  return min((int)b[0], (int)b[1]);
}

// Class branch: n-tile j of one 16-pixel tile; v = logits (bias added) of classes 16j + 4kg + q.  (best, bc) = running first
// maximum over this lane's classes (call with ascending j: within a lane the classes ascend with (j, q)).
__device__ __forceinline__ void upa_detect_cls_store(const DetectEpi& d, const f32x4& v, int j, int b, int a, bool ok, int kg,
                                                     float& best, int& bc) {
  constexpr float LOG2E = 1.44269504088896340736f;
  float* yb = d.y + ((size_t)b * (4 + d.nc) + 4) * d.a_total + d.a0 + a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 16 * j + 4 * kg + q;
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[q] * -LOG2E));
    if (ok && c < d.nc) {
      if (!d.keys_only) yb[(size_t)c * d.a_total] = sg;
      if (sg > best) { best = sg; bc = c; }
    }
  }
}

// keys_only form of a whole pixel tile: logits[j] = n-tile j (bias added), NTC n-tiles.  The score of the best class is the sigmoid
// of the largest LOGIT when the sigmoid (v_exp_f32 + v_rcp_f32, each within ~1 ulp) separates it from the runner-up for certain:
// largest logit <= 4 (slope of the sigmoid >= 0.017: no saturation) and a gap of >= 1e-4 to the second largest logit of the PIXEL
// (relative score gap >= 1.8e-6, an order of magnitude above the two approximations' error) - then the first maximum of the scores
// (torch.max order, nms.py:109) is that class and ONE sigmoid per lane replaces NTC * 4.  Any other pixel in the wave (uniform
// test) sends the wave through the full per-class sigmoids exactly as the score-writing form does.  Returns (best, bc) of the lane.
template <int NTC>
__device__ __forceinline__ void upa_detect_cls_keys_only(const DetectEpi& d, const f32x4 (&logit)[NTC], bool ok, int kg, float& best, int& bc) {
  constexpr float LOG2E = 1.44269504088896340736f;
  float m1 = -INFINITY, m2 = -INFINITY;  // the lane's largest and second largest logit (valid classes only)
  int c1 = 0;
#pragma unroll
  for (int j = 0; j < NTC; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 16 * j + 4 * kg + q;
      const float v = c < d.nc ? logit[j][q] : -INFINITY;
      if (v > m1) { m2 = m1; m1 = v; c1 = c; }
      else m2 = fmaxf(m2, v);
    }
  const float pm = upa_row_max4(m1);                            // the pixel's largest logit
  const float ps = upa_row_max4(m1 == pm ? m2 : m1);            // ... and a lower bound of its second largest (exact unless two rows tie at pm)
  const bool sure = !ok || (pm <= 4.0f && pm - ps >= 1e-4f && upa_row_sum4(m1 == pm ? 1.f : 0.f) == 1.f);
  if (__all(sure)) {
    best = ok && m1 == pm ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m1 * -LOG2E)) : -1.f;
    bc = c1;
    return;
  }
  best = -1.f;
  bc = 0;
#pragma unroll
  for (int j = 0; j < NTC; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 16 * j + 4 * kg + q;
      const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(logit[j][q] * -LOG2E));
      if (ok && c < d.nc && sg > best) { best = sg; bc = c; }
    }
}

// After the class rows of one 16-pixel tile: (best, bc) = this lane's first maximum (lane (kg, p16): classes 4kg + q of every
// n-tile of pixel p16).  The pixel's first maximum over all classes = the row maximum of the scores and, among the lane rows
// that hold it, the smallest class index; lane row 0 stores the anchor's NMS key (no atomics, no counters: a dense array).
__device__ __forceinline__ void upa_detect_best_key_store(const DetectEpi& d, float best, int bc, int b, int al, bool ok, int lane) {
  const float m = upa_row_max4(best);
  const int c = upa_row_min4i(best == m ? bc : 0x7FFFFFFF);
  if (lane < 16 && ok) {
    const int a = d.a0 + al;
    d.best_keys[(size_t)b * d.a_total + a] = ((unsigned long long)(~__float_as_uint(m)) << 32) | (unsigned)(a * d.nc + c);
  }
}

// flat pixel index (n, h, w flattened) -> (image, level-local anchor).  Exact for every pix the launch can produce: the host
// refuses shapes outside upa_magic_exact(n * h * w - 1, h * w).
__device__ __forceinline__ void upa_detect_split(const DetectEpi& d, int pix, int& b, int& a) {
  b = (int)__umulhi((unsigned)pix, d.magicHW);
  a = pix - b * d.HW;
}

static inline unsigned upa_magic_div(int d) { return (unsigned)(0x100000000ULL / (unsigned long long)d) + 1u; }
// umulhi(x, upa_magic_div(d)) == x / d for every 0 <= x <= xmax  <=>  xmax * (magic * d - 2^32) < 2^32
static inline bool upa_magic_exact(long xmax, int d) {
  if (xmax < 0 || d <= 0 || xmax > 0xFFFFFFFFL) return false;
  const unsigned long long e = (unsigned long long)upa_magic_div(d) * (unsigned long long)d - 0x100000000ULL;
  return (unsigned long long)xmax * e < 0x100000000ULL;
}
