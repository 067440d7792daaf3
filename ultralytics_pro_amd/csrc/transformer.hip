// Row-wise f32 kernels of the RT-DETR decoder head (ultralytics/nn/modules/head.py:1905-2224, transformer.py:438-773).
// The GEMMs (nn.Linear) run on the MFMA conv kernel as 1x1 convolutions over token rows (upa_linear below); this file
// holds what is left: LayerNorm (+ residual), query selection top-k, row gather / mask, the multi-scale deformable
// bilinear sampling, and the small box arithmetic.  All HBM/latency-bound; one wave or one lane per row.
#include "common.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------------------------------
// y = LayerNorm(x (+ r)) * gamma + beta, one wave per row (nn.LayerNorm, transformer.py:660-685; head.py:1998)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* x, const float* r, const float* gamma,
                                                         const float* beta, float eps, float* y, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  const float* rr = r ? r + (size_t)row * C : nullptr;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
    v += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = 1.0f / sqrtf(v / (float)C + eps);
  float* yr = y + (size_t)row * C;
  for (int c = lane; c < C; c += 64) yr[c] = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int upa_layer_norm(const float* x, const float* residual, int m, int c, const float* gamma, const float* beta,
                              float eps, float* y, void* stream) {
  UPA_CHECK_ARG(x && gamma && beta && y && m > 0 && c > 0, "layer_norm: bad args");
  hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, gamma,
                     beta, eps, y, m, c);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row utilities: y = a + b ; y = x * mask[row] ; y[i] = x[idx[i]]
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_kernel(int mode, const float* a, const float* b, const int* idx, float* y,
                                                   long M, int C) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    if (mode == 0) y[i] = a[i] + b[i];
    else if (mode == 1) y[i] = a[i] * b[row];
    else y[i] = a[(size_t)idx[row] * C + c];
  }
}

static unsigned rows_grid(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int upa_rows_add(const float* a, const float* b, float* y, long m, int c, void* stream) {
  UPA_CHECK_ARG(a && b && y, "rows_add: null pointer");
  hipLaunchKernelGGL(rows_kernel, dim3(rows_grid(m * c)), dim3(256), 0, (hipStream_t)stream, 0, a, b, nullptr, y, m, c);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
extern "C" int upa_rows_scale(const float* x, const float* row_scale, float* y, long m, int c, void* stream) {
  UPA_CHECK_ARG(x && row_scale && y, "rows_scale: null pointer");
  hipLaunchKernelGGL(rows_kernel, dim3(rows_grid(m * c)), dim3(256), 0, (hipStream_t)stream, 1, x, row_scale, nullptr, y, m,
                     c);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
extern "C" int upa_rows_gather(const float* x, const int32_t* row_idx, float* y, long m_out, int c, void* stream) {
  UPA_CHECK_ARG(x && row_idx && y, "rows_gather: null pointer");
  hipLaunchKernelGGL(rows_kernel, dim3(rows_grid(m_out * c)), dim3(256), 0, (hipStream_t)stream, 2, x, nullptr, row_idx, y,
                     m_out, c);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Query selection: per image, top-k tokens by max class logit (head.py:2175).  Tokens are stored level-major
// (row = lvl_row0[l] + b*lvl_hw[l] + p); the reference's token index t = lvl_tok0[l] + p orders ties (lower t first).
// One 1024-thread workgroup per image: row max -> 64-bit keys -> LDS bitonic sort -> first k.
// ---------------------------------------------------------------------------------------------------------------------
struct LevelTable {
  int n_levels;
  int hw[4];
  int row0[4];  // first row of the level in the level-major token matrix
  int tok0[4];  // first reference token index of the level
};

// One workgroup per image.  (1) key of every token = (descending best-class score, token index), four lanes per token with
// 16-byte loads when nc % 4 == 0; (2) the K smallest keys: radix select of the K-th score (four 8-bit passes over the LDS
// keys), the tokens at or above it compacted and bitonic-sorted (<= TOPK_CAP of them; ties included, the index half of the
// key orders them); more candidates than that (a flat score map) take the full sort of all tokens.  (The first form sorted
// all 16384 padded keys - 105 barrier-separated passes - and read the class rows one float at a time: 357 us per call.)
constexpr int TOPK_CAP = 2048;

__global__ __launch_bounds__(1024) void topk_tokens_kernel(const float* scores, int nc, LevelTable lt, int B, int T, int K,
                                                           int npad, int32_t* out_rows, int32_t* out_tok) {
  extern __shared__ __attribute__((aligned(16))) u64 keys[];   // [max(npad, T + TOPK_CAP)]
  __shared__ int hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  __shared__ int s_count;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const bool vec = (nc & 3) == 0 && ((uintptr_t)scores & 15) == 0;
  if (vec) {
    const int sub = tid & 3, nq = nc >> 2;
    for (int t0 = 0; t0 < T; t0 += 256) {
      const int t = t0 + (tid >> 2);
      float m = -INFINITY;
      if (t < T) {
        int l = 0;
        while (l + 1 < lt.n_levels && t >= lt.tok0[l + 1]) ++l;
        const size_t row = (size_t)lt.row0[l] + (size_t)b * lt.hw[l] + (t - lt.tok0[l]);
        const f32x4* s4 = reinterpret_cast<const f32x4*>(scores + row * nc);
        for (int q = sub; q < nq; q += 4) {
          const f32x4 v = s4[q];
          m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        }
      }
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 2));
      if (t < T && sub == 0) {
        unsigned u = __float_as_uint(m);   // order-preserving float -> uint (handles negatives), descending
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        keys[t] = ((u64)(~u) << 32) | (unsigned)t;
      }
    }
  } else {
    for (int t = tid; t < T; t += 1024) {
      int l = 0;
      while (l + 1 < lt.n_levels && t >= lt.tok0[l + 1]) ++l;
      const size_t row = (size_t)lt.row0[l] + (size_t)b * lt.hw[l] + (t - lt.tok0[l]);
      const float* s = scores + row * nc;
      float m = s[0];
      for (int c = 1; c < nc; ++c) m = fmaxf(m, s[c]);
      unsigned u = __float_as_uint(m);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      keys[t] = ((u64)(~u) << 32) | (unsigned)t;
    }
  }
  if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)K; s_count = 0; }
  __syncthreads();
  // ---- the K-th smallest score word (the high half of the keys)
  unsigned mask = 0u;
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int t = tid; t < T; t += 1024) {
      const unsigned hi = (unsigned)(keys[t] >> 32);
      if ((hi & mask) == prefix) atomicAdd(&hist[(hi >> (8 * pass)) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned rem = s_remaining, acc = 0u;
      int bin = 0;
      for (; bin < 255; ++bin) {
        const unsigned h = (unsigned)hist[bin];
        if (acc + h >= rem) break;
        acc += h;
      }
      s_remaining = rem - acc;
      s_prefix = prefix | ((unsigned)bin << (8 * pass));
    }
    mask |= 255u << (8 * pass);
    __syncthreads();
  }
  const unsigned thr = s_prefix;
  // ---- candidates: every token whose score word is <= the K-th one (ties included)
  u64* cand = keys + T;
  for (int t = tid; t < T; t += 1024) {
    const u64 k = keys[t];
    if ((unsigned)(k >> 32) <= thr) {
      const int slot = atomicAdd(&s_count, 1);
      if (slot < TOPK_CAP) cand[slot] = k;
    }
  }
  __syncthreads();
  const int ncand = s_count;
  u64* sorted = cand;
  if (ncand <= TOPK_CAP) {
    int n2 = 2;
    while (n2 < ncand) n2 <<= 1;
    for (int t = ncand + tid; t < n2; t += 1024) cand[t] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (n2 >> 1); t += 1024) {
          const int i = 2 * t - (t & (j - 1));
          const int l = i + j;
          const bool up = (i & k) == 0;
          const u64 x = cand[i], y = cand[l];
          if ((x > y) == up) { cand[i] = y; cand[l] = x; }
        }
        __syncthreads();
      }
    }
  } else {   // a flat score map: sort everything
    sorted = keys;
    for (int t = T + tid; t < npad; t += 1024) keys[t] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (npad >> 1); t += 1024) {
          const int i = 2 * t - (t & (j - 1));
          const int l = i + j;
          const bool up = (i & k) == 0;
          const u64 x = keys[i], y = keys[l];
          if ((x > y) == up) { keys[i] = y; keys[l] = x; }
        }
        __syncthreads();
      }
    }
  }
  for (int q = tid; q < K; q += 1024) {
    const int t = (int)(sorted[q] & 0xFFFFFFFFull);
    int l = 0;
    while (l + 1 < lt.n_levels && t >= lt.tok0[l + 1]) ++l;
    out_rows[(size_t)b * K + q] = lt.row0[l] + b * lt.hw[l] + (t - lt.tok0[l]);
    out_tok[(size_t)b * K + q] = t;
  }
}

extern "C" int upa_topk_tokens(const float* scores, int nc, int n_levels, const int32_t* level_hw, int b, int k,
                               int32_t* out_rows, int32_t* out_tok, void* stream) {
  UPA_CHECK_ARG(scores && level_hw && out_rows && out_tok && n_levels >= 1 && n_levels <= 4, "topk_tokens: bad args");
  LevelTable lt;
  lt.n_levels = n_levels;
  int tok = 0, row = 0;
  for (int l = 0; l < n_levels; ++l) {
    lt.hw[l] = level_hw[l];
    lt.tok0[l] = tok;
    lt.row0[l] = row;
    tok += level_hw[l];
    row += level_hw[l] * b;
  }
  for (int l = n_levels; l < 4; ++l) lt.hw[l] = lt.tok0[l] = lt.row0[l] = 0;
  const int T = tok;
  UPA_CHECK_ARG(k <= T, "topk_tokens: k > tokens");
  int npad = 2;
  while (npad < T) npad <<= 1;
  const size_t lds = (size_t)(npad > T + TOPK_CAP ? npad : T + TOPK_CAP) * 8;
  UPA_CHECK_ARG(lds <= 150 * 1024, "topk_tokens: %d tokens do not fit LDS", T);
  (void)upa_full_lds<topk_tokens_kernel>();
  hipLaunchKernelGGL(topk_tokens_kernel, dim3((unsigned)b), dim3(1024), lds, (hipStream_t)stream, scores, nc, lt, b, T, k,
                     npad, out_rows, out_tok);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Box arithmetic.  mode 0: y = sigmoid(d + inverse_sigmoid(ref))   (transformer.py:756-757, nn/modules/utils.py:79-100)
//                  mode 1: y = d + anchors[tok]  (logit-space reference boxes, head.py:2183)   mode 2: y = sigmoid(x)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float inv_sigmoid_f(float x, float eps) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
  return logf(x1 / x2);
}

__global__ __launch_bounds__(256) void box_kernel(int mode, const float* d, const float* ref, const int32_t* tok,
                                                  const float* anchors, float* y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) y[i] = sigmoid_f(d[i] + inv_sigmoid_f(ref[i], 1e-5f));
  else if (mode == 1) y[i] = d[i] + anchors[(size_t)tok[i >> 2] * 4 + (i & 3)];
  else y[i] = sigmoid_f(d[i]);
}

extern "C" int upa_box_refine(const float* delta, const float* ref, float* y, long n_boxes, void* stream) {
  UPA_CHECK_ARG(delta && ref && y, "box_refine: null pointer");
  hipLaunchKernelGGL(box_kernel, dim3((unsigned)cdiv((int)(n_boxes * 4), 256)), dim3(256), 0, (hipStream_t)stream, 0, delta,
                     ref, nullptr, nullptr, y, n_boxes * 4);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
extern "C" int upa_box_add_anchors(const float* delta, const int32_t* tok, const float* anchors, float* y, long n_boxes,
                                   void* stream) {
  UPA_CHECK_ARG(delta && tok && anchors && y, "box_add_anchors: null pointer");
  hipLaunchKernelGGL(box_kernel, dim3((unsigned)cdiv((int)(n_boxes * 4), 256)), dim3(256), 0, (hipStream_t)stream, 1, delta,
                     nullptr, tok, anchors, y, n_boxes * 4);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
extern "C" int upa_sigmoid(const float* x, float* y, long n, void* stream) {
  UPA_CHECK_ARG(x && y, "sigmoid: null pointer");
  hipLaunchKernelGGL(box_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, 2, x, nullptr, nullptr,
                     nullptr, y, n);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// y[m, 0:4] = boxes[m], y[m, 4:4+nc] = sigmoid(scores[m])        (head.py:2074)
__global__ __launch_bounds__(256) void rtdetr_output_kernel(const float* boxes, const float* scores, float* y, long M,
                                                            int nc) {
  const int no = 4 + nc;
  const long total = M * no;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / no;
    const int c = (int)(i - m * no);
    y[i] = c < 4 ? boxes[m * 4 + c] : sigmoid_f(scores[m * nc + (c - 4)]);
  }
}
extern "C" int upa_rtdetr_output(const float* boxes, const float* scores, float* y, long m, int nc, void* stream) {
  UPA_CHECK_ARG(boxes && scores && y, "rtdetr_output: null pointer");
  hipLaunchKernelGGL(rtdetr_output_kernel, dim3(rows_grid(m * (4 + nc))), dim3(256), 0, (hipStream_t)stream, boxes, scores,
                     y, m, nc);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// RTDETRPredictor.postprocess (ultralytics/models/rtdetr/predict.py:35-74), one workgroup per image, one thread per
// query: xywh2xyxy (ops.py:268-284: xy -/+ wh / 2), best class (first maximum, as torch.max), score > conf (and the
// optional class filter), descending score order (ties: lower query first - the oracle's stable argsort), [:max_det],
// x * ow, y * oh.  Fixed-shape outputs (out (B, max_det, 6), counts (B,)) so the step stays capturable; every float
// operation is the reference's, in its order, with FP contraction off: bit-exact.
__global__ __launch_bounds__(1024) void rtdetr_postprocess_kernel(const float* preds, int Q, int nc, float conf,
                                                                  const unsigned char* classes_mask, int max_det,
                                                                  const float* orig_wh, float ow, float oh, float* out,
                                                                  int* counts) {
  extern __shared__ float sc[];  // [Q] score of a valid query, -1 for a filtered one (scores are sigmoids: > 0)
  const int b = blockIdx.x, q = threadIdx.x;
  float box[4] = {0.f, 0.f, 0.f, 0.f}, best = -1.f;
  int cls = 0;
  bool valid = false;
  if (q < Q) {
    const float* p = preds + ((size_t)b * Q + q) * (4 + nc);
    const float hw = p[2] / 2, hh = p[3] / 2;
    box[0] = p[0] - hw; box[1] = p[1] - hh; box[2] = p[0] + hw; box[3] = p[1] + hh;
    best = p[4];
    for (int c = 1; c < nc; ++c)
      if (p[4 + c] > best) { best = p[4 + c]; cls = c; }
    valid = best > conf && (!classes_mask || classes_mask[cls]);
    sc[q] = valid ? best : -1.f;
  }
  __syncthreads();
  if (orig_wh) { ow = orig_wh[2 * b]; oh = orig_wh[2 * b + 1]; }
  int rank = 0, nvalid = 0;
  for (int j = 0; j < Q; ++j) {
    const float s = sc[j];
    nvalid += s >= 0.f;
    rank += (s > best) || (s == best && j < q);
  }
  if (q == 0) counts[b] = nvalid < max_det ? nvalid : max_det;
  if (valid && rank < max_det) {
    float* o = out + ((size_t)b * max_det + rank) * 6;
    o[0] = box[0] * ow; o[1] = box[1] * oh; o[2] = box[2] * ow; o[3] = box[3] * oh;
    o[4] = best; o[5] = (float)cls;
  }
}
extern "C" int upa_rtdetr_postprocess(const float* preds, int b, int q, int nc, float conf, const unsigned char* classes_mask,
                                      int max_det, const float* orig_wh, float ow, float oh, float* out, int32_t* counts,
                                      void* stream) {
  UPA_CHECK_ARG(preds && out && counts && b > 0 && nc > 0 && max_det > 0, "rtdetr_postprocess: bad args");
  UPA_CHECK_ARG(q > 0 && q <= 1024, "rtdetr_postprocess: 1..1024 queries per image (head.py:1905 uses 300), got %d", q);
  const int threads = (q + 63) / 64 * 64;
  hipLaunchKernelGGL(rtdetr_postprocess_kernel, dim3((unsigned)b), dim3(threads), (size_t)q * sizeof(float), (hipStream_t)stream,
                     preds, q, nc, conf, classes_mask, max_det, orig_wh, ow, oh, out, (int*)counts);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-scale deformable attention sampling (nn/modules/utils.py:103-159 + transformer.py:540-556, 4-d reference boxes):
//   for (b, q, head): w = softmax over (levels x points) of the attention logits;
//   loc = ref_xy + off / n_points * ref_wh * 0.5 ; bilinear sample (align_corners=False, zero padding) of the value map
//   of each level; out[b,q,head,:] = sum_{l,p} w * sample.
// Half a wave (32 lanes = the 32 channels of one head) per (b, q, head): every corner fetch is one 128-byte line.
// value rows are level-major: row(l, b, y, x) = row0[l] + (b*H_l + y)*W_l + x, width = heads*D.
// ---------------------------------------------------------------------------------------------------------------------
struct DeformLevels {
  int n_levels;
  int h[4], w[4], row0[4];
};

// VT = element type of the projected values: float rows, or bf16 rows (unsigned short) when the value projections of all
// decoder layers were computed as one bf16 GEMM (perf mode); ldv = row stride of `value` in elements (>= heads * D).
template <int D, int NP, typename VT>
__global__ __launch_bounds__(256) void msdeform_kernel(const VT* value, int ldv, DeformLevels lv, int B, int LQ, int heads,
                                                       const float* offsets, const float* logits, const float* ref,
                                                       float* y) {
  auto ldval = [](const VT* q) __attribute__((always_inline)) -> float {
    if constexpr (sizeof(VT) == 2) return __uint_as_float(((unsigned)*q) << 16);
    else return *q;
  };
  static_assert(256 % D == 0, "D lanes per (b, q, head) item");
  const long item = (long)blockIdx.x * (256 / D) + (threadIdx.x / D);  // (b, q, head)
  const int ch = threadIdx.x % D;
  const long total = (long)B * LQ * heads;
  if (item >= total) return;
  const int head = (int)(item % heads);
  const long bq = item / heads;
  const int b = (int)(bq / LQ);
  const int NLP = lv.n_levels * NP;
  const float* lg = logits + bq * (heads * NLP) + head * NLP;
  const float* of = offsets + bq * (heads * NLP * 2) + head * NLP * 2;
  const float rx = ref[bq * 4 + 0], ry = ref[bq * 4 + 1], rw = ref[bq * 4 + 2], rh = ref[bq * 4 + 3];
  float mx = lg[0];
  for (int i = 1; i < NLP; ++i) mx = fmaxf(mx, lg[i]);
  const int C = heads * D;
  float acc = 0.f;
  if constexpr (sizeof(VT) == 2) {
    // perf mode (bf16 value rows): the softmax of the <= 16 sampling logits once per lane on v_exp_f32 / v_rcp_f32, and the four corner
    // fetches of a point UNCONDITIONAL (clamped address, weight zeroed outside the map) so that the 4 NP loads of a level are issued
    // together instead of one dependent branch each - the exact-f32 instantiation below keeps the reference's arithmetic order
    float e[16];
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      e[i] = i < NLP ? __expf(lg[i] - mx) : 0.f;
      den += e[i];
    }
    const float inv = __builtin_amdgcn_rcpf(den);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l >= lv.n_levels) break;  // uniform
      const int H = lv.h[l], W = lv.w[l];
      const VT* vbase = value + ((size_t)lv.row0[l] + (size_t)b * H * W) * (size_t)ldv + head * D + ch;
      float v[NP][4], wc[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = l * NP + p;
        const float lx = rx + of[2 * i] / (float)NP * rw * 0.5f;
        const float ly = ry + of[2 * i + 1] / (float)NP * rh * 0.5f;
        const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
        const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        // (clamped before the int conversion: a wild offset must not overflow it)
        const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)W), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)H);
        const float tx = ix - x0f, ty = iy - y0f;
        const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
        const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
        const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
        const float wgt = e[i] * inv;
        wc[p][0] = (yin0 && xin0) ? wgt * (1.f - tx) * (1.f - ty) : 0.f;
        wc[p][1] = (yin0 && xin1) ? wgt * tx * (1.f - ty) : 0.f;
        wc[p][2] = (yin1 && xin0) ? wgt * (1.f - tx) * ty : 0.f;
        wc[p][3] = (yin1 && xin1) ? wgt * tx * ty : 0.f;
        v[p][0] = ldval(vbase + ((size_t)ya * W + xa) * ldv);
        v[p][1] = ldval(vbase + ((size_t)ya * W + xb) * ldv);
        v[p][2] = ldval(vbase + ((size_t)yb * W + xa) * ldv);
        v[p][3] = ldval(vbase + ((size_t)yb * W + xb) * ldv);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) acc += wc[p][c4] * v[p][c4];
    }
  } else {
  float den = 0.f;
  for (int i = 0; i < NLP; ++i) den += expf(lg[i] - mx);
  for (int l = 0; l < lv.n_levels; ++l) {
    const int H = lv.h[l], W = lv.w[l];
    const VT* vbase = value + ((size_t)lv.row0[l] + (size_t)b * H * W) * (size_t)ldv + head * D + ch;
    for (int p = 0; p < NP; ++p) {
      const int i = l * NP + p;
      const float wgt = expf(lg[i] - mx) / den;
      const float lx = rx + of[2 * i] / (float)NP * rw * 0.5f;
      const float ly = ry + of[2 * i + 1] / (float)NP * rh * 0.5f;
      // grid_sample(align_corners=False): grid = 2*loc-1 -> pixel = ((grid+1)*size-1)/2
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
      const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
      const float x0f = floorf(ix), y0f = floorf(iy);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float tx = ix - x0f, ty = iy - y0f;
      float s = 0.f;
      const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
      const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
      if (yin0 && xin0) s += ldval(vbase + ((size_t)y0 * W + x0) * ldv) * (1.f - tx) * (1.f - ty);
      if (yin0 && xin1) s += ldval(vbase + ((size_t)y0 * W + x0 + 1) * ldv) * tx * (1.f - ty);
      if (yin1 && xin0) s += ldval(vbase + ((size_t)(y0 + 1) * W + x0) * ldv) * (1.f - tx) * ty;
      if (yin1 && xin1) s += ldval(vbase + ((size_t)(y0 + 1) * W + x0 + 1) * ldv) * tx * ty;
      acc += wgt * s;
    }
  }
  }
  y[bq * C + head * D + ch] = acc;
}

extern "C" int upa_msdeform_attn_strided(const void* value, int value_dtype, int ldv, const int32_t* shapes_hw, int n_levels,
                                         int b, int heads, int d, const float* offsets, const float* attn_logits,
                                         const float* ref_boxes, int len_q, int n_points, float* y, void* stream) {
  UPA_CHECK_ARG(value && shapes_hw && offsets && attn_logits && ref_boxes && y, "msdeform_attn: null pointer");
  UPA_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && (d == 32 || d == 8) && n_points == 4,
                "msdeform_attn: supports <=4 levels, head dim 32 (or 8), 4 points (head.py:1951-1961)");
  UPA_CHECK_ARG((value_dtype == UPA_F32 || value_dtype == UPA_BF16) && ldv >= heads * d, "msdeform_attn: bad value dtype / stride");
  DeformLevels lv;
  lv.n_levels = n_levels;
  int row = 0;
  for (int l = 0; l < 4; ++l) {
    lv.h[l] = l < n_levels ? shapes_hw[2 * l] : 0;
    lv.w[l] = l < n_levels ? shapes_hw[2 * l + 1] : 0;
    lv.row0[l] = row;
    row += lv.h[l] * lv.w[l] * b;
  }
  const long items = (long)b * len_q * heads;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g32((unsigned)((items + 7) / 8)), g8((unsigned)((items + 31) / 32));
  if (value_dtype == UPA_F32) {
    const float* v = (const float*)value;
    if (d == 32) hipLaunchKernelGGL((msdeform_kernel<32, 4, float>), g32, dim3(256), 0, s, v, ldv, lv, b, len_q, heads, offsets, attn_logits, ref_boxes, y);
    else hipLaunchKernelGGL((msdeform_kernel<8, 4, float>), g8, dim3(256), 0, s, v, ldv, lv, b, len_q, heads, offsets, attn_logits, ref_boxes, y);
  } else {
    const unsigned short* v = (const unsigned short*)value;
    if (d == 32) hipLaunchKernelGGL((msdeform_kernel<32, 4, unsigned short>), g32, dim3(256), 0, s, v, ldv, lv, b, len_q, heads, offsets, attn_logits, ref_boxes, y);
    else hipLaunchKernelGGL((msdeform_kernel<8, 4, unsigned short>), g8, dim3(256), 0, s, v, ldv, lv, b, len_q, heads, offsets, attn_logits, ref_boxes, y);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_msdeform_attn(const float* value, const int32_t* shapes_hw, int n_levels, int b, int heads, int d,
                                 const float* offsets, const float* attn_logits, const float* ref_boxes, int len_q,
                                 int n_points, float* y, void* stream) {
  return upa_msdeform_attn_strided(value, UPA_F32, heads * d, shapes_hw, n_levels, b, heads, d, offsets, attn_logits, ref_boxes,
                                   len_q, n_points, y, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.Linear on token rows = 1x1 convolution over a (1, 1, M, K) NHWC view on the f32 MFMA kernel (exact f32 chain).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int upa_conv2d_bias_act(const void*, int, int, int, int, int, const void*, const float*, void*, int, int,
                                   const void*, int, int, int, int, int, int, const upa_opts*, void*);

// ---- small-M linear: the RT-DETR decoder's 4800-row GEMMs (65 per step, K = 256 | 1024, N = 4 .. 1024).  On the conv kernel each
// is four 64-channel chunks of [halo DMA, barrier, one tap] - 34 us for 0.6 GFLOP, all latency.  Here a 4-wave workgroup stages its
// LR rows x ALL K of x in ONE LDS-DMA burst (16-byte groups XOR-swizzled by the row, as the conv halos), every wave keeps the A
// fragments of its n-tile for 256 channels of K in registers (the next 256 are fetched while these multiply), and the K loop runs
// without a barrier: 4 exact-f32 MFMAs (v_mfma_f32_16x16x4_f32) per 16-byte fragment pair, one accumulator chain over the whole K
// (K <= 1024: no partial-sum folding needed at f32 accuracy).  Epilogue from the accumulators: bias, ReLU / SiLU (precise), residual,
// 16-byte f32 stores.
typedef __attribute__((address_space(1))) const void* lgptr_t;
typedef __attribute__((address_space(3))) void* llptr_t;
__device__ __attribute__((aligned(16))) unsigned g_lin_zero16[4] = {0u, 0u, 0u, 0u};

struct LinParams {
  const char* x; const char* w; const float* bias; const char* res; char* y;
  int M, K, N, ldx, ldy, ldr, act, KTT, NTn;
};

template <int MT>  // m-tiles (16 rows each) per workgroup: 2 for K <= 512, 1 up to K = 1024 (64 KB of LDS either way)
__global__ __launch_bounds__(256) void linear_f32_kernel(const LinParams p) {
  constexpr int LR = MT * 16;
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int row0 = blockIdx.x * LR;
  const int nt = blockIdx.y * 4 + wave;  // this wave's n-tile (16 output columns)
  const int G = p.K >> 2;                // 16-byte groups per row
  // ---- stage the x tile: item (row, slot) <- group slot ^ (row & 7) of that row
  const int items = LR * G;
  for (int base = wave * 64; base < items; base += 256) {
    const int it = base + lane;
    const int row = it / G, slot = it - row * G;
    const int cg = slot ^ (row & 7);
    const char* src = reinterpret_cast<const char*>(g_lin_zero16);
    if (row0 + row < p.M) src = p.x + ((size_t)(row0 + row) * p.ldx + cg * 4) * 4;
    __builtin_amdgcn_global_load_lds((lgptr_t)src, (llptr_t)(lsm + base * 16), 16, 0, 0);
  }
  const bool live = nt < p.NTn;  // wave-uniform: column blocks past N only help staging
  // ---- A fragments of this n-tile: [k-tile][n-tile][lane][16 B]; 16 k-tiles (256 channels) per register set
  u32x4 a0[16], a1[16];
  auto fetch = [&](u32x4(&dst)[16], int kt0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      dst[q] = (live && kt0 + q < p.KTT) ? *reinterpret_cast<const u32x4*>(p.w + (((size_t)(kt0 + q) * p.NTn + nt) * 64 + lane) * 16)
                                         : u32x4{0u, 0u, 0u, 0u};
  };
  fetch(a0, 0);
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto mult = [&](const u32x4(&a)[16], int kt0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (kt0 + q >= p.KTT) break;  // uniform
      const float* af = reinterpret_cast<const float*>(&a[q]);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + r;
        const u32x4 b = *reinterpret_cast<const u32x4*>(lsm + ((size_t)row * G + ((((kt0 + q) << 2) + g) ^ (row & 7))) * 16);
        const float* bf = reinterpret_cast<const float*>(&b);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc[i], 0, 0, 0);
      }
    }
  };
  for (int kt0 = 0; kt0 < p.KTT; kt0 += 32) {
    if (kt0 + 16 < p.KTT) fetch(a1, kt0 + 16);
    mult(a0, kt0);
    if (kt0 + 16 < p.KTT) {
      if (kt0 + 32 < p.KTT) fetch(a0, kt0 + 32);
      mult(a1, kt0 + 16);
    }
  }
  if (!live) return;
  // ---- epilogue: lane (g, r) holds columns 16 nt + 4g .. + 3 of row r of m-tile i
  const int col = nt * 16 + 4 * g;
  if (col >= p.N) return;
  const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};  // (bias padded to 16 by the packer's caller)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = row0 + i * 16 + r;
    if (row >= p.M) continue;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[i][e] + bv[e];
      if (p.act == UPA_ACT_RELU) t = fmaxf(t, 0.0f);
      else if (p.act == UPA_ACT_SILU) t = t / (1.0f + expf(-t));
      v[e] = t;
    }
    if (p.res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + ((size_t)row * p.ldr + col) * 4);
      v += rv;
    }
    *reinterpret_cast<f32x4*>(p.y + ((size_t)row * p.ldy + col) * 4) = v;
  }
}

// The same GEMM with the PRODUCT on the bf16 matrix cores (perf mode of the RT-DETR decoder): x stays float32 in memory and in LDS (no
// layout or neighbour kernel changes), every lane rounds its 8 consecutive k of a row to bf16 on the way into the B fragment (two
// 16-byte reads + four v_cvt_pk_bf16_f32), the weights come packed as bf16 (upa_pack_conv_weight(UPA_BF16): k-tiles of 32), one
// v_mfma_f32_16x16x32_bf16 replaces eight exact-f32 MFMAs; accumulation, bias, activation, residual and the stored result are float32.
// This is the arithmetic of the reference's half-precision predict (`model.half()`, engine/predictor.py:151-173: every nn.Linear of the
// decoder multiplies 16-bit operands) with the activations kept in f32 between the layers.
// XB: x rows are bf16 already (the attention output feeding out_proj): 16-byte groups of 8 values, no conversion.  YB: the result is
// stored as bf16 rows (q, k, v for the matrix-core attention kernel).
template <int MT, bool XB, bool YB>
__global__ __launch_bounds__(256) void linear_bf16_kernel(const LinParams p) {
  constexpr int LR = MT * 16;
  constexpr int GE = XB ? 8 : 4;  // elements per 16-byte group
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int row0 = blockIdx.x * LR;
  const int nt = blockIdx.y * 4 + wave;
  const int G = p.K / GE;  // 16-byte groups per row
  const int items = LR * G;
  for (int base = wave * 64; base < items; base += 256) {
    const int it = base + lane;
    const int row = it / G, slot = it - row * G;
    const int cg = slot ^ (row & 7);
    const char* src = reinterpret_cast<const char*>(g_lin_zero16);
    if (row0 + row < p.M) src = p.x + ((size_t)(row0 + row) * p.ldx + cg * GE) * (XB ? 2 : 4);
    __builtin_amdgcn_global_load_lds((lgptr_t)src, (llptr_t)(lsm + base * 16), 16, 0, 0);
  }
  const bool live = nt < p.NTn;
  // A fragments of this n-tile: [k-tile of 32][n-tile][lane][16 B]; 8 k-tiles (256 channels) per register set
  u32x4 a0[8], a1[8];
  auto fetch = [&](u32x4(&dst)[8], int kt0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      dst[q] = (live && kt0 + q < p.KTT) ? *reinterpret_cast<const u32x4*>(p.w + (((size_t)(kt0 + q) * p.NTn + nt) * 64 + lane) * 16)
                                         : u32x4{0u, 0u, 0u, 0u};
  };
  fetch(a0, 0);
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto mult = [&](const u32x4(&a)[8], int kt0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (kt0 + q >= p.KTT) break;  // uniform
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + r;
        u32x4 b;
        if constexpr (XB) {
          b = *reinterpret_cast<const u32x4*>(lsm + ((size_t)row * G + ((((kt0 + q) << 2) + g) ^ (row & 7))) * 16);
        } else {
          const int g0 = ((kt0 + q) << 3) + 2 * g;  // this lane's 8 k = float groups g0, g0 + 1 of the row
          const f32x4 lo = *reinterpret_cast<const f32x4*>(lsm + ((size_t)row * G + (g0 ^ (row & 7))) * 16);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(lsm + ((size_t)row * G + ((g0 + 1) ^ (row & 7))) * 16);
          b = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[q]), *reinterpret_cast<const bf16x8*>(&b), acc[i], 0, 0, 0);
      }
    }
  };
  for (int kt0 = 0; kt0 < p.KTT; kt0 += 16) {
    if (kt0 + 8 < p.KTT) fetch(a1, kt0 + 8);
    mult(a0, kt0);
    if (kt0 + 8 < p.KTT) {
      if (kt0 + 16 < p.KTT) fetch(a0, kt0 + 16);
      mult(a1, kt0 + 8);
    }
  }
  if (!live) return;
  const int col = nt * 16 + 4 * g;
  if (col >= p.N) return;
  const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = row0 + i * 16 + r;
    if (row >= p.M) continue;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[i][e] + bv[e];
      if (p.act == UPA_ACT_RELU) t = fmaxf(t, 0.0f);
      else if (p.act == UPA_ACT_SILU) t = t / (1.0f + expf(-t));
      v[e] = t;
    }
    if (p.res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + ((size_t)row * p.ldr + col) * 4);
      v += rv;
    }
    if constexpr (YB) *reinterpret_cast<u32x2*>(p.y + ((size_t)row * p.ldy + col) * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    else *reinterpret_cast<f32x4*>(p.y + ((size_t)row * p.ldy + col) * 4) = v;
  }
}

/* y = act(x W^T + b) (+ residual) with the product on the bf16 matrix cores: w_packed = upa_pack_conv_weight(UPA_BF16) of the (n, k, 1, 1)
 * weight; x rows float32 (rounded to bf16 into the MFMA) or bf16, y rows float32 or bf16 (x_dtype / y_dtype = UPA_F32 | UPA_BF16; strides
 * in elements), residual float32.  UPA_EUNSUPPORTED outside the form (k % 32 - bf16 x: % 64 -, k <= 1024, n % 4, 16-byte rows; more than
 * 16384 rows only up to 1024 columns): the caller then uses upa_linear with float32-packed weights. */
extern "C" int upa_linear_mixed(const void* x, int x_dtype, long m, int k, int ldx, const void* w_packed, const float* bias, void* y,
                                int y_dtype, int n, int ldy, const float* residual, int ldr, int act, void* stream) {
  UPA_CHECK_ARG(x && w_packed && y && m > 0 && m < (1L << 31), "linear_mixed: bad args");
  UPA_CHECK_ARG((x_dtype == UPA_F32 || x_dtype == UPA_BF16) && (y_dtype == UPA_F32 || y_dtype == UPA_BF16), "linear_mixed: dtypes are f32 | bf16");
  const bool xb = x_dtype == UPA_BF16, yb = y_dtype == UPA_BF16;
  const bool ok = k % (xb ? 64 : 32) == 0 && k <= 1024 && n % 4 == 0 && ldx % (xb ? 8 : 4) == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0) &&
                  ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % (yb ? 8 : 16)) == 0 && (!residual || (uintptr_t)residual % 16 == 0) &&
                  (!bias || (uintptr_t)bias % 16 == 0) && m <= (1 << 20) && (m <= 16384 || n <= 1024) &&
                  (act == UPA_ACT_NONE || act == UPA_ACT_RELU || act == UPA_ACT_SILU);
  if (!ok) {
    upa_set_error("linear_mixed: outside the form (k %% 32 == 0 (bf16 x: %% 64), k <= 1024, n %% 4 == 0, 16-byte aligned rows)");
    return UPA_EUNSUPPORTED;
  }
  LinParams p;
  p.x = (const char*)x; p.w = (const char*)w_packed; p.bias = bias; p.res = (const char*)residual; p.y = (char*)y;
  p.M = (int)m; p.K = k; p.N = n; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.act = act;
  p.KTT = k / 32; p.NTn = (n + 15) / 16;
  const int mt = k <= 512 ? 2 : 1;
  const size_t lds = (size_t)mt * 16 * k * (xb ? 2 : 4);
  const dim3 grid((unsigned)((m + mt * 16 - 1) / (mt * 16)), (unsigned)((p.NTn + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
#define UPA_LIN_LAUNCH(MT_, XB_, YB_)                                                                  \
  do {                                                                                               \
    if (upa_full_lds<linear_bf16_kernel<MT_, XB_, YB_>>() != hipSuccess) return UPA_ELAUNCH;           \
    hipLaunchKernelGGL((linear_bf16_kernel<MT_, XB_, YB_>), grid, dim3(256), lds, s, p);               \
  } while (0)
  if (mt == 2) {
    if (xb) { if (yb) UPA_LIN_LAUNCH(2, true, true); else UPA_LIN_LAUNCH(2, true, false); }
    else { if (yb) UPA_LIN_LAUNCH(2, false, true); else UPA_LIN_LAUNCH(2, false, false); }
  } else {
    if (xb) { if (yb) UPA_LIN_LAUNCH(1, true, true); else UPA_LIN_LAUNCH(1, true, false); }
    else { if (yb) UPA_LIN_LAUNCH(1, false, true); else UPA_LIN_LAUNCH(1, false, false); }
  }
#undef UPA_LIN_LAUNCH
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
/* = upa_linear_mixed with float32 rows in and out. */
extern "C" int upa_linear_bf16(const float* x, long m, int k, int ldx, const void* w_packed, const float* bias, float* y, int n,
                               int ldy, const float* residual, int ldr, int act, void* stream) {
  return upa_linear_mixed(x, UPA_F32, m, k, ldx, w_packed, bias, y, UPA_F32, n, ldy, residual, ldr, act, stream);
}

extern "C" int upa_linear(const float* x, long m, int k, int ldx, const void* w_packed, const float* bias, float* y, int n,
                          int ldy, const float* residual, int ldr, int act, void* stream) {
  UPA_CHECK_ARG(m > 0 && m < (1L << 31), "linear: bad row count");
  // small-M form: whole K in LDS.  Needs 16-byte rows everywhere, K in whole 32-channel units (the row swizzle permutes 8 groups), whole
  // 4-column store groups.
  const bool small = x && w_packed && y && k % 32 == 0 && k <= 1024 && n % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 &&
                     (!residual || ldr % 4 == 0) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                     (!residual || (uintptr_t)residual % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0) && m <= (1 << 20) &&
                     // every 64-column block of the grid re-stages the workgroup's x rows: measured a win for the decoder's 4800-row
                     // GEMMs at any N and for the encoder's 134 k-row ones at N <= 256 (4 column blocks); beyond both - many rows AND
                     // many column blocks - x would be read N / 64 times, so those go to the conv kernel
                     (m <= 16384 || n <= 256) &&
                     (act == UPA_ACT_NONE || act == UPA_ACT_RELU || act == UPA_ACT_SILU);
  if (small) {
    LinParams p;
    p.x = (const char*)x; p.w = (const char*)w_packed; p.bias = bias; p.res = (const char*)residual; p.y = (char*)y;
    p.M = (int)m; p.K = k; p.N = n; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.act = act;
    p.KTT = k / 16; p.NTn = (n + 15) / 16;
    const int mt = k <= 512 ? 2 : 1;
    const size_t lds = (size_t)mt * 16 * k * 4;
    const dim3 grid((unsigned)((m + mt * 16 - 1) / (mt * 16)), (unsigned)((p.NTn + 3) / 4));
    if (mt == 2) {
      if (upa_full_lds<linear_f32_kernel<2>>() != hipSuccess) return UPA_ELAUNCH;
      hipLaunchKernelGGL(linear_f32_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
      if (upa_full_lds<linear_f32_kernel<1>>() != hipSuccess) return UPA_ELAUNCH;
      hipLaunchKernelGGL(linear_f32_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
  return upa_conv2d_bias_act(x, 1, 1, (int)m, k, ldx, w_packed, bias, y, n, ldy, residual, ldr, 1, 1, 0, act, UPA_F32, nullptr, stream);
}
