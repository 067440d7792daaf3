// MHSA core of the BoT3 block: per (image, head) dense self-attention over the L = H*W pixels of a feature map,
//   energy[p][q] = sum_i Q[i][p] K[i][q]   (NOT scaled by 1/sqrt(d)),  attn = softmax_q(energy),
//   out[i][p]    = sum_q V[i][q] attn[p][q]                                  (+ residual, BottleneckTransformer)
// The same kernel is the core of nn.MultiheadAttention in the RT-DETR decoder (scale = 1/sqrt(d), 300 queries,
// transformer.py:670-673): there "pixels" are query tokens.
// Replaces torch.matmul / Softmax / matmul of MHSA.forward (ultralytics/nn/modules/block.py:6036-6062) and the
// `x + ...` of BottleneckTransformer.forward (:6090-6091).  q/k/v come from three 1x1 convs written into one NHWC
// buffer (channels [q | k | v]), so a (pixel, head) row is d contiguous elements.
//
// L = 400, d = 32 on the hot path (yolov5n-BoT3 @640): K and V of one (image, head) fit LDS (f32: 100 KiB, bf16: 50 KiB),
// are staged once per workgroup and broadcast-read by all lanes; one lane owns one query row and keeps a running
// (max, sum, out[d]) online softmax in registers, so the L x L energy matrix never exists in memory.
#include "common.h"

// Work split: a workgroup = 64 queries x KP key partitions (KP waves): wave `part` walks keys [part*L/KP, (part+1)*L/KP) for
// the 64 queries (one per lane) with the online softmax above; the KP partial states (max, sum, out[D]) are merged through
// LDS by wave 0 (exact: out = sum_p out_p * exp(m_p - m) / sum_p l_p * exp(m_p - m)).  With one lane walking all L keys
// (the first form) the RT-DETR self-attention (300 queries, 128 (image, head) pairs) ran 202 us on 384 two-wave workgroups.
template <typename T, int D, int KP>
__global__ __launch_bounds__(64 * KP) void mhsa_kernel(const char* q, const char* k, const char* v, int ld, int L, int heads,
                                                      const char* res, int ldr, char* y, int ldy, float scale) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  T* ks = reinterpret_cast<T*>(sm);  // [L][D]
  T* vs = ks + (size_t)L * D;        // [L][D]
  float* part_sm = reinterpret_cast<float*>(vs + (size_t)L * D);  // [KP][D + 2][64] partial states (lane-major: conflict free)
  constexpr int NT = 64 * KP;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const size_t pix0 = (size_t)b * L;
  constexpr int E = 16 / sizeof(T);
  if constexpr (D % E == 0) {
    constexpr int G = D / E;  // 16-byte groups per row
    for (int i = threadIdx.x; i < L * G; i += NT) {
      const int p = i / G, gq = i % G;
      const size_t off = ((pix0 + p) * ld + h * D + gq * E) * sizeof(T);
      reinterpret_cast<u32x4*>(ks)[i] = *reinterpret_cast<const u32x4*>(k + off);
      reinterpret_cast<u32x4*>(vs)[i] = *reinterpret_cast<const u32x4*>(v + off);
    }
  } else {  // tiny heads (unit-test sized): element-wise staging
    for (int i = threadIdx.x; i < L * D; i += NT) {
      const int p = i / D, e = i % D;
      const size_t off = ((pix0 + p) * ld + h * D + e) * sizeof(T);
      ks[i] = *reinterpret_cast<const T*>(k + off);
      vs[i] = *reinterpret_cast<const T*>(v + off);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int p = blockIdx.y * 64 + lane;
  const bool valid = p < L;
  float qr[D], o[D];
  {
    const T* qp = reinterpret_cast<const T*>(q + ((pix0 + (valid ? p : 0)) * ld + h * D) * sizeof(T));
#pragma unroll
    for (int i = 0; i < D; ++i) {
      qr[i] = ElemTraits<T>::load(qp + i) * scale;  // nn.MultiheadAttention scales q; BoT3's MHSA passes 1.0
      o[i] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const int per = (L + KP - 1) / KP;
  const int j0 = part * per, j1 = min(L, j0 + per);
  for (int j = j0; j < j1; ++j) {
    const T* kr = ks + (size_t)j * D;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) s = fmaf(qr[i], ElemTraits<T>::load(kr + i), s);
    const float mn = fmaxf(m, s);
    const float alpha = expf(m - mn);  // exp(-inf) = 0 on the first key
    const float pj = expf(s - mn);
    l = l * alpha + pj;
    const T* vr = vs + (size_t)j * D;
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] = fmaf(pj, ElemTraits<T>::load(vr + i), o[i] * alpha);
    m = mn;
  }
  if constexpr (KP > 1) {
    float* mine = part_sm + (size_t)part * (D + 2) * 64;
    mine[lane] = m;
    mine[64 + lane] = l;
#pragma unroll
    for (int i = 0; i < D; ++i) mine[(2 + i) * 64 + lane] = o[i];
    __syncthreads();
    if (part != 0) return;
    // merge in partition order (deterministic); a partition without keys carries m = -inf, l = 0
    float mm = m;
#pragma unroll
    for (int q_ = 1; q_ < KP; ++q_) mm = fmaxf(mm, part_sm[(size_t)q_ * (D + 2) * 64 + lane]);
    const float a0 = expf(m - mm);
    l *= a0;
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] *= a0;
#pragma unroll
    for (int q_ = 1; q_ < KP; ++q_) {
      const float* ot = part_sm + (size_t)q_ * (D + 2) * 64;
      const float aq = expf(ot[lane] - mm);
      l += ot[64 + lane] * aq;
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] = fmaf(ot[(2 + i) * 64 + lane], aq, o[i]);
    }
  }
  if (!valid) return;
  const float inv = 1.0f / l;
  T* yp = reinterpret_cast<T*>(y + ((pix0 + p) * ldy + h * D) * sizeof(T));
  const T* rp = res ? reinterpret_cast<const T*>(res + ((pix0 + p) * ldr + h * D) * sizeof(T)) : nullptr;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float val = o[i] * inv;
    if (rp) val += ElemTraits<T>::load(rp + i);
    if constexpr (sizeof(T) == 4) yp[i] = val;
    else yp[i] = f32_to_bf16(val);
  }
}

template <typename T, int D>
static int launch_mhsa(const void* q, const void* k, const void* v, int ld, int n, int hw, int heads, const void* residual,
                       int ldr, void* y, int ldy, float scale, hipStream_t s) {
  constexpr int KP = 4;
  const size_t lds = (size_t)2 * hw * D * sizeof(T) + (size_t)KP * (D + 2) * 64 * sizeof(float);
  if (lds > 158 * 1024) {
    upa_set_error("mhsa: %d keys x %d dims do not fit LDS", hw, D);
    return UPA_EUNSUPPORTED;
  }
  dim3 grid((unsigned)(n * heads), (unsigned)cdiv(hw, 64));
  auto kern = mhsa_kernel<T, D, KP>;
  (void)upa_full_lds<mhsa_kernel<T, D, KP>>();
  hipLaunchKernelGGL(kern, grid, dim3(64 * KP), lds, s, (const char*)q, (const char*)k, (const char*)v, ld, hw, heads,
                     (const char*)residual, ldr, (char*)y, ldy, scale);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_mhsa(const void* q, const void* k, const void* v, int ldqkv, int n, int hw, int heads, int d,
                        float scale, const void* residual, int ldr, void* y, int ldy, int dtype, void* stream) {
  UPA_CHECK_ARG(q && k && v && y, "mhsa: null pointer");
  const int es = upa_elem_size(dtype);
  UPA_CHECK_ARG(ldqkv % (16 / es) == 0 && ldy % (16 / es) == 0, "mhsa: strides must be multiples of 16 bytes");
  hipStream_t s = (hipStream_t)stream;
#define UPA_MHSA_CASE(DD)                                                                                         \
  if (d == DD)                                                                                                    \
    return dtype == UPA_BF16 ? launch_mhsa<bf16_t, DD>(q, k, v, ldqkv, n, hw, heads, residual, ldr, y, ldy, scale, s)     \
                             : launch_mhsa<float, DD>(q, k, v, ldqkv, n, hw, heads, residual, ldr, y, ldy, scale, s);
  UPA_MHSA_CASE(4)
  UPA_MHSA_CASE(8)
  UPA_MHSA_CASE(16)
  UPA_MHSA_CASE(32)
  UPA_MHSA_CASE(64)
#undef UPA_MHSA_CASE
  upa_set_error("mhsa: head dim %d not supported (4, 8, 16, 32, 64)", d);
  return UPA_EUNSUPPORTED;
}
