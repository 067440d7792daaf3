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

// =====================================================================================================================
// bf16, d = 32 (BoT3's MHSA on the hot path: 4 heads x 32 dims, 400 keys): the same attention on the matrix cores.
// One wave owns 16 queries and walks the keys in blocks of 32 with an online softmax; both GEMMs are transposed so that a lane
// keeps ONE query for the whole walk:
//   S^T (keys x queries)  = K . Q^T   A = 16 keys x 32 dims from the K image in LDS ([key][64 B], 16-byte groups XOR-swizzled by
//                                     key >> 1: conflict-free ds_read_b128), B = the wave's 16 queries (registers, loaded once);
//                                     D: lane (g, r) holds keys 4g .. 4g + 3 of the tile for query r: the softmax statistics of a
//                                     query live in its four lanes (two butterfly steps), none cross queries;
//   O^T (dims x queries) += V^T . P^T  B = P^T: the exponentials of two S^T tiles packed to bf16 ARE a B operand (k order: keys
//                                     4g + e of the first tile, then of the second - the conv_big tail trick), A = 16 dims x those 32
//                                     keys from a TRANSPOSED V image in LDS ([dim][key], pitch = 4 mod 8 dwords: conflict-free
//                                     ds_read_b64); D: lane (g, r): dims 4g .. 4g + 3 of query r, so the running rescale
//                                     exp2(m_old - m_new) of query r multiplies registers of the lane that computed it.
// 4 MFMAs per 32 keys x 16 queries instead of ~2 k FMAs + 2 accurate expf per (query, key) on the vector ALU: 176 us -> see DESIGN.md.
// exp via v_exp_f32 on log2(e)-scaled scores (the scale of nn.MultiheadAttention folds into the same multiply), P rounded to bf16
// (the values being averaged are bf16 already); f32 accumulation, f32 residual add, one rounding of the output as before.
// =====================================================================================================================
typedef __attribute__((address_space(1))) const void* agptr_t;
typedef __attribute__((address_space(3))) void* alptr_t;
__device__ __attribute__((aligned(16))) unsigned g_mhsa_zero16[4] = {0u, 0u, 0u, 0u};

__global__ __launch_bounds__(512) void mhsa_mfma_bf16_d32_kernel(const char* q, const char* k, const char* v, int ld, int L, int Lp, int VP,
                                                                 int heads, const char* res, int ldr, char* y, int ldy, float c_log2) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* ks = sm;                       // [Lp][64 B]
  char* vt = sm + (size_t)Lp * 64;     // [32][VP dwords]
  const int tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const size_t pix0 = (size_t)b * L;
  // ---- K by LDS-DMA (slot i of the image = 16-byte group (i & 3) ^ swz of key i >> 2; zero page past L)
  for (int base = wave * 64; base < Lp * 4; base += NT) {
    const int i = base + lane, key = i >> 2, cg = (i & 3) ^ ((key >> 1) & 3);
    const char* src = key < L ? k + ((pix0 + key) * (size_t)ld + h * 32 + cg * 8) * 2 : reinterpret_cast<const char*>(g_mhsa_zero16);
    __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)(ks + base * 16), 16, 0, 0);
  }
  // ---- V transposed: item = (key, 8 dims) -> eight 2-byte stores
  for (int i = tid; i < Lp * 4; i += NT) {
    const int key = i >> 2, dg = i & 3;
    u32x4 x = u32x4{0u, 0u, 0u, 0u};
    if (key < L) x = *reinterpret_cast<const u32x4*>(v + ((pix0 + key) * (size_t)ld + h * 32 + dg * 8) * 2);
    unsigned short* col = reinterpret_cast<unsigned short*>(vt) + key;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      col[(size_t)(dg * 8 + 2 * e) * VP * 2] = (unsigned short)(x[e] & 0xFFFFu);
      col[(size_t)(dg * 8 + 2 * e + 1) * VP * 2] = (unsigned short)(x[e] >> 16);
    }
  }
  // this wave's queries
  const int q0 = (blockIdx.y * (NT >> 6) + wave) * 16;
  const int qi = q0 + r;
  u32x4 qB = u32x4{0u, 0u, 0u, 0u};
  if (qi < L) qB = *reinterpret_cast<const u32x4*>(q + ((pix0 + qi) * (size_t)ld + h * 32 + g * 8) * 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (q0 >= L) return;  // (after the barrier: every wave takes part in the staging)

  float m = -INFINITY, l = 0.f;
  f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = o0;
  const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* vrow0 = vt + ((size_t)r * VP + 2 * g) * 4;          // dims r / 16 + r, keys 4g .. of a block
  const char* vrow1 = vt + ((size_t)(16 + r) * VP + 2 * g) * 4;
  for (int kb = 0; kb < Lp; kb += 32) {
    const int k0 = kb + r, k1 = kb + 16 + r;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(ks + k0 * 64 + ((g ^ ((k0 >> 1) & 3)) << 4));
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(ks + k1 * 64 + ((g ^ ((k1 >> 1) & 3)) << 4));
    f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a0), *reinterpret_cast<const bf16x8*>(&qB), z, 0, 0, 0);
    f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a1), *reinterpret_cast<const bf16x8*>(&qB), z, 0, 0, 0);
    // the V^T fragments of this block (independent of the softmax: issued early)
    const u32x2 v00 = *reinterpret_cast<const u32x2*>(vrow0 + kb * 2), v01 = *reinterpret_cast<const u32x2*>(vrow0 + kb * 2 + 32);
    const u32x2 v10 = *reinterpret_cast<const u32x2*>(vrow1 + kb * 2), v11 = *reinterpret_cast<const u32x2*>(vrow1 + kb * 2 + 32);
    float t[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[e] = s0[e] * c_log2;
      t[4 + e] = s1[e] * c_log2;
    }
    if (kb + 32 > L) {  // the last block: keys past L do not exist
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kb + 4 * g + e >= L) t[e] = -INFINITY;
        if (kb + 16 + 4 * g + e >= L) t[4 + e] = -INFINITY;
      }
    }
    float bm = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
    bm = fmaxf(bm, __shfl_xor(bm, 16));
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    const float mn = fmaxf(m, bm);  // finite: the first block always holds a key
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    float pe[8], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pe[e] = __builtin_amdgcn_exp2f(t[e] - mn);
      ps += pe[e];
    }
    l = l * alpha + ps;
    m = mn;
    o0 *= alpha;
    o1 *= alpha;
    const u32x4 pB = u32x4{pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]), pack_bf16x2(pe[4], pe[5]), pack_bf16x2(pe[6], pe[7])};
    const u32x4 av0 = u32x4{v00[0], v00[1], v01[0], v01[1]}, av1 = u32x4{v10[0], v10[1], v11[0], v11[1]};
    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&av0), *reinterpret_cast<const bf16x8*>(&pB), o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&av1), *reinterpret_cast<const bf16x8*>(&pB), o1, 0, 0, 0);
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (qi >= L) return;
  const float inv = 1.0f / l;
  const size_t row = pix0 + qi;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x4 o = j ? o1 : o0;
    float val[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) val[e] = o[e] * inv;
    const int d0 = h * 32 + 16 * j + 4 * g;
    if (res) {
      const u32x2 rv = *reinterpret_cast<const u32x2*>(res + (row * (size_t)ldr + d0) * 2);
      val[0] += __uint_as_float(rv[0] << 16); val[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
      val[2] += __uint_as_float(rv[1] << 16); val[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
    }
    *reinterpret_cast<u32x2*>(y + (row * (size_t)ldy + d0) * 2) = u32x2{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
  }
}

static int launch_mhsa_mfma(const void* q, const void* k, const void* v, int ld, int n, int hw, int heads, const void* residual, int ldr,
                            void* y, int ldy, float scale, hipStream_t s) {
  const int Lp = (hw + 31) & ~31;
  const int VP = Lp / 2 + 4;  // dwords per V^T row: = 4 mod 8 -> the 16 rows of a ds_read_b64 half hit 16 distinct bank quads
  const size_t lds = (size_t)Lp * 64 + (size_t)32 * VP * 4;
  if (lds > 158 * 1024) return UPA_EUNSUPPORTED;
  const int tiles = cdiv(hw, 16);
  const int chunks = cdiv(tiles, 8);              // query chunks per (image, head)
  const int waves = cdiv(tiles, chunks);          // 25 tiles -> 4 chunks x 7 waves
  (void)upa_full_lds<mhsa_mfma_bf16_d32_kernel>();
  hipLaunchKernelGGL(mhsa_mfma_bf16_d32_kernel, dim3((unsigned)(n * heads), (unsigned)chunks), dim3(64 * waves), lds, s, (const char*)q,
                     (const char*)k, (const char*)v, ld, hw, Lp, VP, heads, (const char*)residual, ldr, (char*)y, ldy,
                     scale * 1.4426950408889634f);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
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
  if (dtype == UPA_BF16 && d == 32 && hw >= 16 && ldr % 4 == 0) {  // the matrix-core form (8-byte residual / output groups)
    const int rc = launch_mhsa_mfma(q, k, v, ldqkv, n, hw, heads, residual, ldr, y, ldy, scale, s);
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
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
