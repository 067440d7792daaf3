// HBM-bound NHWC kernels: max pooling, the SPPF pooling chain, nearest 2x upsample into a channel slice, view copy/add
// and NCHW<->NHWC boundary conversion.  All move 16 bytes of channels per lane (8 bf16 / 4 f32), lanes run along the
// channel axis first so a wave touches whole contiguous pixel rows.
#include "common.h"

namespace {

template <typename T> struct Vec16;  // 16 bytes of T as float lanes
template <> struct Vec16<float> {
  static constexpr int E = 4;
  float v[4];
  __device__ static Vec16 load(const char* p) {
    Vec16 r;
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
    return r;
  }
  __device__ void store(char* p) const { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int E = 8;
  float v[8];
  __device__ static Vec16 load(const char* p) {
    Vec16 r;
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r.v[2 * i] = __uint_as_float(t[i] << 16);
      r.v[2 * i + 1] = __uint_as_float(t[i] & 0xFFFF0000u);
    }
    return r;
  }
  __device__ void store(char* p) const {
    *reinterpret_cast<u32x4*>(p) =
        u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  }
};

// ---- generic max pool -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const char* x, char* y, int N, int H, int W, int C, int ldx, int OH,
                                                      int OW, int ldy, int K, int S, int P, int padBR) {
  constexpr int E = Vec16<T>::E;
  const int CG = C / E;
  const long total = (long)N * OH * OW * CG;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cg = (int)(gid % CG);
    long pix = gid / CG;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    const int n = (int)(pix / ((long)OW * OH));
    Vec16<T> m;
#pragma unroll
    for (int i = 0; i < E; ++i) m.v[i] = -INFINITY;
    for (int kh = 0; kh < K; ++kh) {
      const int iy = oy * S - P + kh;
      for (int kw = 0; kw < K; ++kw) {
        const int ix = ox * S - P + kw;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          const Vec16<T> t = Vec16<T>::load(x + ((((size_t)n * H + iy) * W + ix) * ldx + cg * E) * sizeof(T));
#pragma unroll
          for (int i = 0; i < E; ++i) m.v[i] = fmaxf(m.v[i], t.v[i]);
        } else if (iy >= 0 && ix >= 0 && iy < H + padBR && ix < W + padBR) {  // nn.ZeroPad2d cells: value 0
#pragma unroll
          for (int i = 0; i < E; ++i) m.v[i] = fmaxf(m.v[i], 0.f);
        }
      }
    }
    m.store(y + ((((size_t)n * OH + oy) * OW + ox) * ldy + cg * E) * sizeof(T));
  }
}

// ---- SPPF: three chained 5x5/s1/p2 max pools (-inf padding), one (image, 16-byte channel group) plane per workgroup.
// The plane lives in LDS; each 5x5 pool is separable (row max then column max), so a pixel costs 30 LDS reads for the
// three outputs instead of 169 global reads, and x is read from HBM exactly once.
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(const char* x, char* y1, char* y2, char* y3, int N, int H, int W,
                                                         int C, int ldx, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int E = Vec16<T>::E;
  const int CG = C / E;
  const int n = blockIdx.x / CG, cg = blockIdx.x % CG;
  const int HW = H * W;
  u32x4* a = reinterpret_cast<u32x4*>(sm);  // current stage input  [HW]
  u32x4* t = a + HW;                        // row-max scratch       [HW]
  for (int p = threadIdx.x; p < HW; p += 256)
    a[p] = *reinterpret_cast<const u32x4*>(x + (((size_t)n * HW + p) * ldx + cg * E) * sizeof(T));
  __syncthreads();
  char* outs[3] = {y1, y2, y3};
  for (int stage = 0; stage < 3; ++stage) {
    for (int p = threadIdx.x; p < HW; p += 256) {  // horizontal 5-max
      const int py = p / W, px = p - py * W;
      Vec16<T> m = Vec16<T>::load(reinterpret_cast<const char*>(&a[p]));
      for (int dx = -2; dx <= 2; ++dx) {
        const int qx = px + dx;
        if (dx == 0 || qx < 0 || qx >= W) continue;
        const Vec16<T> v = Vec16<T>::load(reinterpret_cast<const char*>(&a[py * W + qx]));
#pragma unroll
        for (int i = 0; i < E; ++i) m.v[i] = fmaxf(m.v[i], v.v[i]);
      }
      m.store(reinterpret_cast<char*>(&t[p]));
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {  // vertical 5-max -> stage output (also next stage's input)
      const int py = p / W, px = p - py * W;
      Vec16<T> m = Vec16<T>::load(reinterpret_cast<const char*>(&t[p]));
      for (int dy = -2; dy <= 2; ++dy) {
        const int qy = py + dy;
        if (dy == 0 || qy < 0 || qy >= H) continue;
        const Vec16<T> v = Vec16<T>::load(reinterpret_cast<const char*>(&t[qy * W + px]));
#pragma unroll
        for (int i = 0; i < E; ++i) m.v[i] = fmaxf(m.v[i], v.v[i]);
      }
      m.store(reinterpret_cast<char*>(&a[p]));
      m.store(outs[stage] + (((size_t)n * HW + p) * ldy + cg * E) * sizeof(T));
    }
    __syncthreads();
  }
}

// The bf16 form: the plane is held as ORDER KEYS - bf16 bits with the sign bit flipped (positive values) or all bits flipped (negative ones), so
// that unsigned 16-bit order = float order - and every 5-max is v_pk_max_u16 on pixel vectors of four key pairs: 4 instructions per neighbour
// instead of 8 unpacks + 8 v_max_f32 (+ 8 packs per result); the keys are made once when the plane is read and undone once per stored vector
// (the max of bf16 values is one of them: bit-exact, -0 < +0).  One thread per pixel (blockDim = HW rounded up to a wave, <= 1024).
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16x2_key(unsigned v) {   // and its own inverse on `v ^ 0x80008000`-ordered input: see sppf_unkey
  return v ^ (0x80008000u | (((v >> 15) & 0x00010001u) * 0x7FFFu));
}
__device__ __forceinline__ unsigned bf16x2_unkey(unsigned k) {
  return k ^ (0x80008000u | (((~k >> 15) & 0x00010001u) * 0x7FFFu));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  const u16x2_t r = __builtin_elementwise_max(*reinterpret_cast<const u16x2_t*>(&a), *reinterpret_cast<const u16x2_t*>(&b));
  return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ u32x4 pk_max4(const u32x4& a, const u32x4& b) {
  return u32x4{pk_max_u16(a[0], b[0]), pk_max_u16(a[1], b[1]), pk_max_u16(a[2], b[2]), pk_max_u16(a[3], b[3])};
}
__global__ __launch_bounds__(1024) void sppf_pool3_bf16_kernel(const char* x, char* y1, char* y2, char* y3, int N, int H, int W, int C, int ldx,
                                                               int ldy) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int CG = C / 8;
  const int n = blockIdx.x / CG, cg = blockIdx.x % CG;
  const int HW = H * W, nth = blockDim.x;
  u32x4* a = reinterpret_cast<u32x4*>(sm);  // current stage input  [HW] (keys)
  u32x4* t = a + HW;                        // row-max scratch       [HW]
  for (int p = threadIdx.x; p < HW; p += nth) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * HW + p) * ldx + cg * 8) * 2);
    a[p] = u32x4{bf16x2_key(v[0]), bf16x2_key(v[1]), bf16x2_key(v[2]), bf16x2_key(v[3])};
  }
  __syncthreads();
  char* outs[3] = {y1, y2, y3};
  for (int stage = 0; stage < 3; ++stage) {
    for (int p = threadIdx.x; p < HW; p += nth) {  // horizontal 5-max
      const int py = p / W, px = p - py * W;
      u32x4 m = a[p];
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const int qx = px + dx;
        if (dx != 0 && qx >= 0 && qx < W) m = pk_max4(m, a[p + dx]);
      }
      t[p] = m;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += nth) {  // vertical 5-max -> stage output (also next stage's input)
      const int py = p / W;
      u32x4 m = t[p];
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int qy = py + dy;
        if (dy != 0 && qy >= 0 && qy < H) m = pk_max4(m, t[p + dy * W]);
      }
      a[p] = m;
      *reinterpret_cast<u32x4*>(outs[stage] + (((size_t)n * HW + p) * ldy + cg * 8) * 2) =
          u32x4{bf16x2_unkey(m[0]), bf16x2_unkey(m[1]), bf16x2_unkey(m[2]), bf16x2_unkey(m[3])};
    }
    __syncthreads();
  }
}

// ---- SPPF front: cv1 (1x1, c1 -> c_, BN folded, SiLU) AND the three chained 5x5 pools as ONE launch (block.py:382-406; yolov8n row 9: 256 -> 128 at
// 20 x 20).  Separately, cv1 (conv1x1_stream) and the pools (above) are two launches of ~11 + ~15 us whose work is a fraction of that: the map is
// tiny.  Here a workgroup owns (image, 16 output channels of cv1): its waves multiply the whole plane - a wave's weights (c1 / 32 fragments) stay in
// registers, the pixels' 32-channel groups are read straight from memory as B fragments (the plane is L2-resident: the image's other channel groups
// read the same bytes) - store y0 = SiLU(cv1) into the concat buffer's first slice AND into LDS as bf16 ORDER KEYS; then the pool stages run on
// the plane as in sppf_pool3_bf16_kernel and write slices 1-3.  Rounding point: bf16 y0 (as the separate launches); the pools are exact.
template <int KT>
__global__ __launch_bounds__(512) void sppf_front_kernel(const char* x, int N, int H, int W, int ldx, const char* wp, const float* bias, char* y, int ldy,
                                                         int c_) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int NT = c_ / 16;
  // workgroups go to the XCDs round-robin by blockIdx: the NT channel groups of one image take blockIdx values that are equal mod 8, so the image's
  // plane is fetched into ONE L2 instead of eight (PMC: 65.7 MB per launch with (n, j) = (b / NT, b % NT) against ~20 MB algorithmic)
  int n, j;
  if ((N & 7) == 0) {
    const int r = (int)blockIdx.x >> 3;
    j = r % NT;
    n = ((int)blockIdx.x & 7) + 8 * (r / NT);
  } else {
    n = blockIdx.x / NT;
    j = blockIdx.x % NT;
  }
  const int HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, l16 = lane & 15;
  u32x4* a = reinterpret_cast<u32x4*>(sm);   // [2 halves of the 16 channels][HW] keys: current stage input
  u32x4* t = a + 2 * HW;                     // row-max scratch, same shape
  u32x4 wf[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) wf[kt] = *reinterpret_cast<const u32x4*>(wp + ((size_t)(kt * NT + j) * 64 + lane) * 16);
  const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + j * 16 + 4 * kg) : f32x4{0.f, 0.f, 0.f, 0.f};
  const char* xn = x + (size_t)n * HW * ldx * 2;
  char* yn = y + (size_t)n * HW * ldy * 2;
  // a wave's m-tiles (wave, wave + 8, ...) two at a time: the 2 KT operand loads of a pair are in flight together (the plane comes from the L2 /
  // Infinity Cache: ~2 us per dependent round; one tile per round made cv1 four rounds long)
  for (int mt0 = wave; mt0 * 16 < HW; mt0 += 16) {
    u32x4 b[2][KT];
    int pix[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      pix[q] = (mt0 + 8 * q) * 16 + l16;
      const int pc = pix[q] < HW ? pix[q] : HW - 1;
      const char* src = xn + (size_t)pc * ldx * 2 + kg * 16;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) b[q][kt] = *reinterpret_cast<const u32x4*>(src + kt * 64);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 acc = b4;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[kt]), *reinterpret_cast<const bf16x8*>(&b[q][kt]), acc, 0, 0, 0);
      // lane (kg, l16): channels 16 j + 4 kg .. + 3 of pixel pix
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[r]));
      const u32x2 pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      if (pix[q] < HW) {
        *reinterpret_cast<u32x2*>(yn + ((size_t)pix[q] * ldy + j * 16 + kg * 4) * 2) = pk;
        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(&a[(kg >> 1) * HW + pix[q]]) + (kg & 1) * 8) = u32x2{bf16x2_key(pk[0]), bf16x2_key(pk[1])};
      }
    }
  }
  __syncthreads();
  // one thread = one pixel, both 8-channel halves (HW <= 512 threads' worth at 20 x 20: a single pass per stage half)
  for (int stage = 0; stage < 3; ++stage) {
    for (int p = tid; p < HW; p += 512) {  // horizontal 5-max
      const int py = p / W, px = p - py * W;
      u32x4 m0 = a[p], m1 = a[HW + p];
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const int qx = px + dx;
        if (dx != 0 && qx >= 0 && qx < W) { m0 = pk_max4(m0, a[p + dx]); m1 = pk_max4(m1, a[HW + p + dx]); }
      }
      t[p] = m0; t[HW + p] = m1;
    }
    __syncthreads();
    for (int p = tid; p < HW; p += 512) {  // vertical 5-max -> slice stage + 1 (and the next stage's input)
      const int py = p / W;
      u32x4 m0 = t[p], m1 = t[HW + p];
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int qy = py + dy;
        if (dy != 0 && qy >= 0 && qy < H) { m0 = pk_max4(m0, t[p + dy * W]); m1 = pk_max4(m1, t[HW + p + dy * W]); }
      }
      a[p] = m0; a[HW + p] = m1;
      char* dst = yn + ((size_t)p * ldy + (stage + 1) * c_ + j * 16) * 2;
      *reinterpret_cast<u32x4*>(dst) = u32x4{bf16x2_unkey(m0[0]), bf16x2_unkey(m0[1]), bf16x2_unkey(m0[2]), bf16x2_unkey(m0[3])};
      *reinterpret_cast<u32x4*>(dst + 16) = u32x4{bf16x2_unkey(m1[0]), bf16x2_unkey(m1[1]), bf16x2_unkey(m1[2]), bf16x2_unkey(m1[3])};
    }
    __syncthreads();
  }
}

// ---- nearest 2x upsample / copy / add -------------------------------------------------------------------------------
template <int MODE>  // 0 copy, 1 upsample2x
__global__ __launch_bounds__(256) void move16_kernel(const char* x, char* y, int N, int OH, int OW, int CG, long ldxB,
                                                     long ldyB) {
  const long total = (long)N * OH * OW * CG;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cg = (int)(gid % CG);
    long pix = gid / CG;
    size_t src;
    if (MODE == 1) {
      const int ox = (int)(pix % OW);
      const int oy = (int)((pix / OW) % OH);
      const int n = (int)(pix / ((long)OW * OH));
      src = ((size_t)n * (OH / 2) + (oy >> 1)) * (OW / 2) + (ox >> 1);
    } else {
      src = (size_t)pix;
    }
    *reinterpret_cast<u32x4*>(y + (size_t)pix * ldyB + cg * 16) = *reinterpret_cast<const u32x4*>(x + src * ldxB + cg * 16);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const char* a, const char* b, char* y, long P, int CG, int lda, int ldb,
                                                  int ldy) {
  constexpr int E = Vec16<T>::E;
  const long total = P * CG;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cg = (int)(gid % CG);
    const size_t pix = gid / CG;
    Vec16<T> va = Vec16<T>::load(a + (pix * lda + cg * E) * sizeof(T));
    const Vec16<T> vb = Vec16<T>::load(b + (pix * ldb + cg * E) * sizeof(T));
#pragma unroll
    for (int i = 0; i < E; ++i) va.v[i] += vb.v[i];
    va.store(y + (pix * ldy + cg * E) * sizeof(T));
  }
}

// ---- NCHW f32 <-> NHWC T (module boundary; tiles transposed through LDS so both sides stay coalesced) ------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, char* y, int C, long HW, int ldy) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const long pp = p0 + tx;
    tile[j][tx] = (c < C && pp < HW) ? x[((size_t)n * C + c) * HW + pp] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long pp = p0 + j;
    const int c = c0 + tx;
    if (pp < HW && c < C) {
      const float v = tile[tx][j];
      char* dst = y + (((size_t)n * HW + pp) * ldy + c) * sizeof(T);
      if constexpr (sizeof(T) == 4) *reinterpret_cast<float*>(dst) = v;
      else *reinterpret_cast<bf16_t*>(dst) = f32_to_bf16(v);
    }
  }
}

// Narrow inputs (an image batch: C = 3): one thread per pixel reads its C planes (each plane coalesced across the wave) and
// writes one 16-byte NHWC group, channels C.. of the group zero.  The 32 x 32 tile form above spent 32 channel slots on 3
// channels: 337 us for a 32 x 3 x 640 x 640 batch against a 60 us HBM floor.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_narrow_kernel(const float* x, char* y, int C, long HW, int ldy, long total) {
  constexpr int E = 16 / sizeof(T);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, pp = i - n * HW;
    float v[E];
#pragma unroll
    for (int c = 0; c < E; ++c) v[c] = c < C ? x[((size_t)n * C + c) * HW + pp] : 0.f;
    char* dst = y + (size_t)i * ldy * sizeof(T);
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                             pack_bf16x2(v[6], v[7])};
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const char* x, float* y, int C, long HW, int ldx) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const long pp = p0 + j;
    const int c = c0 + tx;
    float v = 0.f;
    if (pp < HW && c < C) {
      const char* src = x + (((size_t)n * HW + pp) * ldx + c) * sizeof(T);
      if constexpr (sizeof(T) == 4) v = *reinterpret_cast<const float*>(src);
      else v = bf16_to_f32(*reinterpret_cast<const bf16_t*>(src));
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const long pp = p0 + tx;
    if (c < C && pp < HW) y[((size_t)n * C + c) * HW + pp] = tile[tx][j];
  }
}

inline unsigned grid_for(long total) {
  long b = (total + 255) / 256;
  const long cap = 256L * 16;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define CHECK_VIEW(c, ld, dtype)                                                                      \
  UPA_CHECK_ARG((c) % (16 / upa_elem_size(dtype)) == 0 && (ld) % (16 / upa_elem_size(dtype)) == 0, \
                "channel count / stride must be multiples of 16 bytes")

extern "C" int upa_maxpool2d(const void* x, int n, int h, int w, int c, int ldx, void* y, int oh, int ow, int ldy, int k,
                             int stride, int pad, int pad_br, int dtype, void* stream) {
  UPA_CHECK_ARG(x && y && k >= 1 && stride >= 1, "maxpool2d: bad args");
  CHECK_VIEW(c, ldx, dtype);
  CHECK_VIEW(c, ldy, dtype);
  UPA_CHECK_ARG(oh == (h + pad_br + 2 * pad - k) / stride + 1 && ow == (w + pad_br + 2 * pad - k) / stride + 1,
                "maxpool2d: output shape mismatch");
  const long total = (long)n * oh * ow * (c / (16 / upa_elem_size(dtype)));
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)x,
                       (char*)y, n, h, w, c, ldx, oh, ow, ldy, k, stride, pad, pad_br);
  else
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)x,
                       (char*)y, n, h, w, c, ldx, oh, ow, ldy, k, stride, pad, pad_br);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_sppf_pool3(const void* x, int n, int h, int w, int c, int ldx, void* y1, void* y2, void* y3, int ldy,
                              int dtype, void* stream) {
  UPA_CHECK_ARG(x && y1 && y2 && y3, "sppf_pool3: null pointer");
  CHECK_VIEW(c, ldx, dtype);
  CHECK_VIEW(c, ldy, dtype);
  const size_t lds = (size_t)h * w * 32;
  UPA_CHECK_ARG(lds <= 64 * 1024, "sppf_pool3: plane %dx%d does not fit LDS", h, w);
  const int cg = c / (16 / upa_elem_size(dtype));
  dim3 grid((unsigned)(n * cg));
  if (dtype == UPA_BF16) {
    const int hw = h * w, nth = hw >= 1024 ? 1024 : (hw + 63) / 64 * 64;
    hipLaunchKernelGGL(sppf_pool3_bf16_kernel, grid, dim3(nth), lds, (hipStream_t)stream, (const char*)x, (char*)y1, (char*)y2, (char*)y3, n, h,
                       w, c, ldx, ldy);
  } else
    hipLaunchKernelGGL(sppf_pool3_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const char*)x, (char*)y1,
                       (char*)y2, (char*)y3, n, h, w, c, ldx, ldy);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

/* SPPF front: y[:, 0:c_) = SiLU(cv1(x)) and y[:, c_:4c_) = its three chained 5x5 pools, ONE launch (bf16, c1 = 128 | 256 | 512, c_ % 16 == 0, maps of up
 * to 1024 pixels); UPA_EUNSUPPORTED otherwise (callers run upa_conv2d_bias_act + upa_sppf_pool3).  w_packed / bias: upa_pack_conv_weight of cv1. */
extern "C" int upa_sppf_front(const void* x, int n, int h, int w, int c1, int ldx, const void* w_packed, const float* bias, void* y, int c_, int ldy,
                              int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && y && w_packed && n > 0 && h > 0 && w > 0, "sppf_front: bad args");
  const long hw = (long)h * w;
  if (UPA_OPT(opts, no_sppf_front) || dtype != UPA_BF16 || (c1 != 128 && c1 != 256 && c1 != 512) || c_ % 16 != 0 || c_ <= 0 || ldx % 8 != 0 || ldy % 8 != 0 ||
      hw > 1024 || ldy < 4 * c_ || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 || (long)n * hw * (long)(ldx > ldy ? ldx : ldy) * 2 >= (1L << 31)) {
    upa_set_error("sppf_front: outside the fused form");
    return UPA_EUNSUPPORTED;
  }
  const size_t lds = (size_t)hw * 64;   // 2 halves x (plane + scratch) x 16 B
  const dim3 grid((unsigned)(n * (c_ / 16)));
  hipStream_t st = (hipStream_t)stream;
  if (c1 == 128) hipLaunchKernelGGL(sppf_front_kernel<4>, grid, dim3(512), lds, st, (const char*)x, n, h, w, ldx, (const char*)w_packed, bias, (char*)y, ldy, c_);
  else if (c1 == 256) hipLaunchKernelGGL(sppf_front_kernel<8>, grid, dim3(512), lds, st, (const char*)x, n, h, w, ldx, (const char*)w_packed, bias, (char*)y, ldy, c_);
  else hipLaunchKernelGGL(sppf_front_kernel<16>, grid, dim3(512), lds, st, (const char*)x, n, h, w, ldx, (const char*)w_packed, bias, (char*)y, ldy, c_);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_upsample2x(const void* x, int n, int h, int w, int c, int ldx, void* y, int ldy, int dtype,
                              void* stream) {
  UPA_CHECK_ARG(x && y, "upsample2x: null pointer");
  CHECK_VIEW(c, ldx, dtype);
  CHECK_VIEW(c, ldy, dtype);
  const int es = upa_elem_size(dtype);
  const int cg = c * es / 16;
  const long total = (long)n * (2 * h) * (2 * w) * cg;
  hipLaunchKernelGGL(move16_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)x, (char*)y,
                     n, 2 * h, 2 * w, cg, (long)ldx * es, (long)ldy * es);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_copy_view(const void* x, int n, int h, int w, int c, int ldx, void* y, int ldy, int dtype,
                             void* stream) {
  UPA_CHECK_ARG(x && y, "copy_view: null pointer");
  CHECK_VIEW(c, ldx, dtype);
  CHECK_VIEW(c, ldy, dtype);
  const int es = upa_elem_size(dtype);
  const int cg = c * es / 16;
  const long total = (long)n * h * w * cg;
  hipLaunchKernelGGL(move16_kernel<0>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const char*)x, (char*)y,
                     n, h, w, cg, (long)ldx * es, (long)ldy * es);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_add_view(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int n, int h, int w, int c,
                            int dtype, void* stream) {
  UPA_CHECK_ARG(a && b && y, "add_view: null pointer");
  CHECK_VIEW(c, lda, dtype);
  CHECK_VIEW(c, ldb, dtype);
  CHECK_VIEW(c, ldy, dtype);
  const long P = (long)n * h * w;
  const int cg = c / (16 / upa_elem_size(dtype));
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(P * cg)), dim3(256), 0, (hipStream_t)stream, (const char*)a,
                       (const char*)b, (char*)y, P, cg, lda, ldb, ldy);
  else
    hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(P * cg)), dim3(256), 0, (hipStream_t)stream, (const char*)a,
                       (const char*)b, (char*)y, P, cg, lda, ldb, ldy);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_nchw_to_nhwc(const float* x, int n, int c, int h, int w, void* y, int ldy, int dtype, void* stream) {
  UPA_CHECK_ARG(x && y && n > 0 && c > 0, "nchw_to_nhwc: bad args");
  const long hw = (long)h * w;
  const int E = 16 / upa_elem_size(dtype);
  if (c <= E && ldy % E == 0 && ((uintptr_t)y % 16) == 0) {  // image batches: one 16-byte group per pixel
    const long total = (long)n * hw;
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    if (dtype == UPA_BF16)
      hipLaunchKernelGGL(nchw_to_nhwc_narrow_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (char*)y, c, hw, ldy, total);
    else
      hipLaunchKernelGGL(nchw_to_nhwc_narrow_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (char*)y, c, hw, ldy, total);
    UPA_LAUNCH_CHECK();
    return UPA_OK;
  }
  dim3 grid((unsigned)((hw + 31) / 32), (unsigned)cdiv(c, 32), (unsigned)n);
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (char*)y, c, hw, ldy);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (char*)y, c, hw, ldy);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_nhwc_to_nchw(const void* x, int n, int h, int w, int c, int ldx, float* y, int dtype, void* stream) {
  UPA_CHECK_ARG(x && y && n > 0 && c > 0, "nhwc_to_nchw: bad args");
  const long hw = (long)h * w;
  dim3 grid((unsigned)((hw + 31) / 32), (unsigned)cdiv(c, 32), (unsigned)n);
  if (dtype == UPA_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, y, c, hw, ldx);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, y, c, hw, ldx);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LetterBox on uint8 HWC frames (ultralytics/data/augment.py:1544-1700, the predictor's pre_transform,
// engine/predictor.py:151-173): bilinear resize to (new_h, new_w) + constant border, one pass, one thread per output pixel.
// The resize is OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic (imgproc/src/resize.cpp: coefficients
// saturate_cast<short>(w * 2048), horizontal pass in int, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16)
// + 2) >> 2), restated in oracle/letterbox.py; integer results are bit-exact against that oracle.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct LetterboxParams {
  const unsigned char* src;
  unsigned char* dst;
  long src_image_stride;  // bytes
  int src_row_stride;     // bytes
  int h0, w0, H, W, new_h, new_w, top, left, pad;
  int resize;             // 0: the unpadded size equals the source size (copy)
  double scale_x, scale_y;
};

__device__ __forceinline__ void lb_coeff(int d, double scale, int ssize, bool pin, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (pin) {  // horizontal set-up: index pinned AND fraction dropped at the borders
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    s0 = s;
    s1 = s + 1 < ssize ? s + 1 : ssize - 1;
  } else {    // vertical: fraction kept, rows clipped into the image
    s0 = s < 0 ? 0 : (s > ssize - 1 ? ssize - 1 : s);
    s1 = s + 1 < 0 ? 0 : (s + 1 > ssize - 1 ? ssize - 1 : s + 1);
  }
  const float w0 = rintf((1.f - f) * 2048.f), w1 = rintf(f * 2048.f);
  c0 = (int)fminf(fmaxf(w0, -32768.f), 32767.f);
  c1 = (int)fminf(fmaxf(w1, -32768.f), 32767.f);
}

__global__ __launch_bounds__(256) void letterbox_u8_kernel(const LetterboxParams p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.W || y >= p.H) return;
  unsigned char* o = p.dst + ((size_t)blockIdx.z * p.H + y) * (size_t)p.W * 3 + (size_t)x * 3;
  const int dx = x - p.left, dy = y - p.top;
  if (dx < 0 || dx >= p.new_w || dy < 0 || dy >= p.new_h) {
    o[0] = o[1] = o[2] = (unsigned char)p.pad;
    return;
  }
  const unsigned char* img = p.src + (size_t)blockIdx.z * p.src_image_stride;
  if (!p.resize) {
    const unsigned char* s = img + (size_t)dy * p.src_row_stride + dx * 3;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
    return;
  }
  int x0, x1, a0, a1, y0, y1, b0, b1;
  lb_coeff(dx, p.scale_x, p.w0, true, x0, x1, a0, a1);
  lb_coeff(dy, p.scale_y, p.h0, false, y0, y1, b0, b1);
  const unsigned char* r0 = img + (size_t)y0 * p.src_row_stride;
  const unsigned char* r1 = img + (size_t)y1 * p.src_row_stride;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int s0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
    const int s1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
    int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[c] = (unsigned char)v;
  }
}
}  // namespace

extern "C" int upa_letterbox_u8(const void* src, int n, int h0, int w0, long src_image_stride, int src_row_stride, void* dst,
                                int H, int W, int new_h, int new_w, int top, int left, int pad_value, void* stream) {
  UPA_CHECK_ARG(src && dst && n > 0 && h0 > 0 && w0 > 0 && H > 0 && W > 0, "letterbox_u8: bad args");
  UPA_CHECK_ARG(new_h > 0 && new_w > 0 && top >= 0 && left >= 0 && top + new_h <= H && left + new_w <= W,
                "letterbox_u8: the resized frame does not fit the output");
  UPA_CHECK_ARG(pad_value >= 0 && pad_value <= 255 && src_row_stride >= w0 * 3, "letterbox_u8: bad pad value / row stride");
  LetterboxParams p;
  p.src = (const unsigned char*)src; p.dst = (unsigned char*)dst;
  p.src_image_stride = src_image_stride; p.src_row_stride = src_row_stride;
  p.h0 = h0; p.w0 = w0; p.H = H; p.W = W; p.new_h = new_h; p.new_w = new_w; p.top = top; p.left = left; p.pad = pad_value;
  p.resize = !(new_h == h0 && new_w == w0);
  p.scale_x = 1.0 / ((double)new_w / (double)w0);  // resize.cpp: scale = 1. / inv_scale, inv_scale = dsize / ssize
  p.scale_y = 1.0 / ((double)new_h / (double)h0);
  hipLaunchKernelGGL(letterbox_u8_kernel, dim3((unsigned)cdiv(W, 64), (unsigned)cdiv(H, 4), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
