// Shared device/host helpers for libupa_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/upa.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef unsigned short bf16_t;  // raw bf16 bits

// A field of the caller's upa_opts (NULL, or shorter than this header's struct: 0 = the default)
#include <stddef.h>
#define UPA_OPT(o, field) \
  (((o) != nullptr && (o)->size >= offsetof(upa_opts, field) + sizeof((o)->field)) ? (int)(o)->field : 0)

// Kernel ablation switches (no loads / no MFMA / no stores ...: tools/experiments/ablate_sweep.sh) exist only in the
// -DUPA_ABLATE build (`make ablate` -> libupa_hip_ablate.so); the product kernels carry neither the tests nor the field.
#ifdef UPA_ABLATE
#define UPA_ABL(p, bits) ((p).ablate & (bits))
#else
#define UPA_ABL(p, bits) (0)
#endif

void upa_set_error(const char* fmt, ...);

#define UPA_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      upa_set_error(__VA_ARGS__);     \
      return UPA_EINVAL;              \
    }                                 \
  } while (0)

#define UPA_LAUNCH_CHECK()                                       \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) {                                     \
      upa_set_error("launch failed: %s", hipGetErrorString(e__)); \
      return UPA_ELAUNCH;                                        \
    }                                                            \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even f32 -> bf16 (a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two round-to-nearest-even conversions in one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 r = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  return *reinterpret_cast<unsigned*>(&r);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int E = 4;       // elements per 16 bytes
  static constexpr int KT_CH = 16;  // channels per 64-byte k-tile
  __device__ static float load(const float* p) { return *p; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int E = 8;
  static constexpr int KT_CH = 32;
  __device__ static float load(const bf16_t* p) { return bf16_to_f32(*p); }
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == UPA_ACT_SILU) return silu_f(v);
  if (act == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

// Raise a kernel's dynamic-LDS limit to the whole 160 KB of a gfx950 CU (less its static __shared__), once per kernel instantiation (C++11 static
// initialisation: thread safe, no API call on later launches, and no launch ever lowers a limit another launch - or a
// captured graph node - relies on).
template <auto Kern>
inline hipError_t upa_full_lds() {
  static const hipError_t e = [] {
    hipFuncAttributes fa;
    int room = 160 * 1024;
    if (hipFuncGetAttributes(&fa, (const void*)Kern) == hipSuccess) room -= (int)fa.sharedSizeBytes;  // static __shared__
    return hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, room);
  }();
  return e;
}

// Zero n 32-bit words with a kernel.  Used instead of hipMemsetAsync wherever the launch sequence may be captured into a
// hipGraph: with several graphs of the same step in flight on separate streams, the runtime's memset nodes now and then
// left the NMS candidate counters un-zeroed (ROCm 7.2) - a stale counter became a negative slot index and the candidate
// store faulted gigabytes below the workspace ("Memory access fault by GPU", nms_candidates_kernel under
// librocm-debug-agent).  A kernel node has none of that.
__global__ void upa_zero_words_kernel(unsigned* p, int n);
inline void upa_zero_words(void* p, int n_words, hipStream_t s) {
  hipLaunchKernelGGL(upa_zero_words_kernel, dim3((n_words + 255) / 256), dim3(256), 0, s, (unsigned*)p, n_words);
}

// ---- LDS pixel pitch of a [pixel][64 | 128 B] image whose 16-byte groups are XOR-swizzled by the pixel index and read back as
// MFMA B fragments (ds_read_b128: lane (g, r) reads group g of the r-th pixel of a 16-pixel m-tile; m-tiles enumerate a W-wide
// tile row-major, so one may straddle two or three image rows).  The hardware serves a ds_read_b128 in four groups of 16 lanes
// - rows {0..3, 12..15} of one g with rows {4..11} of its neighbour - and with the swizzles used here (conv_big / conv_pair /
// c2f_fused) a group is conflict-free exactly when the 8 pixel indices of each row set are distinct mod 8.  Consecutive pixels
// always are; a straddling m-tile is when the image's row step is right for the tile width (W = 40: step = 0 mod 8; W = 20: 4
// mod 8).  These helpers count the extra LDS cycles of one tap over a tile and pick the smallest pitch >= min_pitch with the
// fewest (0 for every tile width the dispatchers use).  row_mult: image rows between consecutive tile rows (the conv stride).
static constexpr int upa_lds_conflict_cycles(int W, int npix, int row_step) {
  int extra = 0;
  for (int m0 = 0; m0 < npix; m0 += 16) {
    for (int set = 0; set < 2; ++set) {
      int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, worst = 1;
      for (int r = 0; r < 16; ++r) {
        if ((set == 0) != (r < 4 || r >= 12)) continue;
        const int pp = m0 + r;
        if (pp >= npix) continue;
        const int q = ((pp / W) * row_step + pp % W) & 7;
        if (++cnt[q] > worst) worst = cnt[q];
      }
      extra += worst - 1;
    }
  }
  return extra;
}
static constexpr int upa_lds_pick_pitch(int min_pitch, int W, int npix, int row_mult) {
  int best = min_pitch, best_cost = 1 << 30;
  for (int P = min_pitch; P < min_pitch + 8; ++P) {
    const int c = upa_lds_conflict_cycles(W, npix, (P * row_mult) & 7);
    if (c < best_cost) { best_cost = c; best = P; }
  }
  return best;
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md: blocks b and b + 8 share one;
// speed only, never correctness), each with its own L2.  Neighbouring tiles share halo pixels, so a launch whose workgroup b takes
// tile b scatters every neighbourhood over all eight L2s and each halo is fetched from the fabric by every tile that needs it.  This
// maps workgroup b to tile start(b % 8) + b / 8, where XCD x owns the CONTIGUOUS tile range [start(x), start(x + 1)) (sizes differ
// by at most one): a bijection on [0, total), and neighbours in the tile list now meet in one L2.
__device__ __forceinline__ int upa_xcd_tile(int b, int total) {
  const int q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
  return x * q + (x < r ? x : r) + i;
}

static inline int upa_elem_size(int dtype) { return dtype == UPA_BF16 ? 2 : 4; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- profiling build only (-DUPA_STAMP, tools/bench_conv.py --stamps, tools/experiments/c2f_stamps.py): wave 0 of each of the
// first 4096 workgroups records s_memtime at its phase boundaries (slot 15: hardware id); UPA_STAMP_DEFINE(tag) in a translation
// unit gives it the buffer and the reader upa_debug_stamps_<tag>(out, count).
#ifdef UPA_STAMP
#define UPA_STAMP_DEFINE(tag)                                                                                  \
  __device__ unsigned long long g_upa_stamps[4096 * 16];                                                       \
  extern "C" int upa_debug_stamps_##tag(unsigned long long* out, int count) {                                  \
    const int rc = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_upa_stamps), (size_t)count * 8) == hipSuccess ? 0 : -1; \
    return rc;                                                                                                 \
  }                                                                                                            \
  extern "C" int upa_debug_stamps_clear_##tag() {                                                              \
    void* d = nullptr;                                                                                         \
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_upa_stamps)) != hipSuccess) return -1;                            \
    return hipMemset(d, 0, sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : -1;                     \
  }
#define UPA_STAMP_AT(k)                                                                                        \
  do {                                                                                                         \
    if ((threadIdx.x >> 6) == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                                     \
      unsigned long long t_;                                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
      if ((threadIdx.x & 63) == 0) g_upa_stamps[blockIdx.x * 16 + (k)] = t_;                                   \
    }                                                                                                          \
  } while (0)
#define UPA_STAMP_HWID()                                                                                       \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                                            \
      unsigned hw_, xcc_;                                                                                      \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                        \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                      \
      g_upa_stamps[blockIdx.x * 16 + 15] = ((unsigned long long)xcc_ << 32) | hw_;                             \
    }                                                                                                          \
  } while (0)
#else
#define UPA_STAMP_DEFINE(tag)
#define UPA_STAMP_AT(k) do {} while (0)
#define UPA_STAMP_HWID() do {} while (0)
#endif
