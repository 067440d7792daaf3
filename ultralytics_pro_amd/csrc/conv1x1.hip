// Streaming pointwise (1x1, stride 1) convolution for gfx950, bf16 perf mode: C2f cv1 / cv2, SPPF cv1 / cv2 and the
// final 1x1 of every Detect branch (Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197 with BN folded per
// utils/torch_utils.py:236-266; nn.Conv2d(c, 4*reg_max | nc, 1) of head.py:94-100).  Same math and packed-weight layout as
// conv.hip; a 1x1 conv is a GEMM [pixels x Cin] x [Cin x Cout] at 30-130 FLOP/B, i.e. HBM-bound on MI355X, so the kernel
// is built like a copy engine with MFMAs attached:
//   * the (n, h, w) pixels of an NHWC view have one uniform stride, so the image is a flat list of 16-pixel tiles; a wave
//     is a persistent worker that owns groups of MT tiles x all NTW*16 output channels of its workgroup - every input
//     byte is fetched once per workgroup row, nothing is staged twice;
//   * input tiles go global -> LDS by LDS-DMA straight into MFMA B-fragment order (lane (p, g) fetches the 16 bytes of
//     pixel p, channel group g: the LDS image of a (tile, k-tile) is 1 KiB lane-linear, read back by one conflict-free
//     ds_read_b128), into a wave-private ring of three (group, k-tile) stages: two stages are always in flight while the
//     third is multiplied, with COUNTED s_waitcnt vmcnt(N) - loads, LDS-DMAs and stores retire in issue order, so N is
//     the number of vector-memory instructions this wave issued after the stage it needs (tracked in a scalar);
//   * the workgroup's weight slice (KTT x NTW KiB, A-fragment order) is DMA'd into LDS once; A fragments are re-read from
//     LDS per k-tile and reused over the MT pixel tiles;
//   * no workgroup barrier after the weight fill, no per-tile index arithmetic beyond one multiply per (tile, group);
//   * epilogue from the accumulators: bias, SiLU, bf16 pack, v_permlane16_swap pairs channel quads into 16-byte stores.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* c1gptr_t;
typedef __attribute__((address_space(3))) void* c1lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_c1_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
// stages of the wave-private input ring: two stages (of MT KiB) are in flight per wave while the third is multiplied.
// Measured on MI355X (bs 32 yolov8n layers): 4 stages at MT 2 and 8 at MT 1 were slower wherever the bigger ring cost a
// co-resident workgroup (192->128 @40x40: 15.5 vs 13.4 us) and equal elsewhere - occupancy, not ring depth, hides latency.
constexpr int ring_of(int) { return 3; }

template <class F, int... Is>
__device__ __forceinline__ void for_stages(F& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// s_waitcnt takes an immediate.  In steady state the count a wave needs is one of three values - the DMAs of the RING-1
// younger stages plus the stores of zero, one or two epilogues - so three uniform compares pick it; anything else (the
// first / last groups of a wave, masked tail stores) waits for everything, which is always safe.  (A generic 49-way
// switch here compiled to a compare tree that made the kernel SALU-bound: 150 s_waitcnt sites, 430 branches.)
template <int N>
__device__ __forceinline__ void vm_wait_imm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int C0, int SE>
__device__ __forceinline__ void vm_wait(int n) {
  if (n == C0) vm_wait_imm<C0>();
  else if (n == C0 + SE) vm_wait_imm<(C0 + SE < 64 ? C0 + SE : 0)>();
  else if (n == C0 + 2 * SE) vm_wait_imm<(C0 + 2 * SE < 64 ? C0 + 2 * SE : 0)>();
  else if (n >= 63) vm_wait_imm<63>();  // more younger operations than the field holds: any smaller count is safe
  else vm_wait_imm<0>();
}
}  // namespace

template <int NTW, int MT, int WAVES, int EPI = 0>
__global__ __launch_bounds__(WAVES * 64) void conv1x1_stream_kernel(const C1Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if UPA_ABL(p, 32) return;  // debug: launch + workgroup dispatch only
  constexpr int RING = ring_of(MT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p16 = lane & 15, kg = lane >> 4;
  const int nt0 = blockIdx.y * NTW;
  char* wl = smem;                                                   // weights [kt][j][lane][16 B]
  char* ring = smem + p.KTT * NTW * 1024 + wave * (RING * MT * 1024);  // wave-private input stages
  float* bl = reinterpret_cast<float*>(smem + p.KTT * NTW * 1024 + WAVES * (RING * MT * 1024));  // bias slice

  const int gw = blockIdx.x * WAVES + wave, GW = gridDim.x * WAVES;
  const int ldx2 = p.ldx * 2, ldy2 = p.ldy * 2;
  const int ngrpAll = p.Cin >> 3;  // valid 16-byte channel groups of a pixel

  // ---- issue side: DMA of unit (group ig, k-tile ikt) into a ring stage; `seq` counts this wave's vector-memory
  // instructions, mark[s] = seq right after stage s was requested
  int seq = 0;
  int mark[RING];
  int ig = gw, ikt = 0;
  unsigned ioff[MT];  // byte offset of this lane's pixel in each tile of the issue group (0xffffffff: past the end)
  unsigned uoff[MT];  // ... of its source pixel (y / 2, x / 2) in the half-resolution tensor of a virtual Upsample + Concat
  auto issue_group = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int pix = (ig * MT + i) * 16 + p16;
      ioff[i] = pix < p.P ? (unsigned)pix * (unsigned)ldx2 + (unsigned)(kg * 16) : 0xffffffffu;
      if (EPI == 0 && p.upKT) {  // uniform (the Detect tails never take a virtual input)
        const unsigned row = __umulhi((unsigned)pix, p.upMagicW);  // n * H + y
        const unsigned xx = (unsigned)pix - row * (unsigned)p.upW;
        const unsigned nn = __umulhi(row, p.upMagicH);
        const unsigned yy = row - nn * (unsigned)p.upH;
        const unsigned sp = (nn * (unsigned)(p.upH >> 1) + (yy >> 1)) * (unsigned)(p.upW >> 1) + (xx >> 1);
        uoff[i] = sp * (unsigned)(p.up_ld * 2) + (unsigned)(kg * 16);
      }
    }
  };
  issue_group();
  auto issue = [&](int stage) __attribute__((always_inline)) {
    if (ig < p.groups) {
      const bool chok = kg < ngrpAll - ikt * 4;
      const bool fromUp = EPI == 0 && ikt < p.upKT;  // uniform
      const char* xk = (fromUp ? p.up : p.x) + ikt * 64;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const char* src = (chok && ioff[i] != 0xffffffffu && !UPA_ABL(p, 1)) ? xk + (fromUp ? uoff[i] : ioff[i])
                                                                            : reinterpret_cast<const char*>(g_c1_zero16);
        __builtin_amdgcn_global_load_lds((c1gptr_t)src, (c1lptr_t)(ring + (stage * MT + i) * 1024), 16, 0, 0);
      }
      seq += MT;
      if (++ikt == p.KTT) {
        ikt = 0;
        ig += GW;
        issue_group();
      }
    }
    mark[stage] = seq;
  };

  f32x4 acc[MT][NTW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // EPI 3 (training forward, act none, no bias): this wave's running per-channel sum / sum of squares of the bf16-ROUNDED values it
  // stores, over all its pixel groups - lane (kg, p16) keeps the sums of its own pixel column; they meet at the end of the kernel
  float ssum[EPI == 3 ? NTW : 1][4], ssq[EPI == 3 ? NTW : 1][4];
  if constexpr (EPI == 3) {
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) ssum[j][q] = ssq[j][q] = 0.f;
  }
  auto epilogue = [&](int g, auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    auto act = [](float v) __attribute__((always_inline)) {
      if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
      else return v;
    };
    const int co0 = nt0 * 16;
    f32x4 biasv[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) biasv[j] = *reinterpret_cast<const f32x4*>(bl + j * 16 + kg * 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int tb = (g * MT + i) * 16;  // uniform
      if (tb >= p.P) continue;
      const int pix = tb + p16;
      const bool pok = pix < p.P && !UPA_ABL(p, 4);
      if constexpr (EPI == 1) {  // Detect box branch: DFL + dist2bbox + stride on the accumulators (detect_epi.h)
        static_assert(EPI != 1 || NTW == 4, "box branch = 4 sides x 16 bins");
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + biasv[j];
        int db, da;
        upa_detect_split(p.de, pix < p.P ? pix : p.P - 1, db, da);
        upa_detect_box_store(p.de, v, db, da, pok, kg);
        ++seq;
      } else if constexpr (EPI == 2) {  // Detect class branch: sigmoid, channel-major f32 rows
        int db, da;
        upa_detect_split(p.de, pix < p.P ? pix : p.P - 1, db, da);
        float best_ = -1.f;  // this lane's running first maximum over its classes of all n-tiles (NMS prefilter key)
        int bc_ = 0;
        if (p.de.keys_only) {  // uniform: only the best-class keys leave the kernel (upa_opts.keys_only, as the conv_big class tail): no
                               // score rows, one sigmoid per pixel, and nothing to add to `seq` (a count that is too small only waits more)
          f32x4 lg[NTW];
#pragma unroll
          for (int j = 0; j < NTW; ++j) lg[j] = acc[i][j] + biasv[j];
          upa_detect_cls_keys_only<NTW>(p.de, lg, pok, kg, best_, bc_);
        } else {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          upa_detect_cls_store(p.de, acc[i][j] + biasv[j], j, db, da, pok, kg, best_, bc_);
          seq += 16 * j + 3 < p.de.nc ? 4 : 0;  // only stores that lane row 0 certainly issues are counted (the count may
                                               // only be too small: a wait for more than needed is always safe)
          if (16 * j + 3 >= p.de.nc)
            for (int q = 0; q < 4; ++q) seq += 16 * j + q < p.de.nc ? 1 : 0;
        }
        }
        // (not counted in seq: the count may only be too small)
        if (p.de.best_keys) upa_detect_best_key_store(p.de, best_, bc_, db, da, pok, lane);  // uniform
      }
      if (EPI != 0 && p.y == nullptr) continue;  // decoded rows only (the raw maps are not materialised)
      char* yrow = p.y + ((size_t)pix * ldy2 + co0 * 2);
#pragma unroll
      for (int j = 0; j + 1 < NTW; j += 2) {
        if (nt0 + j >= p.NTn) continue;  // uniform
        // 16-lane row kg holds channels 16j+4kg..+3 (tile j) and 16(j+1)+4kg..+3 (tile j+1); after the swap even rows
        // own 8 consecutive channels of tile j, odd rows 8 of tile j+1
        const int cb = 16 * (j + (kg & 1)) + 8 * (kg >> 1);
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = act(acc[i][j][q] + biasv[j][q]);
          v1[q] = act(acc[i][j + 1][q] + biasv[j + 1][q]);
        }
        const unsigned pk[2][2] = {{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3])}, {pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])}};
        if constexpr (EPI == 3) {
          const float mk = pok ? 1.f : 0.f;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const float a = __uint_as_float(pk[jj][h2] << 16) * mk, b = __uint_as_float(pk[jj][h2] & 0xFFFF0000u) * mk;
              ssum[j + jj][2 * h2] += a; ssq[j + jj][2 * h2] = fmaf(a, a, ssq[j + jj][2 * h2]);
              ssum[j + jj][2 * h2 + 1] += b; ssq[j + jj][2 * h2 + 1] = fmaf(b, b, ssq[j + jj][2 * h2 + 1]);
            }
        }
        auto lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (pok && co0 + cb < p.Cout) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        ++seq;  // lane (pixel tb, kg 0) is always active here: the store is certainly issued
      }
      if constexpr (NTW & 1) {
        constexpr int j = NTW - 1;
        if (nt0 + j < p.NTn) {
          const int cb = 16 * j + 4 * kg;
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = act(acc[i][j][q] + biasv[j][q]);
          if (pok && co0 + cb < p.Cout)
            *reinterpret_cast<u32x2*>(yrow + cb * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          ++seq;
        }
      }
    }
  };

  // ---- compute side
  int cg = gw, ckt = 0;
  bool running = true;
  auto step = [&](auto stage_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(stage_tag)::value;
    issue((S + RING - 1) % RING);  // the stage multiplied one step ago is free again
    // stores per epilogue in steady state: bf16 rows (two n-tiles per 16-byte store) / one box row / one f32 row per class
    constexpr int SE = EPI == 0 ? MT * ((NTW + 1) / 2) : (EPI == 1 ? MT : MT * 4 * NTW);
    vm_wait<(RING - 1) * MT, SE>(seq - mark[S]);
    if (!UPA_ABL(p, 8)) {
      const char* st = ring + S * MT * 1024 + lane * 16;
      const char* wk = wl + ckt * (NTW * 1024) + lane * 16;
      u32x4 b[MT], a[NTW];
#pragma unroll
      for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const u32x4*>(st + i * 1024);
#pragma unroll
      for (int j = 0; j < NTW; ++j) a[j] = *reinterpret_cast<const u32x4*>(wk + j * 1024);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                              *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
    }
    if (++ckt == p.KTT) {
      if (p.act == UPA_ACT_SILU) epilogue(cg, std::integral_constant<int, UPA_ACT_SILU>{});
      else epilogue(cg, std::integral_constant<int, UPA_ACT_NONE>{});
      zero_acc();
      ckt = 0;
      cg += GW;
      running = cg < p.groups;
    }
  };
  // the first input stages leave before the weights: one memory round trip for both
  auto first = [&](auto stage_tag) __attribute__((always_inline)) { issue(decltype(stage_tag)::value); };
  for_stages(first, std::make_integer_sequence<int, RING - 1>{});
  // ---- weight slice -> LDS (once per workgroup); n-tiles past the packed weights read zeros
  for (int q = wave; q < p.KTT * NTW; q += WAVES) {
    const int kt = q / NTW, j = q - kt * NTW;
    const char* src = reinterpret_cast<const char*>(g_c1_zero16);
    if (nt0 + j < p.NTn && !UPA_ABL(p, 2)) src = p.w + ((size_t)(kt * p.NTn + nt0 + j) * 1024 + lane * 16);
    __builtin_amdgcn_global_load_lds((c1gptr_t)src, (c1lptr_t)(wl + q * 1024), 16, 0, 0);
  }
  // bias slice -> LDS (read back per epilogue: keeps NTW*4 registers free for the accumulators)
  if (threadIdx.x < NTW * 16) {
    const int co = nt0 * 16 + threadIdx.x;
    bl[threadIdx.x] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (gw >= p.groups) {
    if constexpr (EPI != 3) return;
    running = false;  // (EPI 3: a wave without pixel groups still takes part in the workgroup's row of sums)
  }
  auto guarded = [&](auto stage_tag) __attribute__((always_inline)) {
    if (running) step(stage_tag);
  };
  while (running) for_stages(guarded, std::make_integer_sequence<int, RING>{});
  if constexpr (EPI == 3) {
    static_assert(EPI != 3 || (NTW % 2) == 0, "the statistics epilogue pairs n-tiles");
    // the lane's sums over the 16 pixel columns of its row group (fixed butterfly: xor 1, xor 2, mirror in 8, mirror in 16), then over the
    // workgroup's waves in wave order through LDS (the weight slice is dead once every wave is here): row blockIdx.x of p.stats
    auto row16 = [](float v) __attribute__((always_inline)) {
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, true));
      return v;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (zero-page pieces a wave may still have in flight into its ring)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { ssum[j][q] = row16(ssum[j][q]); ssq[j][q] = row16(ssq[j][q]); }
      if (p16 == 0) {
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * (NTW * 16) + j * 16 + 4 * kg) = f32x4{ssum[j][0], ssum[j][1], ssum[j][2], ssum[j][3]};
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * (NTW * 16) + j * 16 + 4 * kg) = f32x4{ssq[j][0], ssq[j][1], ssq[j][2], ssq[j][3]};
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * NTW * 16) {
      const int k = (int)threadIdx.x / (NTW * 16), cl = (int)threadIdx.x - k * (NTW * 16);
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < WAVES; ++m) t += red[(m * 2 + k) * (NTW * 16) + cl];
      const int ch = nt0 * 16 + cl;
      if (ch < p.stats_ld) p.stats[((size_t)blockIdx.x * 2 + k) * p.stats_ld + ch] = t;
    }
  }
}

namespace {

template <int NTW, int MT, int WAVES, int EPI>
int launch_c1_inst(const C1Params& p, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv1x1_stream_kernel<NTW, MT, WAVES, EPI>;
  (void)upa_full_lds<conv1x1_stream_kernel<NTW, MT, WAVES, EPI>>();
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int NTW, int EPI = 0>
int launch_c1_ntw(const C1Params& p, int mt, int waves, dim3 grid, size_t lds, hipStream_t s) {
  if (waves == 8) {
    if (mt == 4) return launch_c1_inst<NTW, 4, 8, EPI>(p, grid, lds, s);
    if (mt == 2) return launch_c1_inst<NTW, 2, 8, EPI>(p, grid, lds, s);
    return launch_c1_inst<NTW, 1, 8, EPI>(p, grid, lds, s);
  }
  if (mt == 4) return launch_c1_inst<NTW, 4, 4, EPI>(p, grid, lds, s);
  if (mt == 2) return launch_c1_inst<NTW, 2, 4, EPI>(p, grid, lds, s);
  return launch_c1_inst<NTW, 1, 4, EPI>(p, grid, lds, s);
}

}  // namespace

bool upa_conv1x1_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, bool residual, int k, int stride,
                          int pad, int act, int dtype, const upa_opts* opts) {
  if (UPA_OPT(opts, no_1x1)) return false;
  if (dtype != UPA_BF16 || k != 1 || stride != 1 || pad != 0 || residual) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE) return false;
  if (cin % 8 != 0 || cout % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0) return false;
  const long px = (long)n * h * w;
  if (px * ldx * 2 >= (1L << 31) - 4096 || px * ldy * 2 >= (1L << 31) - 4096) return false;  // 32-bit byte offsets
  const int ktt = (cin + 31) / 32;
  const int ntw = cout > 128 ? 8 : (cout + 15) / 16;
  if ((size_t)ktt * ntw * 1024 + 4 * ring_of(1) * 1024 + 512 > 160 * 1024) return false;  // weight slice + the smallest ring
  return true;
}

static int c1_launch(C1Params p, int n_pixels, int query_only, int* variant, int* rows, long max_rows, void* stream, const upa_opts* opts);
int upa_conv1x1_launch(C1Params p, int n_pixels, int query_only, int* variant, void* stream, const upa_opts* opts) {
  return c1_launch(p, n_pixels, query_only, variant, nullptr, 0, stream, opts);
}
int upa_conv1x1_launch_stats(C1Params p, int n_pixels, int* rows, long max_rows, void* stream, const upa_opts* opts) {
  if (!p.stats || p.act != UPA_ACT_NONE || p.bias || p.up) return UPA_EUNSUPPORTED;
  p.epi = 3;
  return c1_launch(p, n_pixels, 0, nullptr, rows, max_rows, stream, opts);
}
static int c1_launch(C1Params p, int n_pixels, int query_only, int* variant, int* rows, long max_rows, void* stream, const upa_opts* opts) {
  p.P = n_pixels;
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
#ifdef UPA_ABLATE
  p.ablate = UPA_OPT(opts, ablate_c1);
#endif
  const int ntw = p.NTn > 8 ? 8 : p.NTn;
  const int gridY = (p.NTn + ntw - 1) / ntw;
  const size_t wbytes = (size_t)p.KTT * ntw * 1024;
  const int tiles = (p.P + 15) / 16;
  static int numCU = 0;
  if (!numCU) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCU <= 0) numCU = 256;
  }
  // pixel tiles per wave and step: enough groups that every SIMD of the chip has a few to pipeline
  const int f_mt = UPA_OPT(opts, c1_mt), f_waves = UPA_OPT(opts, c1_waves), f_wgs = UPA_OPT(opts, c1_wgs);  // tuning / tests
  int mt = tiles >= 8192 ? 4 : (tiles >= 2048 ? 2 : 1);
  // the statistics epilogue keeps 8 ntw running sums per lane on top of the 4 mt ntw accumulators: eight n-tiles leave room for one pixel
  // tile per wave (two spill: 128 -> 128 @80x80 bs 32 45.0 -> 37.0 us, 256 -> 256 @40x40 38.0 -> 27.6, tools/experiments/r05_c1_stats_time.py),
  // six for two
  if (p.epi == 3 && !f_mt) {
    if (ntw > 6 && mt > 1) mt = 1;
    else if (ntw > 4 && mt > 2) mt = 2;
  }
  if (mt * ntw > 32) mt = 32 / ntw;  // accumulator budget: MT * NTW tiles of 4 registers
  if (mt == 3) mt = 2;
  int waves = 8;
  if (f_mt) mt = f_mt;
  if (f_waves) waves = f_waves;
  if (waves != 4) waves = 8;
  auto lds_of = [&]() { return wbytes + (size_t)waves * ring_of(mt) * mt * 1024 + 512; };
  size_t lds = lds_of();
  while (lds > 160 * 1024 && mt > 1) { mt >>= 1; lds = lds_of(); }
  while (lds > 160 * 1024 && waves > 4) { waves >>= 1; lds = lds_of(); }
  if (lds > 160 * 1024) { upa_set_error("conv1x1: weight slice does not fit LDS"); return UPA_EUNSUPPORTED; }
  p.groups = (tiles + mt - 1) / mt;
  int perCU = (int)((160 * 1024) / lds);
  const int waveCap = 32 / waves;  // 8 waves per SIMD
  if (perCU > waveCap) perCU = waveCap;
  if (perCU > 4) perCU = 4;
  if (perCU < 1) perCU = 1;
  int gx = (p.groups + waves - 1) / waves;
  int cap = numCU * perCU / gridY;
  if (f_wgs) cap = f_wgs;
  if (cap < 1) cap = 1;
  if (gx > cap) {
    // equal rounds for every wave: ceil(groups / (rounds * waves)) workgroups
    const int rounds = (p.groups + cap * waves - 1) / (cap * waves);
    gx = (p.groups + rounds * waves - 1) / (rounds * waves);
  }
  if (variant) *variant = (1 << 22) | (waves << 8) | (mt << 4) | ntw;
  if (query_only) return UPA_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)gx, (unsigned)gridY);
  if (p.epi == 3) {  // training forward: rows + per-workgroup channel statistics (row blockIdx.x, channel slice blockIdx.y)
    if ((ntw & 1) || gx > max_rows) return UPA_EUNSUPPORTED;
    p.stats_ld = p.NTn * 16;
    *rows = gx;
    switch (ntw) {
      case 2: return launch_c1_ntw<2, 3>(p, mt, waves, grid, lds, s);
      case 4: return launch_c1_ntw<4, 3>(p, mt, waves, grid, lds, s);
      case 6: return launch_c1_ntw<6, 3>(p, mt, waves, grid, lds, s);
      default: return launch_c1_ntw<8, 3>(p, mt, waves, grid, lds, s);
    }
  }
  if (p.epi) {  // Detect tails: one workgroup row holds every output channel of a pixel
    if (gridY != 1 || (p.epi == 1 && ntw != 4)) { upa_set_error("conv1x1 detect tail: unsupported channel count"); return UPA_EUNSUPPORTED; }
    if (p.epi == 1) return launch_c1_ntw<4, 1>(p, mt, waves, grid, lds, s);
    switch (ntw) {
      case 1: return launch_c1_ntw<1, 2>(p, mt, waves, grid, lds, s);
      case 2: return launch_c1_ntw<2, 2>(p, mt, waves, grid, lds, s);
      case 3: return launch_c1_ntw<3, 2>(p, mt, waves, grid, lds, s);
      case 4: return launch_c1_ntw<4, 2>(p, mt, waves, grid, lds, s);
      case 5: return launch_c1_ntw<5, 2>(p, mt, waves, grid, lds, s);
      case 6: return launch_c1_ntw<6, 2>(p, mt, waves, grid, lds, s);
      case 7: return launch_c1_ntw<7, 2>(p, mt, waves, grid, lds, s);
      default: return launch_c1_ntw<8, 2>(p, mt, waves, grid, lds, s);
    }
  }
  switch (ntw) {
    case 1: return launch_c1_ntw<1>(p, mt, waves, grid, lds, s);
    case 2: return launch_c1_ntw<2>(p, mt, waves, grid, lds, s);
    case 3: return launch_c1_ntw<3>(p, mt, waves, grid, lds, s);
    case 4: return launch_c1_ntw<4>(p, mt, waves, grid, lds, s);
    case 5: return launch_c1_ntw<5>(p, mt, waves, grid, lds, s);
    case 6: return launch_c1_ntw<6>(p, mt, waves, grid, lds, s);
    case 7: return launch_c1_ntw<7>(p, mt, waves, grid, lds, s);
    default: return launch_c1_ntw<8>(p, mt, waves, grid, lds, s);
  }
}

// Last 1x1 conv of a Detect branch with the decode fused on the end (bf16 perf mode): replaces
// upa_conv2d_bias_act(k = 1, act none) + that branch's half of upa_detect_decode.
extern "C" int upa_detect_tail(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, const float* bias,
                               int cout, int kind, int nc, float stride_px, float* y, int a_total, int a0, void* raw, int ldraw,
                               unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && w_packed && y, "detect_tail: null pointer");
  UPA_CHECK_ARG(kind == 1 || kind == 2, "detect_tail: kind must be 1 (box) or 2 (class)");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && a0 >= 0 && a0 + h * w <= a_total, "detect_tail: level does not fit a_total");
  // (the flat pixel -> (image, anchor) split of detect_epi.h must be exact up to the last pixel)
  if (dtype != UPA_BF16 || h * w < 2 || w < 2 || (kind == 1 && cout != 64) || (kind == 2 && (cout < nc || cout > 128)) ||
      !upa_magic_exact((long)n * h * w - 1, h * w) || !upa_magic_exact((long)h * w - 1, w) ||
      !upa_conv1x1_eligible(n, h, w, cin, ldx, cout, raw ? ldraw : cout, false, 1, 1, 0, UPA_ACT_NONE, dtype, opts)) {
    upa_set_error("detect_tail: shape / dtype outside the fused form (bf16, reg_max 16, nc <= 128)");
    return UPA_EUNSUPPORTED;  // the caller runs the conv and upa_detect_decode separately
  }
  C1Params q;
  memset(&q, 0, sizeof(q));
  q.x = (const char*)x; q.y = (char*)raw; q.w = (const char*)w_packed; q.bias = bias;
  q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = raw ? ldraw : cout; q.act = UPA_ACT_NONE;
  q.epi = kind;
  q.de.y = y; q.de.a_total = a_total; q.de.a0 = a0; q.de.HW = h * w; q.de.W = w;
  q.de.magicHW = upa_magic_div(h * w); q.de.magicW = upa_magic_div(w);
  q.de.nc = nc; q.de.stride_px = stride_px;
  if (kind == 2 && best_keys && (long)a_total * nc < (1L << 31)) q.de.best_keys = best_keys;
  q.de.keys_only = (q.de.best_keys && UPA_OPT(opts, keys_only) && raw == nullptr) ? 1 : 0;  // (with raw maps requested the rows are wanted too)
  return upa_conv1x1_launch(q, n * h * w, 0, nullptr, stream, opts);
}

// 1x1 conv whose input is Concat([Upsample(2x nearest)(up), skip]) WITHOUT the upsampled tensor ever being written: the first
// up_c channels of a pixel are read from pixel (y / 2, x / 2) of `up` (n, h / 2, w / 2, up_c), the remaining cin - up_c from the
// concat buffer x (n, h, w, cin) itself, where the skip producer wrote them in place.  Replaces nn.Upsample(None, 2, 'nearest')
// + Concat (yolov8.yaml rows 10-11, 13-14; nn/modules/conv.py Concat.forward) + the following C2f's cv1 (block.py:479).
// Returns UPA_EUNSUPPORTED when the shape is not the streaming kernel's (callers then write the upsample and run the conv).
extern "C" int upa_conv1x1_upcat(const void* x, int n, int h, int w, int cin, int ldx, const void* up, int up_c, int up_ld,
                                 const void* w_packed, const float* bias, void* y, int cout, int ldy, int act, int dtype,
                                 const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && up && w_packed && y && n > 0 && h > 0 && w > 0, "conv1x1_upcat: bad args");
  if (UPA_OPT(opts, no_upcat) || (h & 1) || (w & 1) || up_c <= 0 || up_c % 32 != 0 || up_c >= cin || up_ld % 8 != 0 || ((uintptr_t)up % 16) != 0 ||
      (long)n * (h / 2) * (w / 2) * up_ld * 2 >= (1L << 31) - 4096 || !upa_magic_exact((long)n * h * w - 1, w) ||
      !upa_magic_exact((long)n * h - 1, h) ||
      !upa_conv1x1_eligible(n, h, w, cin, ldx, cout, ldy, false, 1, 1, 0, act, dtype, opts)) {
    upa_set_error("conv1x1_upcat: outside the fused form (bf16 streaming 1x1, even h / w, up_c %% 32 == 0)");
    return UPA_EUNSUPPORTED;
  }
  C1Params q;
  memset(&q, 0, sizeof(q));
  q.x = (const char*)x; q.y = (char*)y; q.w = (const char*)w_packed; q.bias = bias;
  q.Cin = cin; q.ldx = ldx; q.Cout = cout; q.ldy = ldy; q.act = act;
  q.up = (const char*)up; q.upKT = up_c / 32; q.up_ld = up_ld; q.upH = h; q.upW = w;
  q.upMagicW = upa_magic_div(w); q.upMagicH = upa_magic_div(h);
  return upa_conv1x1_launch(q, n * h * w, 0, nullptr, stream, opts);
}
