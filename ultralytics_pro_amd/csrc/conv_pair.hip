// Bottleneck as ONE kernel: y = [x +] act(conv3x3_2(act(conv3x3_1(x) + b1)) + b2) with C -> C -> C channels (the C2f inner
// blocks: ultralytics/nn/modules/block.py:644-668 with k = (3, 3), e = 1.0 as built by C2f, block.py:475; BN folded per
// utils/torch_utils.py:236-266), bf16, C = 32 (dispatched) or 64 (built and tested; slower than two launches at 40x40, see the host side).  The intermediate tensor never reaches HBM: it lives as a
// (TH+2) x (TW+2) pixel tile in LDS, recomputed on a one-pixel ring around the output tile.
//
// Structure = conv_big.hip twice inside one workgroup (8 waves, 256 pixel slots, every output channel in the wave: WM 8 x WN 1):
//   stage 1: input halo (TH+4) x (TW+4) x C staged by LDS-DMA (zero page outside the image); the nine weight slabs of conv 1
//            stream through two LDS buffers, one tap ahead; result on the (TH+2) x (TW+2) <= 256 pixels of the mid tile:
//            bias + SiLU -> bf16 -> LDS, ZERO where the mid pixel lies outside the image (the zero padding conv 2 sees);
//   stage 2: the same loop reading the mid tile, weights of conv 2 (their first slab prefetched under stage 1's last tap);
//            epilogue bias + SiLU + residual read from the input halo STILL IN LDS (no second global read of x) ->
//            16-byte NHWC stores.
// Cost: conv 1 is evaluated on (TH+2)(TW+2) / (TH*TW) pixels (1.31x at 14 x 14); saved: one launch, the intermediate's write
// and read, the residual read, and - the point for the latency-bound 40x40 / 80x80 layers - one whole launch -> DMA ->
// MFMA -> store dependency chain per Bottleneck.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* pgptr_t;
typedef __attribute__((address_space(3))) void* plptr_t;

__device__ __attribute__((aligned(16))) unsigned g_pair_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
template <int ACT>
__device__ __forceinline__ float pair_act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
}  // namespace

// CK = k-tiles (32 channels) of C: 1 (C = 32) or 2 (C = 64); NT = C / 16 n-tiles, all in every wave
// CV2 (CK = 1 only): the Bottleneck is the single one of a C2f(.., 64, n = 1) and the C2f's cv2 follows at once - its input
// cat(y0, y1, b) is three 32-channel k-steps: y0 straight from global memory in B-fragment order, y1 = this kernel's own input
// (the halo tile's centre, still in LDS), b = the bf16-packed accumulators (the D layout of two 16-channel tiles IS an MFMA B
// operand in the k order of upa_pack_tail_weight, see conv_big.hip).  b is never written; one launch and 2 x 13 MB less.
// CKM < CK (round 4): the e = 0.5 Bottleneck of the darknet backbones, C -> C / 2 -> C (cfg/models/v3/Detect/yolov3-rtdetr.yaml rows 2, 4:
// Bottleneck(64) at 320 x 320, block.py:644-668 with the default e): CKM k-tiles in the mid tile, CK in the halo and the output; a tap's weight
// slab has CK * CKM * 2 fragments in either stage.
template <int CK, bool RES, bool CV2 = false, int CKM = CK>
__global__ __launch_bounds__(512, 4) void conv_pair_kernel(const PairParams p) {
  static_assert(!CV2 || (CK == 1 && CKM == 1), "the cv2 tail is built for 32-channel Bottlenecks");
  constexpr int NT = CK * 2;               // n-tiles of the output (stage 2)
  constexpr int NTM = CKM * 2;             // n-tiles of the mid tile (stage 1)
  constexpr int NTX = NT > NTM ? NT : NTM;
  constexpr int G16 = CK * 4;              // 16-byte groups per pixel of the halo image
  constexpr int PB = G16 * 16;             // bytes per pixel of the halo image
  constexpr int G16M = CKM * 4, PBM = G16M * 16;   // ... of the mid tile
  constexpr int WBUF = CK * NTM * 1024;    // one tap's weight slab (= CKM * NT fragments in stage 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX;
  const int txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int IH = p.TH + 4, IW = p.TW + 4, MH = p.TH + 2, MW = p.TW + 2;
  const int IWp = p.IWp, MWp = p.MWp;  // LDS pitches (>= IW, MW): chosen so that m-tiles straddling tile rows stay conflict-free

  auto swz = [](int pix) __attribute__((always_inline)) { return (CK == 1 ? (pix >> 1) : pix) & (G16 - 1); };        // halo image
  auto swzm = [](int pix) __attribute__((always_inline)) { return (CKM == 1 ? (pix >> 1) : pix) & (G16M - 1); };  // mid tile

  const int haloItems = IH * IWp * G16;
  const int haloPadded = (haloItems + 63) & ~63;
  const int midBytes = ((MH * MWp * PBM) + 1023) & ~1023;
  char* hal = smem;
  char* mid = smem + (size_t)haloPadded * 16;
  char* wbuf = mid + midBytes;

  // ---- input halo: rows oy0-2 .. oy0+TH+1, columns ox0-2 .. ox0+TW+1, every channel
  for (int base = wave * 64; base < haloPadded; base += 512) {
    const int idx = base + lane;
    const int pix = idx / G16;          // G16 is a power of two
    const int slot = idx & (G16 - 1);
    const int cg = slot ^ swz(pix);
    const int py = (int)__umulhi((unsigned)pix, p.magicIW);
    const int px = pix - py * IWp;
    const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
    const char* src = reinterpret_cast<const char*>(g_pair_zero16);
    if (idx < haloItems && px < IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + cg * 8) * 2;
    __builtin_amdgcn_global_load_lds((pgptr_t)src, (plptr_t)(hal + base * 16), 16, 0, 0);
  }
  // weight slab of one tap: CK * NT fragments of 1 KiB, wave w brings fragments w, w + 8, ...
  auto stage_w = [&](const char* w, int tap, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int f0 = 0; f0 < CK * NTM; f0 += 8) {
      const int f = f0 + wave;
      if (f < CK * NTM)
        __builtin_amdgcn_global_load_lds((pgptr_t)(w + (((size_t)tap * CK * NTM + f) * 64 + lane) * 16),
                                         (plptr_t)(wbuf + b * WBUF + f * 1024), 16, 0, 0);
    }
  };
  stage_w(p.w1, 0, 0);

  // one conv over an LDS image: this wave's two m-tiles (pixel slots wave*32 .. +31) x all NT n-tiles
  f32x4 acc[2][NTX];
  // KT_ k-tiles of the image (CKI_ = the image's k-tiles per pixel record: its swizzle), NT_ n-tiles
  auto conv_stage = [&](auto kt_c, auto nt_c, const char* img, int imgW, const int (&pl0)[2], const char* wcur, const char* wnext,
                        int& buf) __attribute__((always_inline)) {
    constexpr int KT_ = decltype(kt_c)::value, NT_ = decltype(nt_c)::value;
    constexpr int PB_ = KT_ * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kh = 0, kw = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tap + 1 < 9) stage_w(wcur, tap + 1, buf ^ 1);
      else if (wnext) stage_w(wnext, 0, buf ^ 1);  // first slab of the next conv rides under this conv's last tap
      const int tapshift = kh * imgW + kw;
      const char* wb = wbuf + buf * WBUF + lane * 16;
      int paddr[2], pswz[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pl = pl0[i] + tapshift;
        paddr[i] = pl * PB_;
        pswz[i] = (KT_ == 1 ? (pl >> 1) : pl) & (KT_ * 4 - 1);
      }
#pragma unroll
      for (int kt = 0; kt < KT_; ++kt) {
        u32x4 a[NT_], b[2];
#pragma unroll
        for (int j = 0; j < NT_; ++j) a[j] = *reinterpret_cast<const u32x4*>(wb + (kt * NT_ + j) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const u32x4*>(img + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT_; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                                *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
      }
      buf ^= 1;
      if (++kw == 3) { kw = 0; ++kh; }
    }
  };

  // ---- stage 1: conv 1 on the mid region (MH x MW pixels, pixel slot pp -> (my, mx))
  int my[2], mx[2], pl1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = (wave * 2 + i) * 16 + r;
    int y_ = (int)__umulhi((unsigned)pp, p.magicMW);
    int x_ = pp - y_ * MW;
    if (y_ >= MH) { y_ = MH; x_ = 0; }
    my[i] = y_;
    mx[i] = x_;
    pl1[i] = y_ < MH ? y_ * IWp + x_ : 0;
  }
  int buf = 0;
  conv_stage(std::integral_constant<int, CK>{}, std::integral_constant<int, NTM>{}, hal, IWp, pl1, p.w1, p.w2, buf);
  {
    f32x4 bv[NTM];
#pragma unroll
    for (int j = 0; j < NTM; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.b1 + j * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (my[i] >= MH) continue;
      const int gy = oy0 - 1 + my[i], gx = ox0 - 1 + mx[i];
      const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int mp = my[i] * MWp + mx[i];
      char* row = mid + mp * PBM;
      const int sw = swzm(mp);
#pragma unroll
      for (int j = 0; j < NTM; ++j) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = inside ? pair_act<UPA_ACT_SILU>(acc[i][j][q] + bv[j][q]) : 0.f;
        // channels 16j + 4g .. + 3: 16-byte group 2j + (g >> 1), half (g & 1)
        const int cg = 2 * j + (g >> 1);
        *reinterpret_cast<u32x2*>(row + ((cg ^ sw) << 4) + (g & 1) * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
  // (the barrier at the head of stage 2's first tap orders these LDS writes before any read of the mid tile)

  // ---- stage 2: conv 2 on the output tile (TH x TW pixels) from the mid tile
  int ty[2], tx[2], pl2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = (wave * 2 + i) * 16 + r;
    int y_ = (int)__umulhi((unsigned)pp, p.magicTW);
    int x_ = pp - y_ * p.TW;
    if (y_ >= p.TH) { y_ = p.TH; x_ = 0; }
    ty[i] = y_;
    tx[i] = x_;
    pl2[i] = y_ < p.TH ? y_ * MWp + x_ : 0;
  }
  conv_stage(std::integral_constant<int, CKM>{}, std::integral_constant<int, NT>{}, mid, MWp, pl2, p.w2, nullptr, buf);

  f32x4 bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.b2 + j * 16 + g * 4);
  if constexpr (CV2) {
    f32x4 bcv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bcv[j] = *reinterpret_cast<const f32x4*>(p.bc + j * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = oy0 + ty[i], ox = ox0 + tx[i];
      const bool pok = ty[i] < p.TH && oy < p.OH && ox < p.OW;
      const size_t pixoff = ((size_t)n * p.OH + (pok ? oy : 0)) * p.OW + (pok ? ox : 0);
      const int hp = ty[i] < p.TH ? (ty[i] + 2) * IWp + tx[i] + 2 : 0;  // this output pixel in the input halo
      // b = SiLU(conv2 + b2) (+ y1): two 16-channel tiles -> one B operand (k order of upa_pack_tail_weight)
      float v0[4], v1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] = pair_act<UPA_ACT_SILU>(acc[i][0][q] + bv[0][q]);
        v1[q] = pair_act<UPA_ACT_SILU>(acc[i][1][q] + bv[1][q]);
      }
      if constexpr (RES) {
        const char* xrow = hal + hp * PB;
        const int xsw = swz(hp);
        const u32x2 r0 = *reinterpret_cast<const u32x2*>(xrow + (((g >> 1) ^ xsw) << 4) + (g & 1) * 8);
        const u32x2 r1 = *reinterpret_cast<const u32x2*>(xrow + (((2 + (g >> 1)) ^ xsw) << 4) + (g & 1) * 8);
        v0[0] += __uint_as_float(r0[0] << 16); v0[1] += __uint_as_float(r0[0] & 0xFFFF0000u);
        v0[2] += __uint_as_float(r0[1] << 16); v0[3] += __uint_as_float(r0[1] & 0xFFFF0000u);
        v1[0] += __uint_as_float(r1[0] << 16); v1[1] += __uint_as_float(r1[0] & 0xFFFF0000u);
        v1[2] += __uint_as_float(r1[1] << 16); v1[3] += __uint_as_float(r1[1] & 0xFFFF0000u);
      }
      const u32x4 opb = u32x4{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
      const u32x4 opy1 = *reinterpret_cast<const u32x4*>(hal + hp * PB + ((g ^ swz(hp)) << 4));
      const u32x4 opy0 = *reinterpret_cast<const u32x4*>(p.y0 + (pixoff * (size_t)p.ldx) * 2 + g * 16);
      f32x4 o[4] = {bcv[0], bcv[1], bcv[2], bcv[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(p.wc_std + ((size_t)(0 * 4 + j) * 64 + lane) * 16);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(p.wc_std + ((size_t)(1 * 4 + j) * 64 + lane) * 16);
        const u32x4 a2 = *reinterpret_cast<const u32x4*>(p.wc_b + ((size_t)j * 64 + lane) * 16);
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a0), *reinterpret_cast<const bf16x8*>(&opy0), o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a1), *reinterpret_cast<const bf16x8*>(&opy1), o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a2), *reinterpret_cast<const bf16x8*>(&opb), o[j], 0, 0, 0);
      }
      char* yrow = p.out + pixoff * (size_t)p.ldout * 2;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        float w0[4], w1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w0[q] = pair_act<UPA_ACT_SILU>(o[j][q]);
          w1[q] = pair_act<UPA_ACT_SILU>(o[j + 1][q]);
        }
        auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(w0[0], w0[1]), pack_bf16x2(w1[0], w1[1]), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(w0[2], w0[3]), pack_bf16x2(w1[2], w1[3]), false, false);
        const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
        if (pok) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = oy0 + ty[i], ox = ox0 + tx[i];
    const bool pok = ty[i] < p.TH && oy < p.OH && ox < p.OW;
    char* yrow = p.y + (((size_t)n * p.OH + oy) * p.OW + ox) * (size_t)p.ldy * 2;
    const int hp = (ty[i] + 2) * IWp + tx[i] + 2;  // this output pixel in the input halo (residual)
    const char* xrow = hal + (ty[i] < p.TH ? hp : 0) * PB;
    const int xsw = swz(ty[i] < p.TH ? hp : 0);
#pragma unroll
    for (int j = 0; j < NT; j += 2) {
      const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
      float v0[4], v1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] = pair_act<UPA_ACT_SILU>(acc[i][j][q] + bv[j][q]);
        v1[q] = pair_act<UPA_ACT_SILU>(acc[i][j + 1][q] + bv[j + 1][q]);
      }
      if constexpr (RES) {  // x + ...: the lane's own 4 channels of tiles j and j + 1, from the halo tile in LDS
        const u32x2 r0 = *reinterpret_cast<const u32x2*>(xrow + (((2 * j + (g >> 1)) ^ xsw) << 4) + (g & 1) * 8);
        const u32x2 r1 = *reinterpret_cast<const u32x2*>(xrow + (((2 * (j + 1) + (g >> 1)) ^ xsw) << 4) + (g & 1) * 8);
        v0[0] += __uint_as_float(r0[0] << 16); v0[1] += __uint_as_float(r0[0] & 0xFFFF0000u);
        v0[2] += __uint_as_float(r0[1] << 16); v0[3] += __uint_as_float(r0[1] & 0xFFFF0000u);
        v1[0] += __uint_as_float(r1[0] << 16); v1[1] += __uint_as_float(r1[0] & 0xFFFF0000u);
        v1[2] += __uint_as_float(r1[1] << 16); v1[3] += __uint_as_float(r1[1] & 0xFFFF0000u);
      }
      auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
      if (pok) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <int CK, int CKM>
int pair_launch_e(const PairParams& p, size_t lds, bool res, hipStream_t s) {
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N));
  if (res) {
    auto kern = conv_pair_kernel<CK, true, false, CKM>;
    if (upa_full_lds<(conv_pair_kernel<CK, true, false, CKM>)>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  } else {
    auto kern = conv_pair_kernel<CK, false, false, CKM>;
    if (upa_full_lds<(conv_pair_kernel<CK, false, false, CKM>)>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int CK>
int pair_launch(const PairParams& p, size_t lds, bool res, hipStream_t s) {
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N));
  if constexpr (CK == 1) {
    if (p.out) {  // Bottleneck + the C2f's cv2
      if (res) {
        if (upa_full_lds<conv_pair_kernel<1, true, true>>() != hipSuccess) return UPA_ELAUNCH;
        hipLaunchKernelGGL((conv_pair_kernel<1, true, true>), grid, dim3(512), lds, s, p);
      } else {
        if (upa_full_lds<conv_pair_kernel<1, false, true>>() != hipSuccess) return UPA_ELAUNCH;
        hipLaunchKernelGGL((conv_pair_kernel<1, false, true>), grid, dim3(512), lds, s, p);
      }
      UPA_LAUNCH_CHECK();
      return UPA_OK;
    }
  }
  if (res) {
    auto kern = conv_pair_kernel<CK, true>;
    if (upa_full_lds<conv_pair_kernel<CK, true>>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  } else {
    auto kern = conv_pair_kernel<CK, false>;
    if (upa_full_lds<conv_pair_kernel<CK, false>>() != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
}  // namespace

static int pair_impl(const void* x, int n, int h, int w, int c, int cmid, int ldx, const void* w1_packed, const float* b1,
                     const void* w2_packed, const float* b2, void* y, int ldy, int residual, int act, int dtype,
                     const PairParams* cv2, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(x && w1_packed && b1 && w2_packed && b2 && y && n > 0 && h > 0 && w > 0, "bottleneck_pair: bad args");
  // upa_opts.pair: 0 = C = 32 only (the default), 1 = never, 2 = both widths, 3 = C = 64 only.  Measured on MI355X (yolov8n bs 32, four steps in
  // flight): C = 32 pairs at 80x80 28.7 us against 16.4 + 19.1 us as two launches, step 0.800 -> 0.773 ms; C = 64 pairs at
  // 40x40 30.9-32.2 us against 11.5 + 12.2 us (288 one-per-CU workgroups = two rounds; 12 x 12 / 10 x 10 tiles no better),
  // step 0.773 -> 0.790 ms - the 64-channel form stays available but is not dispatched
  const int mode = UPA_OPT(opts, pair);
  // C -> C / 2 -> C (64 -> 32 -> 64): dispatched unless pair = 1.  Measured on MI355X (yolov3-rtdetr bs 16, model.2 at 320 x 320): see
  // DESIGN section 4
  const bool half = c == 64 && cmid == 32;
  const bool equal = cmid == c && (c == 32 || c == 64);
  if (mode == 1 || (equal && ((mode == 0 && c == 64) || (mode == 3 && c == 32))) || dtype != UPA_BF16 || act != UPA_ACT_SILU ||
      !(equal || half) || ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 || (half && cv2)) {
    upa_set_error("bottleneck_pair: outside the fused form (bf16, SiLU, C = 32 | 64, or 64 -> 32 -> 64)");
    return UPA_EUNSUPPORTED;  // the caller runs the two convolutions separately
  }
  PairParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.y = (char*)y; p.w1 = (const char*)w1_packed; p.w2 = (const char*)w2_packed; p.b1 = b1; p.b2 = b2;
  p.N = n; p.H = h; p.W = w; p.OH = h; p.OW = w; p.ldx = ldx; p.ldy = ldy;
  if (cv2) { p.y0 = cv2->y0; p.wc_std = cv2->wc_std; p.wc_b = cv2->wc_b; p.bc = cv2->bc; p.out = cv2->out; p.ldout = cv2->ldout; }
  const int ck = c / 32, ckm = cmid / 32;
  const int pb = ckm * 64;   // bytes per pixel of the mid tile
  // output tile TH x TW with (TH+2)(TW+2) <= 256 mid pixels: fewest tiles per image, then the squarest
  const int fth64 = UPA_OPT(opts, pair_tile64), fth32 = UPA_OPT(opts, pair_tile32);  // square tile edge per width
  const int fth = c == 64 ? fth64 : fth32, ftw = fth;
  long best = -1;
  for (int tw = 2; tw <= 62 && tw <= ((w + 1) & ~1); ++tw) {
    int th = 256 / (tw + 2) - 2;
    if (th > h) th = h;
    if (th < 1) continue;
    const long tiles = (long)cdiv(w, tw) * cdiv(h, th);
    const long cost = tiles * 65536 + (long)(th + 4) * (tw + 4);
    if (best < 0 || cost < best) { best = cost; p.TH = th; p.TW = tw; }
  }
  if (fth > 0 && ftw > 0 && (fth + 2) * (ftw + 2) <= 256) { p.TH = fth; p.TW = ftw; }
  p.tilesX = cdiv(w, p.TW);
  p.tilesY = cdiv(h, p.TH);
  const int IH = p.TH + 4, IW = p.TW + 4, MH = p.TH + 2, MW = p.TW + 2;
  // pitches: stage 1 enumerates the MH x MW mid pixels over the halo image, stage 2 the TH x TW outputs over the mid tile
  p.IWp = upa_lds_pick_pitch(IW, MW, MH * MW, 1);
  p.MWp = upa_lds_pick_pitch(MW, p.TW, p.TH * p.TW, 1);
  p.magicIW = (unsigned)((0x100000000ULL + p.IWp - 1) / p.IWp);
  p.magicMW = (unsigned)((0x100000000ULL + MW - 1) / MW);
  p.magicTW = (unsigned)((0x100000000ULL + p.TW - 1) / p.TW);
  const size_t halo = (((size_t)IH * p.IWp * (ck * 4) + 63) & ~(size_t)63) * 16;
  const size_t mid = (((size_t)MH * p.MWp * pb) + 1023) & ~(size_t)1023;
  const size_t lds = halo + mid + 2 * (size_t)(ck * ckm * 2 * 1024) + 256;
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (half) return pair_launch_e<2, 1>(p, lds, residual != 0, s);
  return ck == 1 ? pair_launch<1>(p, lds, residual != 0, s) : pair_launch<2>(p, lds, residual != 0, s);
}

extern "C" int upa_bottleneck_pair(const void* x, int n, int h, int w, int c, int ldx, const void* w1_packed, const float* b1,
                                   const void* w2_packed, const float* b2, void* y, int ldy, int residual, int act, int dtype,
                                   const upa_opts* opts, void* stream) {
  return pair_impl(x, n, h, w, c, c, ldx, w1_packed, b1, w2_packed, b2, y, ldy, residual, act, dtype, nullptr, opts, stream);
}

// ... with the hidden width given: cmid = c (the C2f inner blocks above) or c / 2 for c = 64 (the e = 0.5 Bottleneck of the darknet
// backbones, block.py:644-668).  w1_packed: upa_pack_conv_weight(c -> cmid, k = 3), w2_packed: (cmid -> c, k = 3).
extern "C" int upa_bottleneck_pair_e(const void* x, int n, int h, int w, int c, int cmid, int ldx, const void* w1_packed, const float* b1,
                                     const void* w2_packed, const float* b2, void* y, int ldy, int residual, int act, int dtype,
                                     const upa_opts* opts, void* stream) {
  return pair_impl(x, n, h, w, c, cmid, ldx, w1_packed, b1, w2_packed, b2, y, ldy, residual, act, dtype, nullptr, opts, stream);
}

// C2f(.., 64, n = 1) with a 32-channel Bottleneck: Bottleneck (both 3x3 convs [+ shortcut]) AND the C2f's cv2 in one launch.
// x = the y1 slice (channels [32, 64)) of the C2f concat buffer, y0 = its y0 slice (channels [0, 32)), same pixel stride ldx;
// wc_std = cv2's columns [0, 64) packed by upa_pack_conv_weight(64 -> 64, k = 1), wc_b = columns [64, 96) packed by
// upa_pack_tail_weight(64, 32), bc = cv2's bias; out = (n, h, w, 64) view.  nn/modules/block.py:457-488, 644-668.
extern "C" int upa_bottleneck_pair_cv2(const void* x, const void* y0, int n, int h, int w, int ldx, const void* w1_packed,
                                       const float* b1, const void* w2_packed, const float* b2, int residual,
                                       const void* wc_std, const void* wc_b, const float* bc, void* out, int ldout, int act,
                                       int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(y0 && wc_std && wc_b && bc && out, "bottleneck_pair_cv2: null pointer");
  if (UPA_OPT(opts, no_pair_cv2) || ldout % 8 != 0 || ((uintptr_t)out % 16) != 0 || ((uintptr_t)y0 % 16) != 0) {
    upa_set_error("bottleneck_pair_cv2: outside the fused form");
    return UPA_EUNSUPPORTED;
  }
  PairParams cv2;
  memset(&cv2, 0, sizeof(cv2));
  cv2.y0 = (const char*)y0; cv2.wc_std = (const char*)wc_std; cv2.wc_b = (const char*)wc_b; cv2.bc = bc;
  cv2.out = (char*)out; cv2.ldout = ldout;
  // (the cv2 form is a 32-channel pair whatever upa_opts.pair says about the plain Bottleneck dispatch)
  upa_opts o2;
  memset(&o2, 0, sizeof(o2));
  o2.size = sizeof(o2);
  o2.pair_tile32 = UPA_OPT(opts, pair_tile32);
  return pair_impl(x, n, h, w, 32, 32, ldx, w1_packed, b1, w2_packed, b2, out, ldout, residual, act, dtype, &cv2, &o2, stream);
}
