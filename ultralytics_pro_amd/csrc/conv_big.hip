// Large-tile implicit-GEMM convolution for the MFMA-bound layers (bf16, stride 1, k = 1 | 3, Cout a multiple of 128):
// darknet53 in yolov3-rtdetr, yolov8s, the 256..1024-channel layers of yolov3-tiny.  Same function as conv.hip
// (Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197, BN folded per utils/torch_utils.py:236-266, optional
// Bottleneck residual block.py:668) and the same packed-weight layout; what differs is where the operands come from.
//
// conv_igemm_kernel lets every wave fetch its own weight fragments from L2 (1 KiB per 4..8 MFMAs per wave): at Cin >= 128
// that saturates the CU's 64 B/clk vector-memory path and the kernel sits at 14-17 % of the MFMA peak.  Here a workgroup of
// 8 waves (2 per SIMD) owns 256 output pixels x 128 output channels and BOTH operands are shared through LDS:
//   * B (pixels): the halo tile of a 64-channel chunk, (TH+k-1) x (TW+k-1) pixels x 128 B, staged once per chunk by LDS-DMA
//     (global_load_lds_dwordx4, zero page outside the image / past Cin), XOR-swizzled so ds_read_b128 is conflict-free;
//     every tap reads it at a shifted pixel offset;
//   * A (weights): the 16 KiB slab of one (tap, chunk) - 2 k-tiles x 8 n-tiles in exact fragment order - DMA'd into one of
//     two LDS buffers while the previous tap is multiplied; 16 coalesced 1 KiB wave-instructions per slab per WORKGROUP,
//     i.e. 64 B of weight traffic per MFMA instead of 128-256 B;
//   * wave tile 64 pixels x 64 channels (4 x 4 MFMA tiles, 64 accumulator registers): 8 ds_read_b128 per 16
//     v_mfma_f32_16x16x32_bf16, half the LDS rate at full MFMA issue; <= 128 VGPRs so two workgroups share a CU
//     (4 waves per SIMD) and one workgroup's barrier / DMA wait is covered by the other's MFMAs;
//   * one barrier per tap (32 MFMAs per wave between barriers), the weight DMA of tap t+1 in flight across it;
//   * tile shape chosen per layer on the host (TW need not be a power of two: 20 x 12 for 20x20 maps, 40 x 6 for 40x40), the
//     pixel -> (row, column) split is done once per lane with a multiply-high.
// Epilogue straight from the accumulators (bias, SiLU by v_exp_f32 / v_rcp_f32, bf16 pack, v_permlane16_swap -> 16-byte
// NHWC stores, residual read with the same shape), as conv.hip.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_pipe.h"

typedef __attribute__((address_space(1))) const void* bgptr_t;
typedef __attribute__((address_space(3))) void* blptr_t;

__device__ __attribute__((aligned(16))) unsigned g_big_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
constexpr int BIG_BM = 256;   // pixels per workgroup
constexpr int BIG_NTB = 8;    // n-tiles (16 output channels) per workgroup
constexpr int BIG_WBUF = 2 * BIG_NTB * 1024;  // one (tap, chunk) weight slab: 2 k-tiles x 8 n-tiles x 1 KiB

template <int ACT>
__device__ __forceinline__ float big_act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
}  // namespace

template <int KS>
__global__ __launch_bounds__(512, 4) void conv_big_kernel(const BigParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  const int tilesPerImg = p.tilesX * p.tilesY;
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX;
  const int txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 - p.pad, ix0 = ox0 - p.pad;
  const int ntb0 = blockIdx.y * BIG_NTB;  // first n-tile of the workgroup

  const int haloItems = p.IH * p.IW * 8;  // 16-byte items: 8 per pixel (64 channels)
  const int haloPadded = (haloItems + 63) & ~63;
  char* hal = smem;
  char* wbuf = smem + (size_t)haloPadded * 16;

  // this lane's pixel of each of the wave's 4 m-tiles: tile row / column, halo pixel of tap (0, 0)
  int pl0[4], pty[4], ptx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pp = (wm * 4 + i) * 16 + r;
    int ty = (int)__umulhi((unsigned)pp, p.magicTW);
    int tx = pp - ty * p.TW;
    if (ty >= p.TH) { ty = p.TH; tx = 0; }  // past the tile (TH * TW < 256): computed on halo row TH, never stored
    pty[i] = ty;
    ptx[i] = tx;
    pl0[i] = (ty < p.TH ? ty : 0) * p.IW + tx;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int TAPS = KS * KS;
  const int nChunks = (p.KTT + 1) >> 1;

  auto stage_halo = [&](int c) __attribute__((always_inline)) {
    const int c0 = c * 64;
    for (int base = wave * 64; base < haloPadded; base += 512) {
      const int idx = base + lane;
      const int pix = idx >> 3;
      const int slot = idx & 7;
      const int cg = slot ^ (pix & 7);
      const int py = (int)__umulhi((unsigned)pix, p.magicIW);
      const int px = pix - py * p.IW;
      const int iy = iy0 + py, ix = ix0 + px;
      const int ch = c0 + cg * 8;
      const char* src = reinterpret_cast<const char*>(g_big_zero16);
      if (idx < haloItems && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin)
        src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * 2;
      __builtin_amdgcn_global_load_lds((bgptr_t)src, (blptr_t)(hal + base * 16), 16, 0, 0);
    }
  };
  // weight slab of (tap, chunk c) -> buffer b: fragment f = kt * 8 + j; wave w brings fragments w (kt 0) and w + 8 (kt 1)
  auto stage_w = [&](int c, int tap, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int ktg = c * 2 + kt;
      const int nt = ntb0 + wave;
      const char* src = reinterpret_cast<const char*>(g_big_zero16);
      if (ktg < p.KTT && nt < p.NTn) src = p.w + (((size_t)(tap * p.KTT + ktg) * p.NTn + nt) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((bgptr_t)src, (blptr_t)(wbuf + b * BIG_WBUF + (kt * 8 + wave) * 1024), 16, 0, 0);
    }
  };

  stage_halo(0);
  stage_w(0, 0, 0);
  int buf = 0;
  for (int c = 0; c < nChunks; ++c) {
    int kh = 0, kw = 0;
    for (int tap = 0; tap < TAPS; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of slab (c, tap) (and of the halo) has landed
      __syncthreads();                                  // ... everyone's; everyone is done with the other weight buffer
      if (tap + 1 < TAPS) stage_w(c, tap + 1, buf ^ 1);
      const int tapshift = kh * p.IW + kw;
      const char* wb = wbuf + buf * BIG_WBUF + (wn * 4) * 1024 + lane * 16;
      int paddr[4], pswz[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pl = pl0[i] + tapshift;
        paddr[i] = pl * 128;
        pswz[i] = pl & 7;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        u32x4 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const u32x4*>(wb + (kt * 8 + j) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const u32x4*>(hal + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                                *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
      }
      buf ^= 1;
      if (++kw == KS) { kw = 0; ++kh; }
    }
    if (c + 1 < nChunks) {
      __syncthreads();  // every wave is done with this chunk's halo before it is overwritten
      stage_halo(c + 1);
      stage_w(c + 1, 0, buf);
    }
  }

  // ---- epilogue from the accumulators (as conv.hip): lane (g, r) holds channels 16j + 4g .. + 3 of pixel r of m-tile i;
  // v_permlane16_swap pairs the quads of two neighbouring n-tiles so every lane stores 16 contiguous bytes
  const int cw = (blockIdx.y * BIG_NTB + wn * 4) * 16;  // first channel of this wave
  f32x4 biasv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = cw + j * 16 + g * 4;
    biasv[j] = (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
      const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW;
      const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
      char* yrow = p.y + (pixoff * p.ldy + cw) * 2;
      const char* rrow = p.res ? p.res + (pixoff * p.ldr + cw) * 2 : nullptr;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = big_act<ACT>(acc[i][j][q] + biasv[j][q]);
          v1[q] = big_act<ACT>(acc[i][j + 1][q] + biasv[j + 1][q]);
        }
        const bool ok = pok && cw + cb < p.Cout;
        if (p.res) {
          float x8[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[q]), __float_as_uint(v1[q]), false, false);
            x8[q] = __uint_as_float(sw[0]);
            x8[4 + q] = __uint_as_float(sw[1]);
          }
          if (ok) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(rrow + cb * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x8[2 * q] += __uint_as_float(rv[q] << 16);
              x8[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
            }
            *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{pack_bf16x2(x8[0], x8[1]), pack_bf16x2(x8[2], x8[3]),
                                                            pack_bf16x2(x8[4], x8[5]), pack_bf16x2(x8[6], x8[7])};
          }
        } else {
          auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if (ok) *reinterpret_cast<u32x4*>(yrow + cb * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      }
    }
  };
  if (p.act == UPA_ACT_SILU) epilogue(std::integral_constant<int, UPA_ACT_SILU>{});
  else if (p.act == UPA_ACT_RELU) epilogue(std::integral_constant<int, UPA_ACT_RELU>{});
  else epilogue(std::integral_constant<int, UPA_ACT_NONE>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int big_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
}  // namespace

// Dispatch mode of the large-tile kernel: 0 = never, 1 = by the size rule of upa_conv_big_eligible (default), 2 = every
// shape the kernel can run (parity tests, tools/bench_conv.py).  Initialised from UPA_CONV_BIG; mode < 0 only queries.
extern "C" int upa_conv_big_mode(int mode) {
  static int cur = big_env("UPA_CONV_BIG", 1);
  const int prev = cur;
  if (mode >= 0) cur = mode;
  return prev;
}

bool upa_conv_big_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad,
                           int act, int dtype) {
  const int mode = upa_conv_big_mode(-1);
  if (mode == 0) return false;
  if (dtype != UPA_BF16 || stride != 1 || !(k == 1 || k == 3) || pad != k / 2) return false;
  if (cin % 8 != 0 || ldx % 8 != 0 || cout % 8 != 0 || ldy % 8 != 0 || ldr % 8 != 0) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE && act != UPA_ACT_RELU) return false;
  if (mode == 2) return cout >= 64;
  // MFMA-bound layers only: both operands wide enough that sharing the weights through LDS pays, output channels a whole
  // number of 128-channel workgroup columns
  static const int min_cin = big_env("UPA_CONV_BIG_MIN_CIN", 128);
  if (cin < min_cin || cout % 128 != 0) return false;
  const long px = (long)n * h * w;
  return px >= 2048;
}

int upa_conv_big_launch(BigParams p, int query_only, int* variant, void* stream) {
  if (variant) *variant = (1 << 23);
  if (query_only) return UPA_OK;
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
  if (p.KS == 1) {  // pointwise: an NHWC view has one uniform pixel stride - flatten (n, h, w) into one row
    const long P = (long)p.N * p.H * p.W;
    p.N = 1; p.H = 1; p.W = (int)P; p.OH = 1; p.OW = (int)P;
    p.TH = 1; p.TW = BIG_BM;
  } else {
    // tile shape: TH x TW <= 256 pixels; fewest tiles per image first (least padding waste), then the smallest halo
    long best = -1;
    int bTH = 16, bTW = 16;
    for (int tw = 4; tw <= 128 && tw <= ((p.OW + 3) & ~3); ++tw) {
      int th = BIG_BM / tw;
      if (th > p.OH) th = p.OH;
      if (th < 1) continue;
      const long tiles = (long)cdiv(p.OW, tw) * cdiv(p.OH, th);
      const long halo = (long)(th + 2) * (tw + 2);
      const long cost = tiles * 4096 + halo;
      if (best < 0 || cost < best) { best = cost; bTH = th; bTW = tw; }
    }
    p.TH = bTH; p.TW = bTW;
  }
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  p.IH = p.TH + p.KS - 1;
  p.IW = p.TW + p.KS - 1;
  p.magicTW = (unsigned)((0x100000000ULL + p.TW - 1) / p.TW);
  p.magicIW = (unsigned)((0x100000000ULL + p.IW - 1) / p.IW);
  // pixels past the tile read halo row TH (allocated: IH >= TH + 1 for k = 3; one extra row for k = 1)
  const int rows = p.KS == 1 ? 2 : p.IH;
  const size_t halo = (((size_t)rows * p.IW * 8 + 63) & ~(size_t)63) * 16;
  const size_t lds = halo + 2 * (size_t)BIG_WBUF + 256;
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N), (unsigned)cdiv(p.NTn, BIG_NTB));
  hipStream_t s = (hipStream_t)stream;
  if (p.KS == 1) {
    auto kern = conv_big_kernel<1>;
    if (hipError_t e = upa_full_lds<conv_big_kernel<1>>(); e != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  } else {
    auto kern = conv_big_kernel<3>;
    if (hipError_t e = upa_full_lds<conv_big_kernel<3>>(); e != hipSuccess) return UPA_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  }
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
