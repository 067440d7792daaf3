// Large-tile implicit-GEMM convolution with BOTH operands shared through LDS (bf16, k = 1 | 3, stride 1 | 2): the MFMA-bound
// layers - darknet53 in yolov3-rtdetr, yolov8s, the 256..1024-channel layers of yolov3-tiny - and the 80-channel class branch
// of Detect.  Same function as conv.hip (Conv.forward_fuse, ultralytics/nn/modules/conv.py:188-197, BN folded per
// utils/torch_utils.py:236-266, optional Bottleneck residual block.py:668) and the same packed-weight layout; what differs
// is where the operands come from.
//
// conv_igemm_kernel lets every wave fetch its own weight fragments from L2 (1 KiB per 4..8 MFMAs per wave): at Cin >= 128
// that saturates the CU's 64 B/clk vector-memory path and the kernel sits at 14-17 % of the MFMA peak.  Here a workgroup of
// 8 waves (2 per SIMD) owns BM = 256 (or 128) output pixels x NTB * 16 output channels:
//   * B (pixels): the halo tile of a 64-channel chunk, ((TH-1)s+k) x ((TW-1)s+k) pixels x 128 B, staged once per chunk by
//     LDS-DMA (global_load_lds_dwordx4, zero page outside the image / past Cin), XOR-swizzled by the pixel index; every tap
//     reads it at a shifted pixel offset.  The image's pixel pitch IWp is chosen on the host (upa_lds_pick_pitch) so that
//     m-tiles that straddle tile rows (TW = 20, 40: not powers of two) still read 8 distinct pixel residues per lane set =
//     conflict-free ds_read_b128; at stride 2 the columns are stored de-interleaved (even | odd) so that the pixels of an
//     m-tile are consecutive again instead of every other one (which used half the banks: 45 % conflict cycles in round 2);
//   * A (weights): the slab of one (tap, chunk) - 2 k-tiles x NTB n-tiles of 1 KiB in exact fragment order - DMA'd into one
//     of two LDS buffers while the previous tap is multiplied: coalesced 1 KiB wave-instructions, once per WORKGROUP,
//     i.e. 64 B of weight traffic per MFMA instead of 128-256 B;
//   * wave tile MT x NT MFMA tiles (4 x 4 = 64 pixels x 64 channels in the main variant: 8 ds_read_b128 per 16
//     v_mfma_f32_16x16x32_bf16, half the LDS rate at full MFMA issue); <= 128 VGPRs, so two workgroups share a CU (4 waves
//     per SIMD) whenever their LDS fits and one workgroup's barrier / DMA wait is covered by the other's MFMAs;
//   * one barrier per tap (2 * MT * NT MFMAs per wave between barriers), the weight DMA of tap t+1 in flight across it;
//   * tile shape chosen per layer on the host (TW need not be a power of two: 20 x 12 for 20x20 maps, 40 x 6 for 40x40), the
//     pixel -> (row, column) split is done once per lane with a multiply-high.
// Variants <WM, WN, MT, NT> (WM * WN = 8 waves): 4,2,4,4 = 256 px x 128 ch; 4,2,2,4 = 128 px x 128 ch (few-pixel layers: twice
// the workgroups); 4,2,4,3 / 4,2,2,3 = x 96 ch (Cout 96, or 80 off the 3x3 stride-1 form); 8,1,2,5 / 8,1,1,5 = x 80 ch (the Detect class
// branch: five tiles, nothing padded); 8,1,2,4 = 256 px x 64 ch; 4,2,2,2 = 128 px x 64 ch.
// Epilogue straight from the accumulators (bias, SiLU by v_exp_f32 / v_rcp_f32, bf16 pack, v_permlane16_swap -> 16-byte
// NHWC stores, residual read with the same shape), as conv.hip.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "common.h"
#include "conv_pipe.h"
UPA_STAMP_DEFINE(conv_big)

typedef __attribute__((address_space(1))) const void* bgptr_t;
typedef __attribute__((address_space(3))) void* blptr_t;

__device__ __attribute__((aligned(16))) unsigned g_big_zero16[4] = {0u, 0u, 0u, 0u};

namespace {
template <int ACT>
__device__ __forceinline__ float big_act(float v) {
  if constexpr (ACT == UPA_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else if constexpr (ACT == UPA_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}
}  // namespace

// TAIL != 0 (Detect branches, WN = 1 so that a wave holds every channel of its pixels): the SiLU'd result of this 3x3 conv is
// not stored but fed, from the accumulators, into the branch's final 1x1 conv and that conv's half of the decode - see the
// tail section below.  TAIL 1 = box branch (DFL + dist2bbox), 2 = class branch (sigmoid).
// TAIL 3 (training forward, act = none, no residual): besides storing z the workgroup leaves the sum and the sum of squares of the
// bf16-ROUNDED values it stored, per output channel, in row blockIdx.x of p.stats - the first stage of BatchNorm's batch statistics
// (conv.py:177-186 in train mode) without a pass over z; a combine kernel adds the rows in a fixed order (train.hip).
// TAIL 4 (training: data gradient of a 3x3 stride-2 convolution as four 2x2 phase correlations stacked along the output channels, train.hip
// dgrad_s2_phase_weights): the epilogue stores phase (py, px)'s value of pixel (i, j) straight to dx[2 (i - 1) + py][2 (j - 1) + px] (+= with
// p.res = dx) - the (n, oh + 1, ow + 1, 4 cin) phase tensor and the interleaving pass over it are gone.
// The kernel body: `bid0` = the workgroup's tile index within ITS problem (blockIdx.x for a single launch; a paired launch -
// conv_big_pair_kernel below - runs two problems of the same instantiation in one grid).
template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
__device__ __forceinline__ void conv_big_body(const BigParams& p, const int bid0) {
  static_assert(WM * WN == 8, "8 waves per workgroup");
  static_assert(TAIL == 0 || TAIL == 3 || TAIL == 4 || WN == 1, "a tail needs every channel of a pixel in one wave");
  static_assert(TAIL != 1 || (NT % 2) == 0, "box branch: 64 channels");
  static_assert(TAIL != 3 || (NT % 2) == 0, "the statistics epilogue pairs n-tiles");
  static_assert(TAIL != 4 || ((NT % 2) == 0 && KS == 2 && STRIDE == 1), "the interleaving epilogue: 2 x 2 phase kernels, paired n-tiles");
  constexpr int NTB = WN * NT;            // n-tiles per workgroup
  constexpr int WBUF = 2 * NTB * 1024;    // one (tap, chunk) weight slab: 2 k-tiles x NTB n-tiles x 1 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;

  const int tilesPerImg = p.tilesX * p.tilesY;
  // XCD-aware tile order (common.h: upa_xcd_tile): within one problem and one row of n-tile blocks the linear workgroup index is
  // bid0 plus a constant, so "bid0 & 7" names the XCD up to a rotation - each XCD still walks one contiguous range of tiles
  int bid = p.no_xcd ? bid0 : upa_xcd_tile(bid0, tilesPerImg * p.N);
  const int n = bid / tilesPerImg;
  bid -= n * tilesPerImg;
  const int tyi = bid / p.tilesX;
  const int txi = bid - tyi * p.tilesX;
  const int oy0 = tyi * p.TH, ox0 = txi * p.TW;
  const int iy0 = oy0 * STRIDE - p.pad, ix0 = ox0 * STRIDE - p.pad;
  const int ntb0 = blockIdx.y * NTB;  // first n-tile of the workgroup

  const int haloItems = p.IH * p.IWp * 8;  // 16-byte items: 8 per pixel (64 channels)
  const int haloPadded = (haloItems + 63) & ~63;
  char* hal = smem;
  char* wbuf = smem + (size_t)haloPadded * 16;

  // this lane's pixel of each of the wave's MT m-tiles: tile row / column, halo pixel of tap (0, 0)
  int pl0[MT], pty[MT], ptx[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int pp = (wm * MT + i) * 16 + r;
    int ty = (int)__umulhi((unsigned)pp, p.magicTW);
    int tx = pp - ty * p.TW;
    if (ty >= p.TH) { ty = p.TH; tx = 0; }  // past the tile (TH * TW < BM): multiplied on halo pixel 0, never stored
    pty[i] = ty;
    ptx[i] = tx;
    pl0[i] = ty < p.TH ? (ty * STRIDE) * p.IWp + tx : 0;  // (stride 2: column 2 tx of the de-interleaved image is slot tx)
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
      const int qx = pix - py * p.IWp;  // column slot of the LDS image
      int px = qx;                      // halo column it holds
      if constexpr (STRIDE == 2) px = qx < p.HALF ? 2 * qx : 2 * (qx - p.HALF) + 1;
      const int iy = iy0 + py, ix = ix0 + px;
      const int ch = c0 + cg * 8;
      const char* src = reinterpret_cast<const char*>(g_big_zero16);
      if (idx < haloItems && px < p.IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin)
        src = p.x + ((((size_t)n * p.H + iy) * p.W + ix) * (size_t)p.ldx + ch) * 2;
      __builtin_amdgcn_global_load_lds((bgptr_t)src, (blptr_t)(hal + base * 16), 16, 0, 0);
    }
  };
  // weight slab of (tap, chunk c) -> buffer b: fragment f = kt * NTB + j (f < 2 * NTB); wave w brings fragments w, w + 8, ...
  auto stage_w = [&](int c, int tap, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int f0 = 0; f0 < 2 * NTB; f0 += 8) {
      const int f = f0 + wave;
      if (f < 2 * NTB) {
        const int kt = f / NTB, j = f - kt * NTB;  // NTB is a compile-time constant
        const int ktg = c * 2 + kt;
        const int nt = ntb0 + j;
        const char* src = reinterpret_cast<const char*>(g_big_zero16);
        if (ktg < p.KTT && nt < p.NTn) src = p.w + (((size_t)(tap * p.KTT + ktg) * p.NTn + nt) * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((bgptr_t)src, (blptr_t)(wbuf + b * WBUF + f * 1024), 16, 0, 0);
      }
    }
  };

  UPA_STAMP_AT(0);
  UPA_STAMP_HWID();
  stage_halo(0);
  stage_w(0, 0, 0);
  int buf = 0;
  for (int c = 0; c < nChunks; ++c) {
    int kh = 0, kw = 0;
#pragma unroll 1  // nine unrolled taps let the scheduler hoist every tap's addresses: 128+ VGPRs and spills in the 96-channel variant
    for (int tap = 0; tap < TAPS; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of slab (c, tap) (and of the halo) has landed
      __syncthreads();                                  // ... everyone's; everyone is done with the other weight buffer
      if (c == 0) UPA_STAMP_AT(1 + tap);
      if (tap + 1 < TAPS) stage_w(c, tap + 1, buf ^ 1);
      const int tapshift = kh * p.IWp + (STRIDE == 2 ? (kw >> 1) + (kw & 1) * p.HALF : kw);
      const char* wb = wbuf + buf * WBUF + (wn * NT) * 1024 + lane * 16;
      int paddr[MT], pswz[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int pl = pl0[i] + tapshift;
        paddr[i] = pl * 128;
        pswz[i] = pl & 7;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        // class-branch tail (Cin = 80: the second chunk holds ONE k-tile): skip the empty one.  Only there: in the general
        // variants the branch keeps the scheduler from interleaving the two k-tiles and cost the 40x40 / 20x20 layers 4-12 %
        if constexpr (TAIL == 2) {
          if (kt == 1 && c * 2 + 1 >= p.KTT) break;
        }
        if constexpr (NT <= 6) {
          u32x4 a[NT], b[MT];
#pragma unroll
          for (int j = 0; j < NT; ++j) a[j] = *reinterpret_cast<const u32x4*>(wb + (kt * NTB + j) * 1024);
#pragma unroll
          for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const u32x4*>(hal + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                                  *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j], 0, 0, 0);
        } else {
          // nine n-tiles (the stacked first convs of a Detect level): the weight fragments in two groups, so that 2 x 9 accumulator
          // tiles + one group of fragments stay inside 128 registers (two workgroups per CU); every accumulator still sees its products
          // in the same order
          constexpr int GS = (NT + 1) / 2;
          u32x4 b[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const u32x4*>(hal + paddr[i] + (((kt * 4 + g) ^ pswz[i]) << 4));
#pragma unroll
          for (int j0 = 0; j0 < NT; j0 += GS) {
            u32x4 a[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j)
              if (j0 + j < NT) a[j] = *reinterpret_cast<const u32x4*>(wb + (kt * NTB + j0 + j) * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < GS; ++j)
                if (j0 + j < NT)
                  acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a[j]),
                                                                           *reinterpret_cast<const bf16x8*>(&b[i]), acc[i][j0 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // keep the second group's loads behind the first group's products (register budget)
          }
        }
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
  UPA_STAMP_AT(10);

  if constexpr (TAIL == 1 || TAIL == 2) {
    // ---- Detect branch tail (head.py:94-100, 116-126, 151-169): h = SiLU(conv3x3 + b) never leaves the registers.
    // D layout: lane (g, r) holds channels 16j + 4g .. + 3 of pixel r; two neighbouring n-tiles (2s, 2s + 1) packed to bf16
    // ARE the B operand of a v_mfma_f32_16x16x32_bf16 k-step, in the k order (element e < 4: channel 32s + 4g + e, e >= 4:
    // 32s + 16 + 4g + e - 4) - the 1x1 weights are packed on the host in the same order (upa_pack_tail_weight), so the final
    // 1x1 conv is NT/2 k-steps x NT n-tiles of MFMA per m-tile straight from the accumulators: no LDS round trip, no store
    // of the (B,H,W,64|80) intermediate, no third launch.  Its output feeds the decode epilogue of detect_epi.h.
    f32x4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + g * 4);  // padded to NT * 16 by the host
    f32x4 acc2[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
      u32x4 hb[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = big_act<UPA_ACT_SILU>(acc[i][2 * s2][q] + bv[2 * s2][q]);
          v1[q] = big_act<UPA_ACT_SILU>(acc[i][2 * s2 + 1][q] + bv[2 * s2 + 1][q]);
        }
        hb[i] = u32x4{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x4 a2 = *reinterpret_cast<const u32x4*>(p.tw + ((size_t)(s2 * NT + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a2),
                                                               *reinterpret_cast<const bf16x8*>(&hb[i]), acc2[i][j], 0, 0, 0);
      }
    }
    if constexpr (NT & 1) {
      // odd tile count (80 channels = 5 tiles): the last tile alone is a 16-wide k-step - its packed accumulators are the B
      // operand of v_mfma_f32_16x16x16_bf16 (lane (g, r): channels 4g .. 4g + 3 of pixel r), and the A operand is the LOW
      // half of the k-step's packed fragment (element e < 4 = channel 32s + 4g + e, see upa_pack_tail_weight)
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      constexpr int j0 = NT - 1, s2 = NT / 2;
      u32x2 hh[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float v0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v0[q] = big_act<UPA_ACT_SILU>(acc[i][j0][q] + bv[j0][q]);
        hh[i] = u32x2{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3])};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x2 a2 = *reinterpret_cast<const u32x2*>(p.tw + ((size_t)(s2 * NT + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(&a2),
                                                                 *reinterpret_cast<const s16x4*>(&hh[i]), acc2[i][j], 0, 0, 0);
      }
    }
    f32x4 tbv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tbv[j] = *reinterpret_cast<const f32x4*>(p.tb + j * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
      const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW;
      const int al = pok ? oy * p.OW + ox : 0;  // level-local anchor; the image index n comes with the tile (no division)
      if constexpr (TAIL == 1) {
        static_assert(TAIL != 1 || NT == 4, "box branch = 4 sides x 16 bins");
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc2[i][j] + tbv[j];
        upa_detect_box_store(p.de, v, n, al, pok, g);
      } else {
        float best = -1.f;
        int bc = 0;
        if (p.de.keys_only) {  // uniform: boxes + best-class keys are all that single-label NMS reads
          f32x4 lg[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) lg[j] = acc2[i][j] + tbv[j];
          upa_detect_cls_keys_only<NT>(p.de, lg, pok, g, best, bc);
        } else {
#pragma unroll
          for (int j = 0; j < NT; ++j) upa_detect_cls_store(p.de, acc2[i][j] + tbv[j], j, n, al, pok, g, best, bc);
        }
        if (p.de.best_keys) upa_detect_best_key_store(p.de, best, bc, n, al, pok, lane);  // uniform
      }
    }
    return;
  }

  if constexpr (TAIL == 3) {
    // ---- training forward: z = the accumulators rounded to bf16 (no bias, no activation, no residual), stored as below, AND the
    // workgroup's per-channel sum / sum of squares of those rounded values.  Two n-tiles at a time (16 running sums live): over the
    // wave's m-tiles in registers, over the 16 pixels of the lane's row group by a fixed butterfly (xor 1, xor 2, mirror in 8, mirror in
    // 16), over the waves that share the channels in wave order through LDS (the halo image is dead behind the barrier).
    const int cw = (blockIdx.y * NTB + wn * NT) * 16;
    float* red = reinterpret_cast<float*>(smem);
    auto row16 = [](float v) __attribute__((always_inline)) {
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, true));
      return v;
    };
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; j += 2) {
      float ss[2][4], sq[2][4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) ss[jj][q] = sq[jj][q] = 0.f;
      const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
        const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW;
        const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
        const unsigned pk[2][2] = {{pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])},
                                   {pack_bf16x2(acc[i][j + 1][0], acc[i][j + 1][1]), pack_bf16x2(acc[i][j + 1][2], acc[i][j + 1][3])}};
        const float mk = pok ? 1.f : 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const float a = __uint_as_float(pk[jj][h2] << 16) * mk, b = __uint_as_float(pk[jj][h2] & 0xFFFF0000u) * mk;
            ss[jj][2 * h2] += a; sq[jj][2 * h2] = fmaf(a, a, sq[jj][2 * h2]);
            ss[jj][2 * h2 + 1] += b; sq[jj][2 * h2 + 1] = fmaf(b, b, sq[jj][2 * h2 + 1]);
          }
        auto lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (pok && cw + cb < p.Cout) *reinterpret_cast<u32x4*>(p.y + (pixoff * p.ldy + cw + cb) * 2) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ss[jj][q] = row16(ss[jj][q]); sq[jj][q] = row16(sq[jj][q]); }
        if (r == 0) {
          *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * (NT * 16) + (j + jj) * 16 + 4 * g) = f32x4{ss[jj][0], ss[jj][1], ss[jj][2], ss[jj][3]};
          *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * (NT * 16) + (j + jj) * 16 + 4 * g) = f32x4{sq[jj][0], sq[jj][1], sq[jj][2], sq[jj][3]};
        }
      }
    }
    __syncthreads();
    if (tid < 2 * NTB * 16) {
      const int k = tid / (NTB * 16), chw = tid - k * (NTB * 16);
      const int wn_ = chw / (NT * 16), cl = chw - wn_ * (NT * 16);
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < WM; ++m) t += red[((m * WN + wn_) * 2 + k) * (NT * 16) + cl];
      const int ch = blockIdx.y * NTB * 16 + chw;
      if (ch < p.stats_ld) p.stats[((size_t)blockIdx.x * 2 + k) * p.stats_ld + ch] = t;
    }
    return;
  }

  // ---- epilogue from the accumulators (as conv.hip): lane (g, r) holds channels 16j + 4g .. + 3 of pixel r of m-tile i;
  // v_permlane16_swap pairs the quads of two neighbouring n-tiles so every lane stores 16 contiguous bytes
  const int cw = (blockIdx.y * NTB + wn * NT) * 16;  // first channel of this wave
  // (nine n-tiles: the bias quads are fetched where they are used - 36 more live registers would spill)
  constexpr bool LAZY_BIAS = NT > 6;
  auto bias_of = [&](int j) __attribute__((always_inline)) {
    const int co = cw + j * 16 + g * 4;
    return (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  f32x4 biasv[LAZY_BIAS ? 1 : NT];
  if constexpr (!LAZY_BIAS) {
#pragma unroll
    for (int j = 0; j < NT; ++j) biasv[j] = bias_of(j);
  }
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = oy0 + pty[i], ox = ox0 + ptx[i];
      const bool pok = pty[i] < p.TH && oy < p.OH && ox < p.OW;
      const size_t pixoff = ((size_t)n * p.OH + oy) * p.OW + ox;
      char* yrow = p.y + (pixoff * p.ldy + cw) * 2;
      const char* rrow = p.res ? p.res + (pixoff * p.ldr + cw) * 2 : nullptr;
#pragma unroll
      for (int j = 0; j + 1 < NT; j += 2) {
        const int cb = 16 * (j + (g & 1)) + 8 * (g >> 1);
        const f32x4 bj0 = LAZY_BIAS ? bias_of(j) : biasv[LAZY_BIAS ? 0 : j], bj1 = LAZY_BIAS ? bias_of(j + 1) : biasv[LAZY_BIAS ? 0 : j + 1];
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v0[q] = big_act<ACT>(acc[i][j][q] + bj0[q]);
          v1[q] = big_act<ACT>(acc[i][j + 1][q] + bj1[q]);
        }
        bool ok = pok && cw + cb < p.Cout;
        char* dst = yrow + cb * 2;
        const char* rsrc = rrow ? rrow + cb * 2 : nullptr;
        if constexpr (TAIL == 4) {  // this lane's 8 channels belong to ONE phase (cin % 8 == 0): its pixel of dx
          const int ch = cw + cb;
          const int ph = ch / p.il_c, c = ch - ph * p.il_c;
          const int y = 2 * (oy - 1) + (ph >> 1), x = 2 * (ox - 1) + (ph & 1);
          ok = ok && y >= 0 && y < p.il_h && x >= 0 && x < p.il_w;
          const size_t off = ((((size_t)n * p.il_h + y) * p.il_w + x) * p.ldy + c) * 2;
          dst = p.y + off;
          rsrc = p.res ? p.res + off : nullptr;  // (accumulating: p.res = dx itself, same pitch)
        }
        if (p.res) {
          float x8[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0[q]), __float_as_uint(v1[q]), false, false);
            x8[q] = __uint_as_float(sw[0]);
            x8[4 + q] = __uint_as_float(sw[1]);
          }
          if (ok) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(rsrc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x8[2 * q] += __uint_as_float(rv[q] << 16);
              x8[2 * q + 1] += __uint_as_float(rv[q] & 0xFFFF0000u);
            }
            *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(x8[0], x8[1]), pack_bf16x2(x8[2], x8[3]),
                                                   pack_bf16x2(x8[4], x8[5]), pack_bf16x2(x8[6], x8[7])};
          }
        } else {
          auto lo = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if (ok) *reinterpret_cast<u32x4*>(dst) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      }
      if constexpr (NT & 1) {  // odd tile count: the last n-tile goes out as 8-byte channel quads
        constexpr int j = NT - 1;
        const int cb = 16 * j + 4 * g;
        float v[4];
        const f32x4 bj = LAZY_BIAS ? bias_of(j) : biasv[LAZY_BIAS ? 0 : j];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = big_act<ACT>(acc[i][j][q] + bj[q]);
        if (pok && cw + cb < p.Cout) {
          if (p.res) {
            const u32x2 rv = *reinterpret_cast<const u32x2*>(rrow + cb * 2);
            v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
            v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
          }
          *reinterpret_cast<u32x2*>(yrow + cb * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      }
    }
  };
  if (p.act == UPA_ACT_SILU) epilogue(std::integral_constant<int, UPA_ACT_SILU>{});
  else if (p.act == UPA_ACT_RELU) epilogue(std::integral_constant<int, UPA_ACT_RELU>{});
  else epilogue(std::integral_constant<int, UPA_ACT_NONE>{});
  UPA_STAMP_AT(11);
#ifdef UPA_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UPA_STAMP_AT(12);
#endif
}

template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
__global__ __launch_bounds__(512, 4) void conv_big_kernel(const BigParams p) {
  conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p, (int)blockIdx.x);
}

// Two problems of one instantiation in ONE grid: workgroups [0, split) belong to p0, the rest to p1 (same column count).  The small
// maps' launches (40 x 40 and 20 x 20 levels of the Detect head: 400 and 100 workgroups) are a fraction of a round each and mostly
// launch ramp + halo latency; side by side the smaller one rides inside the larger one's round.
template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
__global__ __launch_bounds__(512, 4) void conv_big_pair_kernel(const BigParams p0, const BigParams p1, const int split) {
  if ((int)blockIdx.x < split) conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p0, (int)blockIdx.x);
  else conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p1, (int)blockIdx.x - split);
}
// ... and three (all three levels of a Detect head: the 80 x 80 level on the 128-pixel variant too, the two small levels fill the
// last round of the large one).
template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
__global__ __launch_bounds__(512, 4) void conv_big_tri_kernel(const BigParams p0, const BigParams p1, const BigParams p2, const int s0,
                                                              const int s1) {
  if ((int)blockIdx.x < s0) conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p0, (int)blockIdx.x);
  else if ((int)blockIdx.x < s1) conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p1, (int)blockIdx.x - s0);
  else conv_big_body<KS, STRIDE, WM, WN, MT, NT, TAIL>(p2, (int)blockIdx.x - s1);
}

// ... and problems of TWO instantiations in one grid (box and class branches of the same Detect levels: independent chains whose
// kernels differ in the number of n-tiles or in the tail): up to two problems each.  Registers and LDS are the larger of the two.
template <int KS_, int STRIDE_, int WM_, int WN_, int MT_, int NT_, int TAIL_>
struct BigCfg {
  static constexpr int NTB = WN_ * NT_;
  __device__ static __forceinline__ void run(const BigParams& p, int bid) { conv_big_body<KS_, STRIDE_, WM_, WN_, MT_, NT_, TAIL_>(p, bid); }
};
template <class CA, class CB>
__global__ __launch_bounds__(512, 4) void conv_big_mix_kernel(const BigParams a0, const BigParams a1, const BigParams b0, const BigParams b1,
                                                              const int s0, const int s1, const int s2) {
  const int b = (int)blockIdx.x;
  if (b < s0) CA::run(a0, b);
  else if (b < s1) CA::run(a1, b - s0);
  else if (b < s2) CB::run(b0, b - s1);
  else CB::run(b1, b - s2);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int big_num_cu() {
  static int numCU = 0;
  if (!numCU) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCU <= 0) numCU = 256;
  }
  return numCU;
}

// Tile shape for a BM-pixel workgroup: fewest tiles per image first (least padding waste), then the smallest halo; the
// halo of a 64-channel chunk plus the two weight buffers must fit `lds_cap` bytes.  Returns false if nothing fits.
// LDS geometry of the halo image for a TH x TW tile: valid size IH x IW, column slots (stride 2: de-interleaved halves of
// HALF), pixel pitch IWp = the conflict-free one (upa_lds_pick_pitch) or, with `tight`, the smallest.
void big_halo_geometry(BigParams& p, bool tight = false) {
  p.IH = (p.TH - 1) * p.stride + p.KS;
  p.IW = (p.TW - 1) * p.stride + p.KS;
  p.HALF = p.stride == 2 ? (p.IW + 1) / 2 : 0;
  const int minp = p.stride == 2 ? 2 * p.HALF : p.IW;
  p.IWp = tight ? minp : upa_lds_pick_pitch(minp, p.TW, p.TH * p.TW, p.stride);
  p.magicTW = (unsigned)((0x100000000ULL + p.TW - 1) / p.TW);
  p.magicIW = (unsigned)((0x100000000ULL + p.IWp - 1) / p.IWp);
}
size_t big_halo_bytes(const BigParams& p) { return (((size_t)p.IH * p.IWp * 8 + 63) & ~(size_t)63) * 16; }

// Tile shape for a BM-pixel workgroup: fewest tiles per image first (least padding waste), then the smallest halo; the
// halo of a 64-channel chunk plus the two weight buffers must fit `lds_cap` bytes.  Tile heights tried per width: all the
// rows BM allows, and that many rows spread evenly over the tiles of a column (80 rows at 25 per tile = 4 tiles -> 20 rows each:
// same tile count, smaller halo, equal workgroups).  The conflict-free halo pitch is taken when it fits the cap, the tight one
// otherwise (occupancy first: measured in round 3, removing every bank conflict of the 64-column family changed its time by
// less than the run-to-run noise).  Returns false if nothing fits.
bool big_pick_tile(BigParams& p, int bm, int ntb, size_t lds_cap) {
  // the search is a pure function of (map size, kernel, stride, workgroup shape, LDS cap): memoised - an eager (uncaptured)
  // training step launches a hundred convolutions and the search costs ~0.3 ms of host time per call
  struct Pick { int th, tw, tight; };
  static std::mutex mu;
  static std::unordered_map<unsigned long long, Pick> memo;
  const unsigned long long key = ((unsigned long long)p.OH << 48) ^ ((unsigned long long)p.OW << 32) ^ ((unsigned long long)bm << 20) ^
                                 ((unsigned long long)ntb << 12) ^ ((unsigned long long)(lds_cap >> 9) << 3) ^ ((unsigned long long)p.KS << 56) ^ (p.KS == 3 ? 4u : 0u) ^
                                 (p.stride == 2 ? 2u : 0u);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(key);
    if (it != memo.end()) {
      if (it->second.th <= 0) return false;
      p.TH = it->second.th; p.TW = it->second.tw;
      big_halo_geometry(p, it->second.tight != 0);
      return true;
    }
  }
  long best = -1;
  int bth = 0, btw = 0;
  bool btight = false;
  for (int tw = 2; tw <= 256 && tw <= ((p.OW + 1) & ~1); ++tw) {
    int th0 = bm / tw;
    if (th0 > p.OH) th0 = p.OH;
    if (th0 < 1) continue;
    const int thb = cdiv(p.OH, cdiv(p.OH, th0));  // balanced rows
    for (int pass = 0; pass < 2; ++pass) {
      const int th = pass ? thb : th0;
      if (pass && thb == th0) break;
      for (int tight = 0; tight < 2; ++tight) {
        p.TH = th; p.TW = tw;
        big_halo_geometry(p, tight != 0);
        const size_t lds = big_halo_bytes(p) + 2 * (size_t)(2 * ntb * 1024) + 256;
        if (lds > lds_cap) continue;
        const long tiles = (long)cdiv(p.OW, tw) * cdiv(p.OH, th);
        const long cost = tiles * 65536 + (long)p.IH * p.IWp;
        if (best < 0 || cost < best) { best = cost; bth = th; btw = tw; btight = tight != 0; }
        break;  // the conflict-free pitch fits: no need for the tight one
      }
    }
  }
  p.TH = bth; p.TW = btw;
  if (best >= 0) big_halo_geometry(p, btight);
  {
    std::lock_guard<std::mutex> lk(mu);
    memo[key] = Pick{best >= 0 ? bth : 0, btw, btight ? 1 : 0};
  }
  return best >= 0;
}

template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
int big_launch_inst(const BigParams& p, size_t lds, hipStream_t s) {
  constexpr int NTB = WN * NT;
  const dim3 grid((unsigned)((long)p.tilesX * p.tilesY * p.N), (unsigned)cdiv(p.NTn, NTB));
  auto kern = conv_big_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>;
  if (hipError_t e = upa_full_lds<conv_big_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>>(); e != hipSuccess) {
    upa_set_error("conv_big: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
int big_launch_pair_inst(const BigParams& p0, const BigParams& p1, size_t lds, hipStream_t s) {
  constexpr int NTB = WN * NT;
  const long t0 = (long)p0.tilesX * p0.tilesY * p0.N, t1 = (long)p1.tilesX * p1.tilesY * p1.N;
  if (cdiv(p0.NTn, NTB) != cdiv(p1.NTn, NTB) || t0 + t1 >= (1L << 31)) return UPA_EUNSUPPORTED;
  const dim3 grid((unsigned)(t0 + t1), (unsigned)cdiv(p0.NTn, NTB));
  if (hipError_t e = upa_full_lds<conv_big_pair_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>>(); e != hipSuccess) {
    upa_set_error("conv_big: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL((conv_big_pair_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>), grid, dim3(512), lds, s, p0, p1, (int)t0);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int KS, int STRIDE, int WM, int WN, int MT, int NT, int TAIL = 0>
int big_launch_tri_inst(const BigParams& p0, const BigParams& p1, const BigParams& p2, size_t lds, hipStream_t s) {
  constexpr int NTB = WN * NT;
  const long t0 = (long)p0.tilesX * p0.tilesY * p0.N, t1 = (long)p1.tilesX * p1.tilesY * p1.N, t2 = (long)p2.tilesX * p2.tilesY * p2.N;
  if (cdiv(p0.NTn, NTB) != cdiv(p1.NTn, NTB) || cdiv(p0.NTn, NTB) != cdiv(p2.NTn, NTB) || t0 + t1 + t2 >= (1L << 31)) return UPA_EUNSUPPORTED;
  const dim3 grid((unsigned)(t0 + t1 + t2), (unsigned)cdiv(p0.NTn, NTB));
  if (hipError_t e = upa_full_lds<conv_big_tri_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>>(); e != hipSuccess) {
    upa_set_error("conv_big: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL((conv_big_tri_kernel<KS, STRIDE, WM, WN, MT, NT, TAIL>), grid, dim3(512), lds, s, p0, p1, p2, (int)t0, (int)(t0 + t1));
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// na, nb = 1 | 2 problems of each instantiation; every problem must be a single column block of its instantiation
template <class CA, class CB>
int big_launch_mix(const BigParams* pa, int na, const BigParams* pb, int nb, size_t lds, hipStream_t s) {
  long t[4] = {0, 0, 0, 0};
  for (int i = 0; i < na; ++i) { t[i] = (long)pa[i].tilesX * pa[i].tilesY * pa[i].N; if (pa[i].NTn > CA::NTB) return UPA_EUNSUPPORTED; }
  for (int i = 0; i < nb; ++i) { t[2 + i] = (long)pb[i].tilesX * pb[i].tilesY * pb[i].N; if (pb[i].NTn > CB::NTB) return UPA_EUNSUPPORTED; }
  const long tot = t[0] + t[1] + t[2] + t[3];
  if (tot >= (1L << 31)) return UPA_EUNSUPPORTED;
  if (hipError_t e = upa_full_lds<conv_big_mix_kernel<CA, CB>>(); e != hipSuccess) {
    upa_set_error("conv_big: cannot raise LDS limit: %s", hipGetErrorString(e));
    return UPA_ELAUNCH;
  }
  hipLaunchKernelGGL((conv_big_mix_kernel<CA, CB>), dim3((unsigned)tot), dim3(512), lds, s, pa[0], pa[na > 1 ? 1 : 0], pb[0], pb[nb > 1 ? 1 : 0],
                     (int)t[0], (int)(t[0] + t[1]), (int)(t[0] + t[1] + t[2]));
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

template <int WM, int WN, int MT, int NT>
int big_launch_ks(const BigParams& p, size_t lds, hipStream_t s) {
  if (p.KS == 1) return big_launch_inst<1, 1, WM, WN, MT, NT>(p, lds, s);
  if (p.KS == 2) return big_launch_inst<2, 1, WM, WN, MT, NT>(p, lds, s);  // (2 x 2, pad 1: the phase kernels of a stride-2 data gradient, train.hip)
  if (p.stride == 2) return big_launch_inst<3, 2, WM, WN, MT, NT>(p, lds, s);
  return big_launch_inst<3, 1, WM, WN, MT, NT>(p, lds, s);
}
}  // namespace

// Packed weights of a Detect branch's final 1x1 conv for the tail of conv_big_kernel: [k-step s][n-tile j][lane (g, r)][8 bf16],
// element e of lane (g, r) = W[co = 16j + r][ci = 32s + (e < 4 ? 4g + e : 16 + 4g + e - 4)] - the k order in which two packed
// accumulator tiles present their channels as an MFMA B operand.  cin / cout are zero-padded to 32 / 16.
extern "C" size_t upa_tail_packed_weight_bytes(int cout, int cin) {
  return (size_t)cdiv(cin, 32) * cdiv(cout, 16) * 1024;
}

static inline unsigned short big_host_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

extern "C" int upa_pack_tail_weight(const float* w, int cout, int cin, void* out) {
  UPA_CHECK_ARG(w && out && cout > 0 && cin > 0, "pack_tail_weight: bad args");
  const int ks = cdiv(cin, 32), nt = cdiv(cout, 16);
  unsigned short* o = (unsigned short*)out;
  size_t idx = 0;
  for (int s = 0; s < ks; ++s)
    for (int j = 0; j < nt; ++j)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, r = lane & 15;
        const int co = 16 * j + r;
        for (int e = 0; e < 8; ++e, ++idx) {
          const int ci = 32 * s + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4);
          o[idx] = big_host_bf16((co < cout && ci < cin) ? w[(size_t)co * cin + ci] : 0.f);
        }
      }
  return UPA_OK;
}

namespace {
// Fills p for one Detect branch tail and picks its workgroup size / tile; UPA_EUNSUPPORTED outside the fused form.
int branch_tail_prepare(BigParams& p, int& ntb, int& bm, size_t& lds, const void* x, int n, int h, int w, int c, int ldx,
                        const void* w3_packed, const float* b3, const void* wt_packed, const float* bt, int kind, int nc,
                        float stride_px, float* y, int a_total, int a0, unsigned long long* best_keys, int dtype, const upa_opts* opts) {
  UPA_CHECK_ARG(x && w3_packed && b3 && wt_packed && bt && y, "detect_branch_tail: null pointer");
  UPA_CHECK_ARG(kind == 1 || kind == 2, "detect_branch_tail: kind must be 1 (box) or 2 (class)");
  UPA_CHECK_ARG(n > 0 && h > 0 && w > 0 && a0 >= 0 && a0 + h * w <= a_total, "detect_branch_tail: level does not fit a_total");
  const int off = UPA_OPT(opts, no_branch_tail);
  ntb = kind == 1 ? 4 : (c == 80 ? 5 : 6);  // 80 class-branch channels (nc = 80 models): 5 tiles, no padded sixth
  if (off || dtype != UPA_BF16 || c % 8 != 0 || ldx % 8 != 0 || h * w < 2 || w < 2 || (kind == 1 && c != 64) ||
      (kind == 2 && (c > 96 || nc > ntb * 16)) || ((uintptr_t)x % 16) != 0 || !upa_magic_exact((long)h * w - 1, w)) {
    upa_set_error("detect_branch_tail: outside the fused form (bf16; box c = 64; class c <= 96, nc <= 96)");
    return UPA_EUNSUPPORTED;
  }
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.w = (const char*)w3_packed; p.bias = b3; p.tw = (const char*)wt_packed; p.tb = bt;
  p.N = n; p.H = h; p.W = w; p.Cin = c; p.ldx = ldx; p.OH = h; p.OW = w; p.Cout = ntb * 16; p.KS = 3; p.stride = 1; p.pad = 1;
  p.act = UPA_ACT_SILU;
  p.KTT = cdiv(c, 32);
  p.NTn = ntb;
  p.de.y = y; p.de.a_total = a_total; p.de.a0 = a0; p.de.HW = h * w; p.de.W = w;
  p.de.magicHW = upa_magic_div(h * w); p.de.magicW = upa_magic_div(w);
  p.de.nc = nc; p.de.stride_px = stride_px;
  if (kind == 2 && best_keys && (long)a_total * nc < (1L << 31)) p.de.best_keys = best_keys;
  p.de.keys_only = (p.de.best_keys && UPA_OPT(opts, keys_only)) ? 1 : 0;
  p.no_xcd = UPA_OPT(opts, no_xcd);
  const long px = (long)n * h * w;
  bm = (px + 255) / 256 < big_num_cu() ? 128 : 256;  // 128-pixel workgroups (one m-tile per wave) on the small levels
  if (const int f = UPA_OPT(opts, branch_tail_bm); f == 128 || f == 256) bm = f;
  if (!big_pick_tile(p, bm, ntb, 80 * 1024 - 512) && !big_pick_tile(p, bm, ntb, 160 * 1024)) return UPA_EUNSUPPORTED;
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  lds = big_halo_bytes(p) + 2 * (size_t)(2 * ntb * 1024) + 256;
  return UPA_OK;
}
int branch_tail_launch(const BigParams& p, int kind, int ntb, int bm, size_t lds, hipStream_t s) {
  if (kind == 1) return bm == 256 ? big_launch_inst<3, 1, 8, 1, 2, 4, 1>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 1, 4, 1>(p, lds, s);
  if (ntb == 5) return bm == 256 ? big_launch_inst<3, 1, 8, 1, 2, 5, 2>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 1, 5, 2>(p, lds, s);
  return bm == 256 ? big_launch_inst<3, 1, 8, 1, 2, 6, 2>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 1, 6, 2>(p, lds, s);
}
}  // namespace

// Second 3x3 conv of a Detect branch + the final 1x1 conv + that branch's half of the decode in ONE launch (bf16).
// x: (n, h, w, c) NHWC view, c = 64 (box branch, kind 1) or <= 96 (class branch, kind 2).  With CP = 64 (box) / 80 (class, c = 80) /
// 96 (class, any other c):
// w3 / b3 = the 3x3 conv packed by upa_pack_conv_weight as c -> CP (BN folded, zero filters / biases appended up to CP);
// wt = the 1x1 conv as upa_pack_tail_weight(cout = CP, cin = CP) of the zero-padded matrix, bt = its CP biases.
extern "C" int upa_detect_branch_tail(const void* x, int n, int h, int w, int c, int ldx, const void* w3_packed, const float* b3,
                                      const void* wt_packed, const float* bt, int kind, int nc, float stride_px, float* y,
                                      int a_total, int a0, unsigned long long* best_keys, int dtype, const upa_opts* opts,
                                      void* stream) {
  BigParams p;
  int ntb = 0, bm = 0;
  size_t lds = 0;
  if (const int rc = branch_tail_prepare(p, ntb, bm, lds, x, n, h, w, c, ldx, w3_packed, b3, wt_packed, bt, kind, nc, stride_px, y,
                                         a_total, a0, best_keys, dtype, opts); rc != UPA_OK)
    return rc;
  return branch_tail_launch(p, kind, ntb, bm, lds, (hipStream_t)stream);
}

// The same for several levels of one Detect head (same kind): levels whose problems land on the same kernel instantiation are
// launched TWO PER GRID (conv_big_pair_kernel) - the 40 x 40 and 20 x 20 levels at batch 32 are 400 + 100 workgroups of the 128-pixel
// variant, one partial round together instead of two launches.  Results are identical to per-level calls.  Any level outside the
// fused form -> UPA_EUNSUPPORTED with nothing launched.
extern "C" int upa_detect_branch_tail_group(const upa_branch_level* levels, int count, int kind, int nc, float* y, int a_total,
                                            unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(levels && count >= 1 && count <= 8, "detect_branch_tail_group: 1..8 levels");
  BigParams ps[8];
  int ntb[8], bm[8];
  size_t lds[8];
  for (int i = 0; i < count; ++i) {
    const upa_branch_level& v = levels[i];
    if (const int rc = branch_tail_prepare(ps[i], ntb[i], bm[i], lds[i], v.x, v.n, v.h, v.w, v.c, v.ldx, v.w3_packed, v.b3, v.wt_packed,
                                           v.bt, kind, nc, v.stride_px, y, a_total, v.a0, best_keys, dtype, opts); rc != UPA_OK)
      return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  // upa_opts.no_group: 0 = pairs (the default), 1 = one launch per level, 2 = threes too.  Three levels per grid measured SLOWER than
  // a pair + the 80 x 80 level alone (four steps in flight 50.2 k vs 51.5 k images/s, serial step equal): the large level loses more on
  // the 128-pixel variant than the ride saves.
  const int grouping = UPA_OPT(opts, no_group);
  auto max3 = [](size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); };
  for (int i = 0; i < count;) {
    if (grouping == 2 && i + 2 < count && ntb[i] == ntb[i + 1] && ntb[i] == ntb[i + 2] && ntb[i] != 6 && bm[i + 1] == 128 && bm[i + 2] == 128) {
      // a large first level joins on the 128-pixel variant
      BigParams p0 = ps[i];
      int ntb0 = ntb[i], bm0 = bm[i];
      size_t lds0 = lds[i];
      int rc = UPA_OK;
      if (bm0 != 128) {
        upa_opts o;
        memset(&o, 0, sizeof(o));
        if (opts) memcpy(&o, opts, opts->size < sizeof(o) ? opts->size : sizeof(o));
        o.size = sizeof(o);
        o.branch_tail_bm = 128;
        const upa_branch_level& v = levels[i];
        rc = branch_tail_prepare(p0, ntb0, bm0, lds0, v.x, v.n, v.h, v.w, v.c, v.ldx, v.w3_packed, v.b3, v.wt_packed, v.bt, kind, nc,
                                 v.stride_px, y, a_total, v.a0, best_keys, dtype, &o);
      }
      if (rc == UPA_OK && bm0 == 128) {
        const size_t l3 = max3(lds0, lds[i + 1], lds[i + 2]);
        rc = kind == 1 ? big_launch_tri_inst<3, 1, 8, 1, 1, 4, 1>(p0, ps[i + 1], ps[i + 2], l3, s)
                       : big_launch_tri_inst<3, 1, 8, 1, 1, 5, 2>(p0, ps[i + 1], ps[i + 2], l3, s);
        if (rc == UPA_OK) { i += 3; continue; }
        if (rc != UPA_EUNSUPPORTED) return rc;
      }
    }
    if (grouping != 1 && i + 1 < count && ntb[i] == ntb[i + 1] && bm[i] == 128 && bm[i + 1] == 128 && ntb[i] != 6) {
      const size_t l2 = lds[i] > lds[i + 1] ? lds[i] : lds[i + 1];
      const int rc = kind == 1 ? big_launch_pair_inst<3, 1, 8, 1, 1, 4, 1>(ps[i], ps[i + 1], l2, s)
                               : big_launch_pair_inst<3, 1, 8, 1, 1, 5, 2>(ps[i], ps[i + 1], l2, s);
      if (rc == UPA_OK) { i += 2; continue; }
      if (rc != UPA_EUNSUPPORTED) return rc;
    }
    if (const int rc = branch_tail_launch(ps[i], kind, ntb[i], bm[i], lds[i], s); rc != UPA_OK) return rc;
    ++i;
  }
  return UPA_OK;
}

// Box AND class branch tails of several levels of one Detect head: as two upa_detect_branch_tail_group calls, but the two kinds share
// grids too (conv_big_mix_kernel: two instantiations per grid) - a level pair on 128-pixel workgroups becomes ONE launch of four
// problems, a single level one launch of two.  yolov8n at batch 32: the six tails are two launches.  Identical results.
extern "C" int upa_detect_head_tails(const upa_branch_level* box, const upa_branch_level* cls, int count, int nc, float* y, int a_total,
                                     unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream) {
  UPA_CHECK_ARG(box && cls && count >= 1 && count <= 8, "detect_head_tails: 1..8 levels");
  BigParams pb[8], pc[8];
  int ntbb[8], bmb[8], ntbc[8], bmc[8];
  size_t ldsb[8], ldsc[8];
  for (int i = 0; i < count; ++i) {
    const upa_branch_level& b = box[i];
    const upa_branch_level& c = cls[i];
    if (const int rc = branch_tail_prepare(pb[i], ntbb[i], bmb[i], ldsb[i], b.x, b.n, b.h, b.w, b.c, b.ldx, b.w3_packed, b.b3, b.wt_packed, b.bt,
                                           1, nc, b.stride_px, y, a_total, b.a0, nullptr, dtype, opts); rc != UPA_OK)
      return rc;
    if (const int rc = branch_tail_prepare(pc[i], ntbc[i], bmc[i], ldsc[i], c.x, c.n, c.h, c.w, c.c, c.ldx, c.w3_packed, c.b3, c.wt_packed, c.bt,
                                           2, nc, c.stride_px, y, a_total, c.a0, best_keys, dtype, opts); rc != UPA_OK)
      return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const int grouping = UPA_OPT(opts, no_group);
  auto mx = [](size_t a, size_t b) { return a > b ? a : b; };
  using Box128 = BigCfg<3, 1, 8, 1, 1, 4, 1>;
  using Cls128 = BigCfg<3, 1, 8, 1, 1, 5, 2>;
  using Box256 = BigCfg<3, 1, 8, 1, 2, 4, 1>;
  using Cls256 = BigCfg<3, 1, 8, 1, 2, 5, 2>;
  for (int i = 0; i < count;) {
    if (grouping != 1 && ntbc[i] == 5) {
      if (i + 1 < count && ntbc[i + 1] == 5 && bmb[i] == 128 && bmb[i + 1] == 128 && bmc[i] == 128 && bmc[i + 1] == 128) {
        const size_t l = mx(mx(ldsb[i], ldsb[i + 1]), mx(ldsc[i], ldsc[i + 1]));
        // longest workgroups first (blockIdx order is dispatch order): class problems (five n-tiles, two channel chunks) before box
        const int rc = big_launch_mix<Cls128, Box128>(pc + i, 2, pb + i, 2, l, s);
        if (rc == UPA_OK) { i += 2; continue; }
        if (rc != UPA_EUNSUPPORTED) return rc;
      }
      if (bmb[i] == bmc[i]) {
        const size_t l = mx(ldsb[i], ldsc[i]);
        const int rc = bmb[i] == 128 ? big_launch_mix<Cls128, Box128>(pc + i, 1, pb + i, 1, l, s) : big_launch_mix<Cls256, Box256>(pc + i, 1, pb + i, 1, l, s);
        if (rc == UPA_OK) { ++i; continue; }
        if (rc != UPA_EUNSUPPORTED) return rc;
      }
    }
    if (const int rc = branch_tail_launch(pb[i], 1, ntbb[i], bmb[i], ldsb[i], s); rc != UPA_OK) return rc;
    if (const int rc = branch_tail_launch(pc[i], 2, ntbc[i], bmc[i], ldsc[i], s); rc != UPA_OK) return rc;
    ++i;
  }
  return UPA_OK;
}

bool upa_conv_big_pick_tile(BigParams& p, int bm, int ntb, size_t lds_cap) { return big_pick_tile(p, bm, ntb, lds_cap); }

bool upa_conv_big_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad,
                           int act, int dtype, const upa_opts* opts) {
  // upa_opts.conv_big: 0 = by the size rule below (default), 1 = never, 2 = every shape the kernel can run (parity tests,
  // tools/bench_conv.py)
  const int mode = UPA_OPT(opts, conv_big);
  if (mode == 1) return false;
  if (dtype != UPA_BF16 || !((k == 1 && stride == 1) || (k == 2 && stride == 1) || (k == 3 && (stride == 1 || stride == 2))) || pad != k / 2) return false;
  if (cin % 8 != 0 || ldx % 8 != 0 || cout % 8 != 0 || ldy % 8 != 0 || ldr % 8 != 0) return false;
  if (act != UPA_ACT_SILU && act != UPA_ACT_NONE && act != UPA_ACT_RELU) return false;
  if (mode == 2) return cout >= 64;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  const long px = (long)n * oh * ow;
  if (k == 1) {
    // pointwise layers stream faster through conv1x1.hip while their weight slice fits LDS; past that (512+ channels in)
    // the weights have to be shared per tap-chunk anyway
    // ... and GEMM-shaped ones with >= 1024 output columns (the RT-DETR value projections of all decoder layers batched:
    // 134400 x 1536 x 256), where the streaming kernel would re-read its input once per 128-column row of workgroups
    if (cout % 128 != 0 || px < 2048) return false;
    if (cout >= 1024 && cin >= 128 && px >= 32 * 1024) return true;
    return cin >= 512 && !upa_conv1x1_eligible(n, h, w, cin, ldx, cout, ldy, ldr != 0, k, stride, pad, act, dtype, opts);
  }
  // measured on MI355X (tools/bench_conv.py, yolov3-rtdetr bs 16 / yolov8n bs 32, round 2): 3x3 layers with whole 128-channel
  // output columns run at 800-1000 TFLOP/s here against 430-615 on the per-wave-weights kernel; the 80-channel class
  // branch of Detect (96-channel variant: 64->80 @80x80 45.6 -> 33.7 us, 128->80 @40x40 31.7 -> 19.9, 256->80 @20x20
  // 31.7 -> 25.8) and the 64-channel layers (64->64 @40x40 13.3 -> 11.5 us, @80x80 31.5 -> 27.9, 64->128 stride 2 @80x80
  // 27.5 -> 21.4) win from 64 input channels; narrower inputs (32->64) and 1x1 layers stay where they are
  if (cin < 64 || px < 2048) return false;
  if (cout % 128 == 0) return cin >= 128 || (px >= 8192 && (stride == 2 || px >= 200 * 1024));
  if (px < 8192) return false;
  if (cout == 80 || cout == 96) return stride == 1 && k == 3;
  if (cout == 144) return stride == 1 && k == 3;  // the two first convs of a Detect level stacked (64 box + 80 class channels, head.py)
  if (cout == 64) return stride == 1;
  return false;
}

namespace {
// Workgroup shape, tile and LDS of one conv_big problem (p.KTT / NTn filled in here).
int big_prepare(BigParams& p, int& ntb, int& bm, size_t& lds, const upa_opts* opts, bool query_only, int force_ntb = 0) {
  p.KTT = (p.Cin + 31) / 32;
  p.NTn = (p.Cout + 15) / 16;
  // workgroup columns: 128 channels, 96 for Cout in (64, 96], 64 for Cout <= 64 - and 64 for wider layers whose 128-pixel x
  // 128-channel workgroups would still be fewer than the CUs (20x20 maps: twice the workgroups, each with half the weights)
  ntb = p.NTn <= 4 ? 4 : (p.NTn <= 6 ? 6 : 8);
  if (p.NTn == 5 && p.KS == 3 && p.stride == 1) ntb = 5;  // 80 output channels (Detect class branch): 8 x 1 waves x 5 tiles
  if (p.NTn == 9 && p.KS == 3 && p.stride == 1) ntb = 9;  // 144 = 64 + 80: both first convs of a Detect level, one halo, nine tiles
  if (force_ntb == 5 && p.NTn <= 5 && p.KS == 3 && p.stride == 1) ntb = 5;  // a 64-channel problem sharing a grid with an 80-channel one
  const long px = (long)p.N * p.OH * p.OW;
  if (ntb == 8 && p.NTn % 4 == 0 && (px + 127) / 128 * cdiv(p.NTn, 8) < big_num_cu()) ntb = 4;
  const int cols = cdiv(p.NTn, ntb);
  // 256-pixel workgroups unless that leaves most of the chip idle (fewer workgroups than CUs): then 128-pixel ones
  bm = 256;
  if ((px + 255) / 256 * cols < big_num_cu()) bm = 128;
  if (const int f = UPA_OPT(opts, conv_big_bm); f == 128 || f == 256 || (f == 512 && p.KS == 3 && p.stride == 1 && (ntb == 4 || ntb == 5))) bm = f;
  if (query_only) return UPA_OK;
  if (p.KS == 1) {  // pointwise: an NHWC view has one uniform pixel stride - flatten (n, h, w) into one row
    p.N = 1; p.H = 1; p.W = (int)px; p.OH = 1; p.OW = (int)px;
    p.TH = 1; p.TW = bm;
    big_halo_geometry(p, true);
  } else if (!big_pick_tile(p, bm, ntb, 80 * 1024 - 512) && !big_pick_tile(p, bm, ntb, 160 * 1024)) {
    // (first try: two workgroups per CU; stride-2 halos may need the whole LDS)
    return UPA_EUNSUPPORTED;
  }
  p.tilesX = cdiv(p.OW, p.TW);
  p.tilesY = cdiv(p.OH, p.TH);
  lds = big_halo_bytes(p) + 2 * (size_t)(2 * ntb * 1024) + 256;
  if (lds > 160 * 1024) return UPA_EUNSUPPORTED;
  return UPA_OK;
}
// the statistics epilogue (TAIL 3): forward convolutions of the training step - k 1 | 3 -, even n-tile counts per wave
template <int WM, int WN, int MT, int NT>
int big_launch_ks_stats(const BigParams& p, size_t lds, hipStream_t s) {
  if (p.KS == 1) return big_launch_inst<1, 1, WM, WN, MT, NT, 3>(p, lds, s);
  if (p.KS != 3) return UPA_EUNSUPPORTED;
  if (p.stride == 2) return big_launch_inst<3, 2, WM, WN, MT, NT, 3>(p, lds, s);
  return big_launch_inst<3, 1, WM, WN, MT, NT, 3>(p, lds, s);
}
int big_dispatch_stats(const BigParams& p, int ntb, int bm, size_t lds, hipStream_t s) {
  if (bm != 128 && bm != 256) return UPA_EUNSUPPORTED;
  if (ntb == 8) return bm == 256 ? big_launch_ks_stats<4, 2, 4, 4>(p, lds, s) : big_launch_ks_stats<4, 2, 2, 4>(p, lds, s);
  if (ntb == 4) return bm == 256 ? big_launch_ks_stats<8, 1, 2, 4>(p, lds, s) : big_launch_ks_stats<4, 2, 2, 2>(p, lds, s);
  return UPA_EUNSUPPORTED;
}
int big_dispatch(const BigParams& p, int ntb, int bm, size_t lds, hipStream_t s) {
  if (bm == 512) return ntb == 5 ? big_launch_inst<3, 1, 8, 1, 4, 5>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 4, 4>(p, lds, s);
  if (ntb == 5) return bm == 256 ? big_launch_inst<3, 1, 8, 1, 2, 5>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 1, 5>(p, lds, s);
  if (ntb == 9) return bm == 256 ? big_launch_inst<3, 1, 8, 1, 2, 9>(p, lds, s) : big_launch_inst<3, 1, 8, 1, 1, 9>(p, lds, s);
  if (ntb == 8) return bm == 256 ? big_launch_ks<4, 2, 4, 4>(p, lds, s) : big_launch_ks<4, 2, 2, 4>(p, lds, s);
  if (ntb == 6) return bm == 256 ? big_launch_ks<4, 2, 4, 3>(p, lds, s) : big_launch_ks<4, 2, 2, 3>(p, lds, s);
  return bm == 256 ? big_launch_ks<8, 1, 2, 4>(p, lds, s) : big_launch_ks<4, 2, 2, 2>(p, lds, s);
}
}  // namespace

int upa_conv_big_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts) {
  p.no_xcd = UPA_OPT(opts, no_xcd);
  int ntb = 0, bm = 0;
  size_t lds = 0;
  const int rc = big_prepare(p, ntb, bm, lds, opts, query_only != 0);
  if (variant) *variant = (1 << 23) | (ntb << 4) | (bm >> 7);  // (bm >> 7: 1 = 128, 2 = 256, 4 = 512 pixels)
  if (query_only || rc != UPA_OK) return rc;
  return big_dispatch(p, ntb, bm, lds, (hipStream_t)stream);
}

// The four stacked 2x2 phase correlations of a stride-2 data gradient with the interleaving epilogue (TAIL 4): p = the phase convolution
// (dz (n, oh, ow, cout) -> 4 cin channels at (oh + 1, ow + 1), KS 2, pad 1), p.y / p.ldy = dx and its pixel pitch, p.il_h / il_w / il_c = dx's
// height, width and channel count, p.res = dx when accumulating.  UPA_EUNSUPPORTED outside the 128-channel-column variants.
int upa_conv_big_launch_interleave(BigParams p, void* stream, const upa_opts* opts) {
  p.no_xcd = UPA_OPT(opts, no_xcd);
  if (p.KS != 2 || p.stride != 1 || p.act != UPA_ACT_NONE || p.bias || p.il_c % 8 != 0 || p.Cout != 4 * p.il_c) return UPA_EUNSUPPORTED;
  int ntb = 0, bm = 0;
  size_t lds = 0;
  if (const int rc = big_prepare(p, ntb, bm, lds, opts, false); rc != UPA_OK) return rc;
  if (ntb != 8 || (bm != 128 && bm != 256)) return UPA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  return bm == 256 ? big_launch_inst<2, 1, 4, 2, 4, 4, 4>(p, lds, s) : big_launch_inst<2, 1, 4, 2, 2, 4, 4>(p, lds, s);
}

// The convolution with the statistics epilogue: p.stats receives *rows rows of [2][p.stats_ld] floats (one per pixel tile; p.stats_ld is
// set here to 16 * n-tiles).  UPA_EUNSUPPORTED (nothing launched) when the layer's workgroup shape has no such epilogue (80 / 96-channel
// columns) or the rows would not fit max_rows.
int upa_conv_big_launch_stats(BigParams p, int* rows, long max_rows, void* stream, const upa_opts* opts) {
  p.no_xcd = UPA_OPT(opts, no_xcd);
  if (!p.stats || p.res || p.act != UPA_ACT_NONE) return UPA_EUNSUPPORTED;
  int ntb = 0, bm = 0;
  size_t lds = 0;
  if (const int rc = big_prepare(p, ntb, bm, lds, opts, false); rc != UPA_OK) return rc;
  if ((ntb != 8 && ntb != 4) || (bm != 128 && bm != 256)) return UPA_EUNSUPPORTED;
  const long nrows = (long)p.tilesX * p.tilesY * p.N;
  if (nrows > max_rows) return UPA_EUNSUPPORTED;
  p.stats_ld = p.NTn * 16;
  *rows = (int)nrows;
  return big_dispatch_stats(p, ntb, bm, lds, (hipStream_t)stream);
}

// Two or three conv_big problems: one grid when they land on the same 128-pixel 3x3 stride-1 instantiation (see conv_big_pair_kernel;
// with three, a first problem that would take 256-pixel workgroups joins on the 128-pixel variant), else one launch each.  Returns
// UPA_EUNSUPPORTED (nothing launched) if a problem cannot run on conv_big at all.  *consumed = how many problems were launched.
int upa_conv_big_launch_group(const BigParams* probs, int count, int* consumed, void* stream, const upa_opts* opts) {
  *consumed = 0;
  if (count < 2) return UPA_EUNSUPPORTED;
  const int grouping = UPA_OPT(opts, no_group);
  if (grouping == 1) return UPA_EUNSUPPORTED;
  if (count >= 4) {  // two 64-channel and two 80-channel problems on 128-pixel workgroups (the small levels' first convs): one grid
    BigParams q[4];
    int nt4[4], bm4[4];
    size_t l4[4];
    bool ok = true;
    for (int i = 0; i < 4 && ok; ++i) {
      q[i] = probs[i];
      ok = q[i].KS == 3 && q[i].stride == 1 && big_prepare(q[i], nt4[i], bm4[i], l4[i], opts, false) == UPA_OK && bm4[i] == 128;
    }
    if (ok && nt4[0] == 4 && nt4[1] == 4 && nt4[2] == 5 && nt4[3] == 5) {
      size_t l = l4[0];
      for (int i = 1; i < 4; ++i) l = l4[i] > l ? l4[i] : l;
      // longest workgroups first: the 80-channel problems before the 64-channel ones, and within a kind the one with more input
      // channels (more chunks per workgroup: the 20 x 20 level) before the other - blockIdx order is dispatch order
      const bool sw0 = q[1].Cin > q[0].Cin, sw1 = q[3].Cin > q[2].Cin;
      const BigParams a[2] = {sw1 ? q[3] : q[2], sw1 ? q[2] : q[3]}, b[2] = {sw0 ? q[1] : q[0], sw0 ? q[0] : q[1]};
      const int rc = big_launch_mix<BigCfg<3, 1, 8, 1, 1, 5, 0>, BigCfg<3, 1, 4, 2, 2, 2, 0>>(a, 2, b, 2, l, (hipStream_t)stream);
      if (rc == UPA_OK) { *consumed = 4; return UPA_OK; }
      if (rc != UPA_EUNSUPPORTED) return rc;
    }
  }
  BigParams p[3];
  int ntb[3], bm[3];
  size_t lds[3];
  const int m = count >= 3 && grouping == 2 ? 3 : 2;  // (threes: experiment only, see upa_detect_branch_tail_group)
  for (int i = 0; i < m; ++i) {
    p[i] = probs[i];
    if (p[i].KS != 3 || p[i].stride != 1) return UPA_EUNSUPPORTED;
    if (const int rc = big_prepare(p[i], ntb[i], bm[i], lds[i], opts, false); rc != UPA_OK) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  auto ok_ntb = [](int v) { return v == 4 || v == 5; };
  if (m == 3 && ntb[0] == ntb[1] && ntb[0] == ntb[2] && ok_ntb(ntb[0]) && bm[1] == 128 && bm[2] == 128) {
    int rc = UPA_OK;
    if (bm[0] != 128) {
      upa_opts o;
      memset(&o, 0, sizeof(o));
      if (opts) memcpy(&o, opts, opts->size < sizeof(o) ? opts->size : sizeof(o));
      o.size = sizeof(o);
      o.conv_big_bm = 128;
      p[0] = probs[0];
      rc = big_prepare(p[0], ntb[0], bm[0], lds[0], &o, false);
    }
    if (rc == UPA_OK && bm[0] == 128 && ntb[0] == ntb[1]) {
      const size_t l3 = lds[0] > lds[1] ? (lds[0] > lds[2] ? lds[0] : lds[2]) : (lds[1] > lds[2] ? lds[1] : lds[2]);
      rc = ntb[0] == 4 ? big_launch_tri_inst<3, 1, 4, 2, 2, 2>(p[0], p[1], p[2], l3, s) : big_launch_tri_inst<3, 1, 8, 1, 1, 5>(p[0], p[1], p[2], l3, s);
      if (rc == UPA_OK) { *consumed = 3; return UPA_OK; }
      if (rc != UPA_EUNSUPPORTED) return rc;
    }
    // fall through to a pair of the first two as prepared without the override
    p[0] = probs[0];
    if (const int rc2 = big_prepare(p[0], ntb[0], bm[0], lds[0], opts, false); rc2 != UPA_OK) return rc2;
  }
  // a pair: same workgroup size (128 or 256 pixels) and 64 | 80 output channels; a 64-channel problem next to an 80-channel one runs on
  // the five-tile instantiation too (its fifth n-tile multiplies zero weights and is not stored: +25 % MFMAs on that problem, one
  // grid instead of two - the two first convs of a Detect level read the same input)
  if (ok_ntb(ntb[0]) && ok_ntb(ntb[1]) && bm[0] == bm[1] && (bm[0] == 128 || bm[0] == 256)) {
    if (ntb[0] != ntb[1]) {
      const int lo = ntb[0] < ntb[1] ? 0 : 1;
      p[lo] = probs[lo];
      int bmf = 0;
      if (const int rc = big_prepare(p[lo], ntb[lo], bmf, lds[lo], opts, false, 5); rc != UPA_OK) return rc;
      if (bmf != bm[1 - lo] || ntb[lo] != 5) return UPA_EUNSUPPORTED;
    }
    const size_t l2 = lds[0] > lds[1] ? lds[0] : lds[1];
    int rc;
    if (bm[0] == 128) rc = ntb[0] == 4 ? big_launch_pair_inst<3, 1, 4, 2, 2, 2>(p[0], p[1], l2, s) : big_launch_pair_inst<3, 1, 8, 1, 1, 5>(p[0], p[1], l2, s);
    else rc = ntb[0] == 4 ? big_launch_pair_inst<3, 1, 8, 1, 2, 4>(p[0], p[1], l2, s) : big_launch_pair_inst<3, 1, 8, 1, 2, 5>(p[0], p[1], l2, s);
    if (rc == UPA_OK) { *consumed = 2; return UPA_OK; }
    if (rc != UPA_EUNSUPPORTED) return rc;
  }
  return UPA_EUNSUPPORTED;
}
