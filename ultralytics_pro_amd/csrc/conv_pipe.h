// Internal interface between conv.hip (dispatch) and conv_pipe.hip (software-pipelined 3x3 stride-1 kernel).
#pragma once
struct PipeParams {
  const char* x;
  char* y;
  const char* res;
  const char* w;       // packed [tap][ktile][ntile][lane][16 B] (upa_pack_conv_weight)
  const float* bias;
  int N, H, W, Cin, ldx, Cout, ldy, ldr;
  int KTT, NTn, nt0;   // k-tiles (32 bf16 channels), packed n-tiles, first n-tile of this launch
  int tilesX, tilesY, numTiles;
  int act;
  int pool;            // 1 = y holds MaxPool2d(2, 2)(act(conv)) at (H / 2, W / 2): conv3x3_pipe_kernel<.., POOL>
#ifdef UPA_ABLATE
  int ablate;  // debug build only (upa_opts.ablate_pipe): 1 no halo DMA, 2 no weight loads, 4 no stores, 8 no MFMA, 16 no epilogue
#endif
};
bool upa_conv_pipe_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad,
                            int act, int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 21) | NTW of the first launch; query_only = 1 skips the launches
int upa_conv_pipe_launch(PipeParams p, int query_only, int* variant, void* stream, const upa_opts* opts);

// ---- conv1x1.hip: streaming pointwise convolution (bf16, k1 s1 p0, no residual)
#include "detect_epi.h"
struct C1Params {
  const char* x;
  char* y;
  const char* w;       // packed [ktile][ntile][lane][16 B]
  const float* bias;
  int P;               // pixels (n*h*w): an NHWC view has one uniform pixel stride
  int Cin, ldx, Cout, ldy;
  int KTT, NTn, groups;
  int act;
#ifdef UPA_ABLATE
  int ablate;  // debug build only (upa_opts.ablate_c1): 1 no input DMA, 2 no weight loads, 4 no stores, 8 no MFMA
#endif
  int epi;     // 0: y = act(conv + bias) as bf16 rows; 1 / 2: Detect box / class decode fused on the end (detect_epi.h), the
               // bf16 rows are written too when y != nullptr; 3: the rows + the workgroup's per-channel statistics (stats)
  DetectEpi de;
  // virtual Upsample(2x nearest) + Concat in front of the conv (upa_conv1x1_upcat): the first upKT k-tiles of a pixel come from
  // pixel (y / 2, x / 2) of the half-resolution tensor `up`, the rest from x (the concat buffer, whose first channels stay unwritten)
  float* stats;   // epi 3 (training forward): row blockIdx.x = [2][stats_ld] f32 sums / sums of squares of the stored values
  int stats_ld;
  const char* up;
  int upKT, up_ld, upH, upW;       // k-tiles taken from `up`, its pixel stride (elements), FULL-resolution H and W
  unsigned upMagicW, upMagicH;
};
bool upa_conv1x1_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, bool residual, int k, int stride,
                          int pad, int act, int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 22) | waves << 8 | MT << 4 | NTW; query_only = 1 skips the launch
int upa_conv1x1_launch(C1Params p, int n_pixels, int query_only, int* variant, void* stream, const upa_opts* opts);
// p.stats != nullptr (act none, no bias): the convolution + the first stage of the batch statistics; *rows = rows written (one per
// workgroup).  UPA_EUNSUPPORTED = nothing launched (odd n-tile count per workgroup, or more rows than max_rows)
int upa_conv1x1_launch_stats(C1Params p, int n_pixels, int* rows, long max_rows, void* stream, const upa_opts* opts);

// ---- conv_big.hip: large-tile implicit GEMM with both operands shared through LDS (bf16, k 1 | 3, stride 1 | 2)
struct BigParams {
  const char* x;
  char* y;
  const char* res;
  const char* w;       // packed [tap][ktile][ntile][lane][16 B] (upa_pack_conv_weight)
  const float* bias;
  int N, H, W, Cin, ldx, OH, OW, Cout, ldy, ldr;
  int KS, stride, pad;
  int TH, TW, tilesX, tilesY, IH, IW;
  int IWp;             // LDS pixel pitch of the halo image (>= IW, or >= 2 * HALF at stride 2): upa_lds_pick_pitch
  int HALF;            // stride 2: the halo columns are stored de-interleaved, even columns at [0, HALF), odd at [HALF, 2 HALF)
  int KTT, NTn;
  int act;
  unsigned magicTW, magicIW;   // magicIW divides by IWp
  int no_xcd;          // 1 = workgroup b takes tile b (0: the XCD-aware order, upa_opts.no_xcd for A/B)
  const char* tw;      // Detect branch tail: final 1x1 conv packed by upa_pack_tail_weight
  const float* tb;     // ... its bias (zero-padded to 16 * n-tiles)
  DetectEpi de;        // ... and the decode it feeds (detect_epi.h)
  // training forward (act none, no residual): per-workgroup sums of the stored values and of their squares, row blockIdx.x =
  // [2][stats_ld] floats (stats_ld = 16 * n-tiles of the layer); nullptr = a plain convolution
  float* stats;
  int stats_ld;
  int il_h, il_w, il_c;  // interleaving epilogue (conv_big TAIL 4): dx's height, width, channels
};
bool upa_conv_big_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad,
                           int act, int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 23) | n-tiles per workgroup << 4 | pixels per workgroup / 128; query_only = 1 skips the launch
int upa_conv_big_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts);
// p.stats != nullptr: the convolution + the first stage of the batch statistics (see BigParams::stats); *rows = rows written
int upa_conv_big_launch_stats(BigParams p, int* rows, long max_rows, void* stream, const upa_opts* opts);
int upa_conv_big_launch_interleave(BigParams p, void* stream, const upa_opts* opts);  // see conv_big.hip
// the first two or three problems of a list in ONE grid if they share a 128-pixel 3x3 stride-1 instantiation (*consumed = how many);
// UPA_EUNSUPPORTED = nothing launched (the caller launches the first problem alone and tries again from the next)
int upa_conv_big_launch_group(const BigParams* probs, int count, int* consumed, void* stream, const upa_opts* opts);

// tile shape + halo geometry (TH, TW, IH, IW, IWp, HALF, magicTW, magicIW) of a bm-pixel x ntb-n-tile workgroup whose halo + two weight slabs
// + 256 B fit lds_cap (conv_big.hip's search, memoised); false if nothing fits.  p.OH / OW / KS / stride must be set.
bool upa_conv_big_pick_tile(BigParams& p, int bm, int ntb, size_t lds_cap);

// ---- conv_p8.hip: 8-wave two-group phased kernel for the MFMA-bound 3x3 stride-1 layers (bf16, Cin % 64 == 0, Cout % 128 == 0); BigParams as conv_big
bool upa_conv_p8_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad, int act,
                          int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 26) | 8 << 4 | 2; query_only = 1 skips the launch
int upa_conv_p8_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts);

// ---- conv_mm.hip: 4-wave 32x32x16-MFMA kernel for the MFMA-bound 3x3 stride-1 layers (bf16, Cin % 64 == 0, Cout % 128 == 0); BigParams as conv_big
bool upa_conv_mm_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, int ldr, int k, int stride, int pad, int act,
                          int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 25) | 8 << 4 | 2; query_only = 1 skips the launch
int upa_conv_mm_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts);

// ---- conv_ws3.hip: persistent weights-stationary 3x3 (bf16, stride 1, pad 1, Cin <= 64, Cout = 64); BigParams as conv_big
bool upa_conv_ws3_eligible(int n, int h, int w, int cin, int ldx, int cout, int ldy, bool residual, int k, int stride, int pad,
                           int act, int dtype, const upa_opts* opts);
// variant (if non-null) receives (1 << 24) | NT << 4 | MT; query_only = 1 skips the launch
int upa_conv_ws3_launch(BigParams p, int query_only, int* variant, void* stream, const upa_opts* opts);
int upa_conv_ws3_launch_stats(BigParams p, int* rows, long max_rows, void* stream, const upa_opts* opts);  // as upa_conv_big_launch_stats

// ---- conv_pair.hip: Bottleneck (3x3 -> 3x3 [+ x]) as one kernel, the intermediate tile in LDS (bf16, C = 32 | 64)
struct PairParams {
  const char* x;
  char* y;
  const char* w1;      // packed [tap][ktile][ntile][lane][16 B] of conv 1 (C -> C)
  const char* w2;
  const float* b1;
  const float* b2;
  int N, H, W, OH, OW, ldx, ldy;
  int TH, TW, tilesX, tilesY;
  int IWp, MWp;        // LDS pixel pitches of the input halo (>= TW + 4) and of the mid tile (>= TW + 2): upa_lds_pick_pitch
  unsigned magicIW, magicMW, magicTW;  // divide by IWp (halo staging), TW + 2 (mid pixel slots), TW
  // CV2 form (C2f with one Bottleneck of 32 channels): the C2f's cv2 (1x1 over cat(y0, y1, b) -> 64 channels) on the end
  const char* y0;      // the y0 slice of the C2f concat buffer (same pixel stride ldx as x = the y1 slice)
  const char* wc_std;  // cv2 columns [0, 64) = (y0 | y1): upa_pack_conv_weight(64 -> 64, k = 1) layout [2 k-tiles][4 n-tiles]
  const char* wc_b;    // cv2 columns [64, 96) = b: upa_pack_tail_weight(64, 32) layout [1 k-step][4 n-tiles] (accumulator k order)
  const float* bc;     // cv2 bias (64)
  char* out; int ldout;
};
