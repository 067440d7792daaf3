/* upa.h - C ABI of libupa_hip.so: the MI355X (gfx950) kernels behind the ultralytics.nn operator API.
 *
 * The reference (Chriz122/ultralytics_pro) has no FFI: its operator layer is Python classes that call torch ops
 * (SURVEY.md 8b).  Each entry point below replaces the torch dispatch of one reference operator on the detect hot
 * path; the citation after each prototype names the reference code whose arithmetic it takes over
 * (paths relative to ultralytics/).
 *
 * Conventions
 *  - every function returns 0 on success, a negative UPA_E* code otherwise; no exceptions, no hidden allocation,
 *    no global state (the library reads no environment variable and keeps no mode: dispatch overrides travel with the
 *    call as a caller-owned `upa_opts`, NULL = defaults; the only statics are per-kernel one-time LDS-limit
 *    attributes and the cached CU count); all work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *    graph-capturable; re-entrant per stream and per thread (the error string is thread-local).
 *  - activations are NHWC "views": element (n,h,w,c) of a view lives at ptr + ((n*H + h)*W + w)*ld + c, where
 *    `ld` (pixel stride, elements) may exceed C, so a view can be a channel slice of a wider concat buffer
 *    (concat-by-construction, C2f/SPPF/Concat).  Channel offsets and C must be multiples of 16 bytes.
 *  - dtype: UPA_F32 (parity mode, exact f32 MFMA) or UPA_BF16 (perf mode, bf16 storage, f32 accumulate).
 *  - model boundary tensors keep the reference layout: input NCHW, Detect output (B, 4+nc, A) f32,
 *    NMS output (B, max_det, 6) f32 + int32 counts.
 */
#ifndef UPA_H
#define UPA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { UPA_F32 = 0, UPA_BF16 = 1, UPA_U8_BGR_HWC = 2 /* stem input only: uint8 (N,H,W,3) BGR frames */ };
enum { UPA_ACT_NONE = 0, UPA_ACT_SILU = 1, UPA_ACT_RELU = 2 };
enum { UPA_OK = 0, UPA_EINVAL = -1, UPA_EUNSUPPORTED = -2, UPA_EWORKSPACE = -3, UPA_ELAUNCH = -4 };

/* Dispatch / tuning overrides of ONE call (round 3: replaces the process-global UPA_* environment switches and
 * upa_conv_big_mode of rounds 1-2, so two models in one process can choose differently).  Caller-owned, read during the
 * call only; NULL or an all-zero struct = production defaults; `size` = sizeof(upa_opts) of the caller's header (fields
 * past it read as 0, so the struct can grow).  Used by the parity tests (force a kernel family onto small shapes), by
 * tools/bench_conv.py sweeps and by A/B measurements; production code passes NULL. */
typedef struct upa_opts {
  uint32_t size;
  int32_t conv_big;        /* csrc/conv_big.hip inside upa_conv2d_bias_act: 0 = by the size rule, 1 = never, 2 = every shape it can run, 3 = the rule restricted to maps of < 100000 pixels (at most one tile per workgroup) */
  int32_t conv_big_bm;     /* its workgroup pixels: 0 = auto | 128 | 256 | 512 */
  int32_t conv_force[4];   /* conv_igemm variant WM, WN, MT, NT for every conv whose Cout fits it (0 = none) */
  int32_t conv_ckt;        /* conv_igemm k-tiles per chunk: 0 = auto | 1 | 2 | 4 */
  int32_t no_ws, no_pipe, no_1x1, no_c16, no_upcat;  /* 1 = never dispatch conv_ws / conv3x3_pipe / conv1x1_stream / conv3x3_c16 / the virtual upsample */
  int32_t pipe_all;        /* conv3x3_pipe on every eligible channel count (default: the measured winners) */
  int32_t pipe_min_tiles;  /* ... from this many wave tiles (0 = 1024) */
  int32_t pipe_wgs, c16_wgs;        /* persistent workgroups of conv3x3_pipe (0 = 256) / conv3x3_c16 (0 = 512) */
  int32_t c1_mt, c1_waves, c1_wgs;  /* conv1x1_stream: pixel tiles per wave step, waves per workgroup, workgroup cap (0 = auto) */
  int32_t pair;            /* upa_bottleneck_pair: 0 = C = 32 only (default), 1 = never, 2 = C = 32 and 64, 3 = C = 64 only */
  int32_t pair_tile64, pair_tile32;  /* its square output tile edge per width (0 = auto) */
  int32_t no_pair_cv2;     /* upa_bottleneck_pair_cv2: 1 = never */
  int32_t c2f;             /* upa_c2f_fused / upa_c2f64_fused / upa_c2f32_up_fused: 0 = every form, 1 = never, 2 = not the 16-wide, 3 = not the 32-wide, 4 = not the 64-wide, 6 = the 64-wide form only for n = 1 blocks (A/B: 52.3 k vs 52.6 k images/s in flight with 4) */
  int32_t c2f16_waves;     /* C2f(32, 32, n = 1): 0 = the line-buffer form (csrc/c2f16_stream.hip) | 4 | 8 = the 16 x 16 tile form with that many waves per workgroup (A/B) */
  int32_t c2f32_th;        /* output tile rows of the C2f(64, 64, n = 2) form: 0 = 16 | 10 */
  int32_t no_branch_tail;  /* upa_detect_branch_tail: 1 = never */
  int32_t branch_tail_bm;  /* its workgroup pixels: 0 = auto | 128 | 256 */
  int32_t stem_wgs, stemf_wgs, stemf_waves, stem_no_mfma;  /* stem kernels: workgroup caps (0 = 1024 / 512), fused-stem waves (0 = 8 | 4) */
  int32_t ablate_conv, ablate_pipe, ablate_c1, ablate_stem;  /* kernel ablation bit masks: honoured by the -DUPA_ABLATE build only (make ablate) */
  int32_t c2f64_max_px;    /* upa_c2f64_fused only up to this many pixels n * h * w (0 = 100000: the 40 x 40 maps at batch 32; -1 = any size) */
  int32_t conv_ws3;        /* csrc/conv_ws3.hip (persistent 3x3 with register-resident weights, Cin <= 64, Cout 64): 0 = by the size rule, 1 = never, 2 = every shape it can run, 3 = the rule restricted to maps of < 100000 pixels (at most one tile per workgroup) */
  int32_t no_group;        /* upa_conv2d_bias_act_group / upa_detect_branch_tail_group: 0 = two problems per grid where the instantiations allow, 1 = one launch per problem, 2 = three per grid too (measured slower; A/B) */
  int32_t no_c2f32_up;     /* 1 = upa_c2f32_up_fused refuses (the block then runs as upa_conv1x1_upcat + upa_bottleneck_pair_cv2; A/B) */
  int32_t conv_mm;         /* csrc/conv_mm.hip (4-wave 32x32x16-MFMA 3x3 kernel for Cin % 64 == 0, Cout % 128 == 0) inside upa_conv2d_bias_act: 2 = every shape it can run (experiment: measured slower than conv_big, see its header), 0 / 1 = never */
  int32_t no_xcd;          /* 1 = tile kernels take tile blockIdx.x instead of the XCD-aware order (each XCD a contiguous tile range: neighbouring halos meet in one L2); A/B */
  int32_t keys_only;       /* upa_detect_branch_tail* / upa_detect_head_tails with best_keys: 1 = the class rows of y are NOT written - only the boxes and the best-class NMS keys, which is all single-label NMS reads (upa_nms_batched_hot); rows 4.. of y are then undefined */
  int32_t conv_p8;         /* csrc/conv_p8.hip (8-wave two-group phased 3x3 stride-1 kernel for Cin % 64 == 0, Cout % 128 == 0: counted vmcnt, 4-slab weight ring, double-buffered halo) inside upa_conv2d_bias_act: 0 = by the size rule, 1 = never, 2 = every shape it can run */
  int32_t c2f_stream;      /* csrc/c2f_stream.hip (line-buffer form of the C2f(64, 64, n = 2) block: fixed wave roles, LDS ring buffers) inside upa_c2f_fused: 0 = where it applies, 1 = never (the 16 x 16 tile form), 2 = the n = 2 block on its first wave-role set (A/B) */
  int32_t c2f_stream_rows; /* its output rows per workgroup: 0 = auto (one round of workgroups where possible) | even >= 4 | -1 = the whole image height (fewest pipeline fills: least total CU time, for several steps in flight) */
  int32_t no_stack_first;  /* Detect (host side, nn/modules/head.py): 1 = the two first convs of the 80 x 80 level stay two problems of one grid instead of one stacked 144-channel convolution; A/B */
  int32_t no_epi_stats;    /* upa_conv2d_bn_stats: 1 = the batch statistics always by a reduction pass over z (upa_bn_stats), never from the convolution's own workgroups; A/B */
  int32_t nms_stages;      /* upa_nms_batched_opts, multi-label lists longer than 16384 (validation): 0 = prefix keys only (histogram + emit kernels), stages ~nms_first_prefix / ~16384 / exact | 1 = all keys written up front, the sort kernel's pass picks the prefix from the histogram | 2 = no histogram: radix select of the top 16384, then exact (the round-4 form); results identical, A/B */
  int32_t nms_first_prefix; /* target length of the first sorted prefix there: 0 = 4096 | n in [256, 16384) | -1 = none (first prefix ~16384) */
  int32_t detect_stream;   /* upa_detect_level_stream (csrc/detect_stream.hip: one Detect level, both branches, as one line-buffer launch): 0 / 1 = refuse (callers run the tile form: stacked first conv + upa_detect_head_tails - the library default: faster launch for launch), 2 = run wherever the form applies (what the throughput runner asks for with several steps in flight: less CU time and traffic, +2.4 % images/s, profiles/r06_detect_stream.txt) */
  int32_t detect_stream_rows; /* its output rows per workgroup: 0 = the whole image height (one workgroup per strip and branch: least total CU time) | even >= 4 (more, shorter workgroups: a lower latency with one step at a time) */
  int32_t no_sppf_front;   /* 1 = upa_sppf_front refuses (SPPF then runs cv1 and the pools as two launches; A/B) */
  int32_t no_c2f16_down;   /* 1 = upa_c2f16_down_fused refuses (the C2f(32, 32, n = 1) block and the stride-2 Conv behind it then run as two launches; A/B) */
} upa_opts;

/* Library / device info. Returns the ABI version (int); fills name with the kernel target ("gfx950"). */
int upa_version(void);
const char* upa_last_error(void);
size_t upa_opts_size(void); /* sizeof(upa_opts) the library was built with (bindings check their mirror of the struct) */

/* ---- convolution -------------------------------------------------------------------------------------------------
 * y = act(conv2d(x, W) + bias) [+ residual]          nn/modules/conv.py:188-197 (Conv.forward_fuse), block.py:668
 * Weights are pre-packed by upa_pack_conv_weight (BN already folded: utils/torch_utils.py:236-266).
 * Implicit GEMM on MFMA, input halo tile staged in LDS; groups=1, dilation=1, square kernel k in [1,7]. */
size_t upa_conv_packed_weight_bytes(int cout, int cin, int k, int dtype);
/* Host-side packing: w is OIHW f32 (cout,cin,k,k) in HOST memory; out is HOST memory of the size above. */
int upa_pack_conv_weight(const float* w_oihw, int cout, int cin, int k, int dtype, void* out);
int upa_conv2d_bias_act(const void* x, int n, int h, int w, int cin, int ldx,
                        const void* w_packed, const float* bias /* f32[cout padded to 16] or NULL */,
                        void* y, int cout, int ldy,
                        const void* residual /* NULL or view shaped like y */, int ldr,
                        int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream);
/* Several INDEPENDENT convolutions that share k / stride / pad / act / dtype (the first convs of a Detect head's branches on
 * different levels): the same results as one upa_conv2d_bias_act per problem; neighbours in the list that land on the same
 * 128-pixel large-tile instantiation share one grid.  conv.py:188-197 */
typedef struct upa_conv_problem {
  const void* x; int32_t n, h, w, cin, ldx;
  const void* w_packed; const float* bias;
  void* y; int32_t cout, ldy;
  const void* residual; int32_t ldr;
} upa_conv_problem;
int upa_conv2d_bias_act_group(const upa_conv_problem* probs, int count, int k, int stride, int pad, int act, int dtype,
                              const upa_opts* opts, void* stream);

/* Conv(k 3, s 1, p 1) + SiLU followed by nn.MaxPool2d(2, 2, 0) as one launch (cfg/models/v3/Detect/yolov3-tiny.yaml:12-19; conv.py:188-197
 * + torch.nn.MaxPool2d): y = the pooled (n, h / 2, w / 2, cout) view; the full-resolution activation is never written.  Bit-identical
 * to upa_conv2d_bias_act + upa_maxpool2d.  UPA_EUNSUPPORTED outside the fused form (bf16, h % 8 == 0, w % 16 == 0, cin <= 128,
 * cout % 32 == 0): the caller runs the two layers.  upa_conv2d_stem_nchw_pool2: the same for the first layer (NCHW / uint8 input). */
int upa_conv2d_pool2(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, const float* bias, void* y, int cout,
                     int ldy, int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream);
int upa_conv2d_stem_nchw_pool2(const void* x_nchw, int x_dtype, int n, int cin, int h, int w, const float* w_oihw, const float* bias,
                               void* y, int cout, int ldy, int k, int stride, int pad, int act, int dtype, const upa_opts* opts,
                               void* stream);

/* Introspection for benchmarks: the kernel instantiation (WM<<12 | WN<<8 | MTW<<4 | NTW) the call above would use. */
int upa_conv_variant(int n, int h, int w, int cin, int cout, int k, int stride, int pad, int dtype, const upa_opts* opts);

/* Bottleneck as one kernel (nn/modules/block.py:644-668 with k = (3, 3), e = 1.0 as C2f builds it, block.py:475):
 * y = [x +] SiLU(conv3x3(SiLU(conv3x3(x, w1) + b1), w2) + b2), C -> C -> C channels, BN folded, the intermediate tile kept
 * in LDS (csrc/conv_pair.hip).  x / y: NHWC views (n, h, w, c) with row strides ldx / ldy (channel slices of a C2f concat
 * buffer).  Returns UPA_EUNSUPPORTED outside the fused form (bf16, SiLU, c = 32 | 64): the caller then runs two
 * upa_conv2d_bias_act launches. */
int upa_bottleneck_pair(const void* x, int n, int h, int w, int c, int ldx, const void* w1_packed, const float* b1,
                        const void* w2_packed, const float* b2, void* y, int ldy, int residual, int act, int dtype,
                        const upa_opts* opts, void* stream);
/* The same with the hidden width given: cmid = c (above) or c / 2 for c = 64 - the e = 0.5 Bottleneck of the darknet backbones
 * (block.py:644-668 with the default e; cfg/models/v3/Detect/yolov3-rtdetr.yaml: Bottleneck(64) at 320 x 320).
 * w1_packed = upa_pack_conv_weight(c -> cmid, k = 3), w2_packed = (cmid -> c, k = 3). */
int upa_bottleneck_pair_e(const void* x, int n, int h, int w, int c, int cmid, int ldx, const void* w1_packed, const float* b1,
                          const void* w2_packed, const float* b2, void* y, int ldy, int residual, int act, int dtype,
                          const upa_opts* opts, void* stream);


/* First layer: reads the model input NCHW (f32 or bf16, 1..4 channels) directly, writes NHWC.   conv.py:188-197
 * w = device copy of the upa_pack_stem_weight output ([tap][ci][cout padded to 16] f32), bias f32[cout] on the device. */
size_t upa_stem_packed_weight_bytes(int cout, int cin, int k);
int upa_pack_stem_weight(const float* w_oihw, int cout, int cin, int k, float* out /* host */);
int upa_conv2d_stem_nchw(const void* x_nchw, int x_dtype, int n, int cin, int h, int w,
                         const float* w_oihw, const float* bias, void* y, int cout, int ldy,
                         int k, int stride, int pad, int act, int dtype, const upa_opts* opts, void* stream);

/* Conv(3,16,3,2,1)+SiLU -> Conv(16,32,3,2,1)+SiLU fused (yolov8n rows 0-1, cfg/models/v8/yolov8.yaml): bf16 NCHW input
 * (w % 8 == 0, h % 4 == 0), the 16-channel stem output only ever exists as an LDS tile.  w0 / b0: upa_pack_stem_weight
 * layout + folded bias; w1 / b1: upa_pack_conv_weight(UPA_BF16) layout + folded bias; y: NHWC bf16 (n, h/4, w/4, 32). */
int upa_stem_conv_fused(const void* x, int n, int h, int w, const float* w0, const float* b0, const void* w1,
                        const float* b1, void* y, int ldy, const upa_opts* opts, void* stream);
/* ... with the first conv's kernel size given: k0 = 3 (pad 1) | 6 (pad 2: yolov5's Conv(3, 16, 6, 2, 2), cfg/models/v5/Detect/yolov5-BoT3.yaml:15) */
int upa_stem_conv_fused_k(const void* x, int n, int h, int w, int k0, const float* w0, const float* b0, const void* w1,
                          const float* b1, void* y, int ldy, const upa_opts* opts, void* stream);
/* The same with the first conv's channel count given: c0 = 16 (-> 32 channels out, k0 = 3 | 6) or 32 (-> 64, k0 = 3: yolov8s rows 0-1,
 * cfg/models/v8/yolov8.yaml:18-19 at width 0.50). */
int upa_stem_conv_fused_c(const void* x, int n, int h, int w, int k0, int c0, const float* w0, const float* b0, const void* w1,
                          const float* b1, void* y, int ldy, const upa_opts* opts, void* stream);
/* ... and the first conv's stride: s0 = 2, or 1 for the 3 -> 32 -> 64 form (darknet53's Conv(3, 32, 3, 1) -> Conv(32, 64, 3, 2): yolov3-rtdetr
 * rows 0-1, cfg/models/v3/Detect/yolov3-rtdetr.yaml).  y: NHWC bf16 view (n, h / (2 s0), w / (2 s0), 2 * c0). */
int upa_stem_conv_fused_s(const void* x, int n, int h, int w, int k0, int s0, int c0, const float* w0, const float* b0,
                          const void* w1, const float* b1, void* y, int ldy, const upa_opts* opts, void* stream);

/* ---- pooling / resampling / concat (HBM-bound) --------------------------------------------------------------- */
/* nn.MaxPool2d(k, s, p) with -inf padding; pad_br>0 emulates nn.ZeroPad2d([0,pad_br,0,pad_br]) in front of it
 * (zeros, not -inf, take part in the max).                                   cfg yolov3-tiny.yaml rows 1-12 */
int upa_maxpool2d(const void* x, int n, int h, int w, int c, int ldx, void* y, int oh, int ow, int ldy,
                  int k, int stride, int pad, int pad_br, int dtype, void* stream);
/* SPPF pooling chain: y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) (== 5/9/13 windows)          block.py:402-406 */
int upa_sppf_pool3(const void* x, int n, int h, int w, int c, int ldx, void* y1, void* y2, void* y3, int ldy,
                   int dtype, void* stream);
/* SPPF front (block.py:382-406): y[:, 0:c_) = SiLU(cv1(x)) - cv1 = Conv(c1, c_, 1, 1), BN folded, w_packed / bias from upa_pack_conv_weight(bf16) - and
 * y[:, c_:4c_) = the three chained 5x5 pools of it, as ONE launch; y = the (n, h, w, >= 4 c_) concat buffer cv2 reads.  bf16, c1 = 128 | 256 | 512,
 * c_ % 16 == 0, h * w <= 1024; UPA_EUNSUPPORTED otherwise (callers run upa_conv2d_bias_act + upa_sppf_pool3). */
int upa_sppf_front(const void* x, int n, int h, int w, int c1, int ldx, const void* w_packed, const float* bias, void* y, int c_, int ldy, int dtype,
                   const upa_opts* opts, void* stream);
/* nn.Upsample(scale 2, nearest) written into a channel slice (fused Upsample+Concat)       conv.py:850-875 */
int upa_upsample2x(const void* x, int n, int h, int w, int c, int ldx, void* y, int ldy, int dtype, void* stream);
/* view -> view copy (Concat of tensors that could not be produced in place)                conv.py:874 */
int upa_copy_view(const void* x, int n, int h, int w, int c, int ldx, void* y, int ldy, int dtype, void* stream);
/* device -> PINNED host copy issued as a kernel (capturable into a hipGraph without a memcpy node): the hand-over of the
 * fixed-shape detections (B, max_det, 6) + counts to the host at the end of a step - the reference's results leave the GPU
 * in `Results(...)` construction, engine/results.py via models/yolo/detect/predict.py:53-120.  bytes % 4 == 0. */
int upa_copy_to_host(const void* src_dev, void* dst_pinned, size_t bytes, void* stream);
/* The same hand-over in ONE launch for the NMS / RT-DETR result layout: counts (batch) int32 and, per image, only the first
 * counts[b] of its max_rows rows of row_bytes (6 floats: xyxy, conf, cls) go to pinned host memory; rows past the count are not
 * transferred.  results.py (`Results.boxes` on the CPU), predict.py:53-120. */
int upa_results_to_host(const void* rows_dev, const int* counts_dev, int batch, int max_rows, int row_bytes,
                        void* rows_pinned, int* counts_pinned, void* stream);
/* y = a + b (views)                                                                        block.py:6091 */
int upa_add_view(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int n, int h, int w, int c,
                 int dtype, void* stream);
/* layout conversion at the module boundary (NCHW f32 <-> NHWC dtype).  Narrow inputs (c <= 16 / elem size, e.g. an image
 * batch) whose rows are 16-byte aligned (ldy a multiple of 16 / elem size) are written as one 16-byte group per pixel:
 * channels c .. 16/elem-1 of that group are set to zero (the channel padding the first conv expects). */
int upa_nchw_to_nhwc(const float* x, int n, int c, int h, int w, void* y, int ldy, int dtype, void* stream);
int upa_nhwc_to_nchw(const void* x, int n, int h, int w, int c, int ldx, float* y, int dtype, void* stream);
/* LetterBox of uint8 (h0, w0, 3) frames (data/augment.py:1544-1700, the predictor's pre_transform, predictor.py:151-173):
 * cv2.resize(INTER_LINEAR) to (new_h, new_w) - OpenCV's 8-bit fixed-point arithmetic, bit-exact against oracle/letterbox.py -
 * placed at (top, left) of an (H, W, 3) output filled with pad_value.  The geometry (ratio, rounding, padding split) is the
 * caller's: ultralytics_pro_amd/data/augment.py computes it exactly as LetterBox.__call__ does.  n frames of one size per
 * call (src_image_stride bytes apart, rows src_row_stride bytes apart); dst is contiguous (n, H, W, 3) and feeds the stem
 * conv as UPA_U8_BGR_HWC. */
int upa_letterbox_u8(const void* src, int n, int h0, int w0, long src_image_stride, int src_row_stride, void* dst, int H, int W,
                     int new_h, int new_w, int top, int left, int pad_value, void* stream);


/* ---- Detect decode ------------------------------------------------------------------------------------------------
 * One level: box logits (n,h,w,4*reg_max) + cls logits (n,h,w,nc) NHWC -> y[b, 0:4, a0:a0+h*w] = xywh*stride
 * (DFL softmax-expectation, dist2bbox with cell-centre anchors), y[b, 4:4+nc, ...] = sigmoid(cls).
 * y is (n, 4+nc, a_total) f32.        head.py:151-169, block.py:250-253, utils/tal.py:352-376 */
int upa_detect_decode(const void* box, int ldb, const void* cls, int ldc, int n, int h, int w, int reg_max, int nc,
                      float stride_px, float* y, int a_total, int a0, int dtype, void* stream);
/* The last 1x1 conv of one Detect branch (nn.Conv2d(c2, 4*reg_max, 1) / nn.Conv2d(c3, nc, 1), head.py:94-100) with that
 * branch's part of the decode fused on the end (bf16 mode): kind 1 = box (DFL + dist2bbox + stride -> y[b, 0:4, a0 + a]),
 * kind 2 = class (sigmoid -> y[b, 4:4+nc, a0 + a]); the logits never reach HBM unless `raw` (n,h,w,cout) bf16 rows with
 * stride ldraw is given (Detect's second return value, head.py:126).  cout = rows of the packed weights (>= nc: class
 * filters may be zero-padded to the 16-byte store width).  Returns UPA_EUNSUPPORTED outside the fused form (f32 parity
 * mode, reg_max != 16, nc > 128): the caller then runs upa_conv2d_bias_act + upa_detect_decode.  head.py:151-169
 * best_keys (kind 2, may be NULL): (n, a_total) u64 - the NMS sort key of every anchor's best class, as upa_detect_branch_tail
 * writes it (the NMS prefilter of upa_nms_batched_hot for class branches of any width the streaming kernel takes). */
int upa_detect_tail(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, const float* bias, int cout,
                    int kind, int nc, float stride_px, float* y, int a_total, int a0, void* raw, int ldraw,
                    unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream);

/* 1x1 conv over Concat([Upsample(2x nearest)(up), skip]) with the upsample read on the fly (bf16): the first up_c channels of a
 * pixel come from pixel (y/2, x/2) of `up` (n, h/2, w/2, up_c; pixel stride up_ld), the other cin - up_c from the concat buffer x
 * (n, h, w, cin) where the skip producer wrote them in place; the upsampled copy is never materialised.
 * nn.Upsample + Concat + C2f.cv1: yolov8.yaml rows 10-12 / 13-15, nn/modules/conv.py (Concat), block.py:479.
 * UPA_EUNSUPPORTED outside the streaming 1x1 form (callers then write the upsample and call upa_conv2d_bias_act). */
int upa_conv1x1_upcat(const void* x, int n, int h, int w, int cin, int ldx, const void* up, int up_c, int up_ld,
                      const void* w_packed, const float* bias, void* y, int cout, int ldy, int act, int dtype,
                      const upa_opts* opts, void* stream);

/* C2f(.., 64, n = 1) with a 32-channel Bottleneck (yolov8n model.15): the Bottleneck (both 3x3 convs [+ shortcut]) AND the
 * C2f's cv2 in one launch (bf16, SiLU); the Bottleneck's output only exists as MFMA operands.   block.py:457-488, 644-668.
 * x = the y1 slice (channels [32, 64)) of the C2f concat buffer, y0 = its y0 slice (channels [0, 32)), same pixel stride ldx;
 * w1 / w2 = the Bottleneck's convs (upa_pack_conv_weight); wc_std = cv2's input columns [0, 64) as
 * upa_pack_conv_weight(64 -> 64, k = 1), wc_b = columns [64, 96) as upa_pack_tail_weight(64, 32), bc = cv2's bias (BN folded);
 * out = (n, h, w, 64) view.  UPA_EUNSUPPORTED outside the form (callers run upa_bottleneck_pair + upa_conv2d_bias_act). */
int upa_bottleneck_pair_cv2(const void* x, const void* y0, int n, int h, int w, int ldx, const void* w1_packed, const float* b1,
                            const void* w2_packed, const float* b2, int residual, const void* wc_std, const void* wc_b,
                            const float* bc, void* out, int ldout, int act, int dtype, const upa_opts* opts, void* stream);

/* A whole C2f block in one launch (bf16, SiLU): cv1 -> nb x Bottleneck(3x3, 3x3, [+ input]) -> cv2 with the intermediates in
 * LDS / registers only.                            nn/modules/block.py:457-488 (C2f.forward), :644-668 (Bottleneck.forward).
 * c1 / c2: channels in / out, c: hidden width (c2 * e), nb: Bottlenecks; w1,b1 = cv1; wm[2i], wm[2i+1] (bm likewise) =
 * m[i].cv1, m[i].cv2; w2,b2 = cv2 - all upa_pack_conv_weight(bf16) with BN folded; wm / bm are HOST arrays of 2 nb device
 * pointers.  Fused forms: (c1, c, c2, nb) = (32, 16, 32, 1) with shortcut (yolov8n model.2) and (64, 32, 64, 1 | 2) with or
 * without shortcut (yolov8n model.4, yolov8s model.2); UPA_EUNSUPPORTED otherwise (callers run the separate convolutions). */
int upa_c2f_fused(const void* x, int n, int h, int w, int c1, int ldx, int c, int nb, int shortcut, const void* w1,
                  const float* b1, const void* const* wm, const float* const* bm, const void* w2, const float* b2, void* y,
                  int c2, int ldy, int act, int dtype, const upa_opts* opts, void* stream);

/* C2f(32, 32, n = 1, shortcut) AND the Conv(32, 64, 3, 2) row behind it as ONE line-buffer launch (csrc/c2f16_stream.hip; yolov8n rows 2-3,
 * cfg/models/v8/yolov8.yaml:18-19): x (n, h, w, 32) -> y (n, ceil(h / 2), ceil(w / 2), 64); the block's own output only ever exists as eight rows
 * of a strip in LDS.  wd / bd = the stride-2 Conv (upa_pack_conv_weight(bf16), BN folded, SiLU); the other arguments as upa_c2f_fused.
 * UPA_EUNSUPPORTED outside the form (callers run upa_c2f_fused, then upa_conv2d_bias_act).   block.py:457-488, :644-668, conv.py:188-197 */
int upa_c2f16_down_fused(const void* x, int n, int h, int w, int ldx, const void* w1, const float* b1, const void* const* wm,
                         const float* const* bm, const void* w2, const float* b2, const void* wd, const float* bd, void* y, int ldy,
                         int dtype, const upa_opts* opts, void* stream);

/* The same for the 64-channel-half blocks (csrc/c2f64.hip): C2f(c1 -> 128, c = 64, n = nb in {1, 2}), c1 % 64 == 0 - yolov8n
 * model.6 / model.12 / model.18 at 40 x 40, yolov8s model.4 / model.15 at 80 x 80.  `up` (may be NULL): a (n, h/2, w/2, up_c) tensor
 * holding the first up_c (% 64 == 0) channels of every pixel at half resolution - the nn.Upsample(2x nearest) + Concat in front
 * of the block read on the fly (yolov8.yaml rows 10-12), x's first up_c channels are then never read.   block.py:457-488, :644-668 */
int upa_c2f64_fused(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int nb, int shortcut,
                    const void* w1, const float* b1, const void* const* wm, const float* const* bm, const void* w2,
                    const float* b2, void* y, int c2, int ldy, int act, int dtype, const upa_opts* opts, void* stream);
/* C2f(c1, 64, n = 1) with 32-channel halves and c1 = 64 k >= 128 input channels (yolov8n model.15: C2f(192, 64) behind nn.Upsample +
 * Concat) as one launch: cv1 streamed over 64-channel chunks, the first up_c channels read from the half-resolution tensor `up`
 * (NULL: everything from x).  Arguments as upa_c2f64_fused; UPA_EUNSUPPORTED outside the form.  block.py:457-488, conv.py:850-875 */
int upa_c2f32_up_fused(const void* x, int n, int h, int w, int c1, int ldx, const void* up, int up_c, int up_ld, int nb, int shortcut,
                       const void* w1, const float* b1, const void* const* wm, const float* const* bm, const void* w2, const float* b2,
                       void* y, int c2, int ldy, int act, int dtype, const upa_opts* opts, void* stream);

/* The whole back half of a Detect branch in one launch (bf16): second 3x3 conv (BN + SiLU folded) -> final 1x1 conv -> that
 * branch's half of the decode, the intermediate maps never leaving the registers.      head.py:94-100 (cv2/cv3), :116-126,
 * :151-169.  kind 1 = box branch (c = 64 = 4 * reg_max 16), 2 = class branch (c <= 96, nc <= 96).  With CP = 64 (box), 80 (class
 * branch with c = 80) or 96 (class branch, any other c):
 * w3_packed / b3 = upa_pack_conv_weight of the 3x3 conv as c -> CP (zero filters and biases appended);
 * wt_packed = upa_pack_tail_weight(cout = CP, cin = CP) of the zero-padded 1x1 matrix, bt = CP biases.
 * Returns UPA_EUNSUPPORTED outside that form (callers then run conv2d + upa_detect_tail). */
size_t upa_tail_packed_weight_bytes(int cout, int cin);
int upa_pack_tail_weight(const float* w, int cout, int cin, void* out);
/* best_keys (class branch, may be NULL): NMS prefilter - the NMS sort key of every anchor's best class,
 * ~bits(score) << 32 | anchor * nc + class (class = first maximum, nms.py:109), as a dense (B, a_total) array.
 * Consumed by upa_nms_batched_hot. */
int upa_detect_branch_tail(const void* x, int n, int h, int w, int c, int ldx, const void* w3_packed, const float* b3,
                           const void* wt_packed, const float* bt, int kind, int nc, float stride_px, float* y, int a_total,
                           int a0, unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream);
/* The same for several levels of ONE Detect head (same kind, shared y / best_keys): identical results, but levels whose problems
 * land on the same 128-pixel kernel instantiation are launched two per grid (the 40 x 40 and 20 x 20 levels at batch 32: 400 + 100
 * workgroups, one partial round instead of two launches).  UPA_EUNSUPPORTED (nothing launched) if any level is outside the fused
 * form: the caller then goes level by level.  head.py:94-100, 116-126, 151-169 */
typedef struct upa_branch_level {
  const void* x; int32_t n, h, w, c, ldx;                 /* the branch's mid tensor (NHWC view) */
  const void* w3_packed; const float* b3;                 /* second 3x3 conv (as upa_detect_branch_tail) */
  const void* wt_packed; const float* bt;                 /* final 1x1 conv (upa_pack_tail_weight) */
  float stride_px; int32_t a0;                            /* the level's stride and first anchor */
} upa_branch_level;
int upa_detect_branch_tail_group(const upa_branch_level* levels, int count, int kind, int nc, float* y, int a_total,
                                 unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream);
/* Both kinds at once: box[i] and cls[i] = the two branch tails of level i.  As two upa_detect_branch_tail_group calls, but box and
 * class problems share grids as well (two kernel instantiations per grid): the six tails of a three-level head are two launches. */
int upa_detect_head_tails(const upa_branch_level* box, const upa_branch_level* cls, int count, int nc, float* y, int a_total,
                          unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream);
/* ONE level of a Detect head, both branches and all three convolutions of each, in one launch (bf16; csrc/detect_stream.hip):
 * cv2[i] = Conv(cin, 64, 3) -> Conv(64, 64, 3) -> Conv2d(64, 4 * 16, 1) -> DFL + dist2bbox * stride -> y[b, 0:4, a0 + a],
 * cv3[i] = Conv(cin, 80, 3) -> Conv(80, 80, 3) -> Conv2d(80, nc, 1) -> sigmoid -> y[b, 4:4+nc, a0 + a] (+ best-class NMS keys),
 * the intermediate maps only ever existing as a few rows in LDS (line-buffer form: a workgroup streams down a 30-column strip of one
 * image, fixed wave roles, weights in registers).                     head.py:94-100 (cv2 / cv3), :116-126 (forward), :151-191 (_inference)
 * A branch = its channel count c and three upa_pack_conv_weight(UPA_BF16) blobs with BN folded: w1 = 3x3 cin -> c, w2 = 3x3 c -> c,
 * wt = 1x1 c -> 64 (box) | 80 (class, zero filters beyond nc), biases f32 padded to a multiple of 16.
 * Form: cin = 64, box c = 64, class c = 80, nc <= 80 (yolov8n's 80 x 80 level).  UPA_EUNSUPPORTED otherwise (callers run
 * upa_conv2d_bias_act_group + upa_detect_head_tails).  best_keys / upa_opts.keys_only as upa_detect_branch_tail. */
typedef struct upa_detect_branch {
  int32_t c, reserved;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  const void* wt; const float* bt;
} upa_detect_branch;
int upa_detect_level_stream(const void* x, int n, int h, int w, int cin, int ldx, const upa_detect_branch* box,
                            const upa_detect_branch* cls, int nc, float stride_px, float* y, int a_total, int a0,
                            unsigned long long* best_keys, int dtype, const upa_opts* opts, void* stream);


/* ---- NMS ----------------------------------------------------------------------------------------------------------
 * Batched non_max_suppression over pred (b, 4+nc, a) f32 xywh+scores -> out (b, max_det, 6) f32
 * [x1,y1,x2,y2,conf,cls], counts int32[b], keep_idx int32 (b, max_det) anchor indices (may be NULL).
 * classes_mask: NULL or uint8[nc] (1 = keep class).  Greedy hard NMS, class offset cls*max_wh added in f32,
 * IoU without eps, survivor iff IoU <= thr, score-descending (ties: lower candidate index first).
 *                                                                  utils/nms.py:13-166, :239-296 */
size_t upa_nms_workspace_bytes(int b, int nc, int a, int multi_label, int max_nms);
/* upa_nms_batched (single-label form: multi_label = 0) with the candidate scan over the (B, A) best-class keys a Detect class
 * branch wrote next to pred (upa_detect_branch_tail) instead of the (B, nc, A) scores: identical results, 40x fewer bytes. */
int upa_nms_batched_hot(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                        int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh, float* out,
                        int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                        const unsigned long long* best_keys, void* stream);
/* upa_nms_batched with dispatch options (nms_stages, nms_first_prefix: how long multi-label lists are staged; NULL = defaults). */
int upa_nms_batched_opts(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                         int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh, float* out,
                         int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes, const upa_opts* opts,
                         void* stream);
int upa_nms_batched(const float* pred, int b, int nc, int a, float conf_thres, float iou_thres, int multi_label,
                    int agnostic, const uint8_t* classes_mask, int max_det, int max_nms, float max_wh,
                    float* out, int32_t* counts, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ---- attention (BoT3, RT-DETR self-attention) ---------------------------------------------------------------------
 * Dense multi-head attention core: q,k,v are row views (n*hw rows, row stride ldqkv elements) holding heads*d channels;
 * energy = (scale*q)^T k, softmax over keys, out = v.attn^T; optional residual add.
 * scale = 1 for MHSA (block.py:6036-6091, unscaled), 1/sqrt(d) for nn.MultiheadAttention (transformer.py:670-673). */
int upa_mhsa(const void* q, const void* k, const void* v, int ldqkv, int n, int hw, int heads, int d, float scale,
             const void* residual, int ldr, void* y, int ldy, int dtype, void* stream);

/* ---- RT-DETR decoder head pieces (f32 token rows) -------------------------------------------------------------------
 * nn.Linear: y[m,n] = act(x[m,k] . W[n,k]^T + bias[n]) (+ residual); W packed by upa_pack_conv_weight(k=1, UPA_F32).
 *                                                                     transformer.py:348-399, head.py:1993-2003 */
int upa_linear(const float* x, long m, int k, int ldx, const void* w_packed, const float* bias, float* y, int n, int ldy,
               const float* residual, int ldr, int act, void* stream);
/* The same with the PRODUCT on the bf16 matrix cores (perf mode of the RT-DETR decoder, = the reference's half-precision predict where
 * every nn.Linear multiplies 16-bit operands: engine/predictor.py:151-173 `model.half()`): float32 rows in and out, every element of x
 * rounded to bf16 on the way into the MFMA, w_packed = upa_pack_conv_weight(UPA_BF16) of the (n, k, 1, 1) weight, float32
 * accumulation / bias / activation / residual.  UPA_EUNSUPPORTED outside the form (k % 32 == 0, k <= 1024, n % 4 == 0, 16-byte rows). */
int upa_linear_bf16(const float* x, long m, int k, int ldx, const void* w_packed, const float* bias, float* y, int n,
                    int ldy, const float* residual, int ldr, int act, void* stream);
/* ... and with the row types given: x float32 (rounded into the MFMA) or bf16 (k % 64 == 0), y float32 or bf16 - q, k, v of the decoder's
 * self-attention are written as bf16 rows for the matrix-core attention kernel (upa_mhsa), whose bf16 output feeds out_proj. */
int upa_linear_mixed(const void* x, int x_dtype, long m, int k, int ldx, const void* w_packed, const float* bias, void* y,
                     int y_dtype, int n, int ldy, const float* residual, int ldr, int act, void* stream);
/* y = LayerNorm(x (+ residual))                                                  transformer.py:660-685 */
int upa_layer_norm(const float* x, const float* residual, int m, int c, const float* gamma, const float* beta, float eps,
                   float* y, void* stream);
int upa_rows_add(const float* a, const float* b, float* y, long m, int c, void* stream);
int upa_rows_scale(const float* x, const float* row_scale, float* y, long m, int c, void* stream);   /* head.py:2169 */
int upa_rows_gather(const float* x, const int32_t* row_idx, float* y, long m_out, int c, void* stream); /* head.py:2180 */
/* Query selection: per image top-k tokens by max class logit, sorted descending (ties: lower token index first).
 * Token rows are level-major: row = sum_{l'<l} b*hw[l'] + img*hw[l] + p.  out_tok = reference token index.  head.py:2175 */
int upa_topk_tokens(const float* scores, int nc, int n_levels, const int32_t* level_hw, int b, int k, int32_t* out_rows,
                    int32_t* out_tok, void* stream);
/* y = sigmoid(delta + inverse_sigmoid(ref, eps=1e-5))            transformer.py:756-757, nn/modules/utils.py:79-100 */
int upa_box_refine(const float* delta, const float* ref, float* y, long n_boxes, void* stream);
/* y = delta + anchors[tok] (logit-space reference boxes)                                       head.py:2183 */
int upa_box_add_anchors(const float* delta, const int32_t* tok, const float* anchors, float* y, long n_boxes, void* stream);
int upa_sigmoid(const float* x, float* y, long n, void* stream);
/* y[m] = [boxes[m] (4) | sigmoid(scores[m]) (nc)]                                              head.py:2074 */
int upa_rtdetr_output(const float* boxes, const float* scores, float* y, long m, int nc, void* stream);
/* RTDETRPredictor.postprocess (models/rtdetr/predict.py:35-74) for the whole batch: preds (b, q, 4+nc) normalised
 * cxcywh | class scores -> out (b, max_det, 6) = [x1, y1, x2, y2, score, cls] sorted by score (ties: lower query first),
 * rows >= counts[b] are left untouched.  Boxes are scaled by the image size: orig_wh (b, 2) = (width, height) per image in
 * device memory, or NULL to use (ow, oh) for every image.  classes_mask (nc bytes, nullable) = `classes` filter.
 * q <= 1024.  Bit-exact with the reference arithmetic. */
int upa_rtdetr_postprocess(const float* preds, int b, int q, int nc, float conf, const unsigned char* classes_mask, int max_det,
                           const float* orig_wh, float ow, float oh, float* out, int32_t* counts, void* stream);
/* Multi-scale deformable attention sampling: value rows level-major (heads*d wide), offsets (b*len_q, heads*L*P*2),
 * attention logits (b*len_q, heads*L*P) (softmax applied here), 4-d reference boxes (b*len_q, 4) in [0,1];
 * bilinear, zero padding, align_corners=False.                   nn/modules/utils.py:103-159, transformer.py:540-556 */
int upa_msdeform_attn(const float* value, const int32_t* shapes_hw, int n_levels, int b, int heads, int d,
                      const float* offsets, const float* attn_logits, const float* ref_boxes, int len_q, int n_points,
                      float* y, void* stream);
/* The same with the projected values given as rows of `value_dtype` (UPA_F32 | UPA_BF16) with row stride ldv elements: the
 * value projections of ALL decoder layers computed as one GEMM (rows, n_layers * C) - layer i passes value + i * C. */
int upa_msdeform_attn_strided(const void* value, int value_dtype, int ldv, const int32_t* shapes_hw, int n_levels, int b, int heads,
                              int d, const float* offsets, const float* attn_logits, const float* ref_boxes, int len_q,
                              int n_points, float* y, void* stream);


/* ---- validation ---------------------------------------------------------------------------------------------------
 * out[i,j] = IoU(box1[i], box2[j]) with eps in the denominator (xyxy f32).            utils/metrics.py:54-74 */
int upa_box_iou(const float* box1, int n, const float* box2, int m, float eps, float* out, void* stream);

/* In-place scale_boxes + clip_boxes on xyxy rows: box -= (pad_x, pad_y, pad_x, pad_y); box /= gain; clip to (w0, h0).
 * rows: n rows of `row_stride` floats, the box in the first 4.                              utils/ops.py:102-178 */
int upa_scale_boxes(float* rows, long n, int row_stride, float gain, float pad_x, float pad_y, int padding, float w0,
                    float h0, void* stream);
/* True-positive matrices of a whole batch: match_predictions (engine/validator.py:267-308, non-scipy branch) through
 * DetectionValidator._process_batch (models/yolo/detect/val.py:274-288) on the fixed-shape NMS outputs - det (b, max_det, 6)
 * rows [x1,y1,x2,y2,conf,cls] with counts (b,), gt (b, max_gt, 5) rows [cls,x1,y1,x2,y2] with ngt (b,) - at the n_thr = 10
 * IoU thresholds (a HOST array; torch.linspace(0.5, 0.95, 10), val.py:59).  tp: (b, max_det, 10) bytes, rows past counts
 * are zero.  IoU as utils/metrics.py:54-74. */
int upa_match_predictions(const float* det, const int* counts, int b, int max_det, const float* gt, const int* ngt, int max_gt,
                          const float* iou_thresholds, int n_thr, unsigned char* tp, void* stream);


/* ---- training step (BASELINE config 3) ------------------------------------------------------------
 * What the reference gets from torch autograd around Conv = conv2d -> BatchNorm2d(batch statistics) -> SiLU
 * (nn/modules/conv.py:177-186), the SPPF pools (nn/modules/block.py:402-406), nn.Upsample, v8DetectionLoss
 * (utils/loss.py:415-528) and the optimizer step (engine/trainer.py:674-682).  Activations and their gradients are NHWC
 * views (f32 or bf16), parameters / parameter gradients / statistics are f32.
 *
 * Device-side repack of OIHW f32 master weights into the MFMA fragment layout of upa_conv2d_bias_act.
 * transpose_flip = 1 packs V[ci][co][kh][kw] = W[co][ci][k-1-kh][k-1-kw]: the data gradient of a stride-1 conv is then
 * upa_conv2d_bias_act(dz, V, pad = k-1-p) (a stride-2 conv first goes through upa_dilate2x). */
int upa_pack_conv_weight_dev(const float* w_oihw, int cout, int cin, int k, int dtype, int transpose_flip, void* out,
                             void* stream);
/* The same for every conv of a model in ONE launch (the weights change once per optimizer step, trainer.py:674-682):
 * descs_dev = n descriptors in device memory, one per (conv, layout). */
typedef struct UpaPackDesc {
  const float* w_oihw;
  void* out;
  int cout, cin, k, dtype, transpose_flip, reserved;
} UpaPackDesc;
int upa_pack_conv_weights_batched(const UpaPackDesc* descs_dev, int n, void* stream);
/* Per-channel reductions run in two stages without atomics: per-block f64 partial sums, then a fixed-order combine.
 * `ws` = upa_channel_reduce_workspace_bytes(c) bytes of scratch shared by upa_bn_stats / upa_bn_finalize (which must
 * follow each other on one stream), upa_bn_act_bwd and upa_channel_sum.
 * Batch statistics -> mean / biased var + nn.BatchNorm2d running update (running_var takes the unbiased estimate;
 * running_* may be NULL). */
size_t upa_channel_reduce_workspace_bytes(int c);
int upa_bn_stats(const void* z, long npix, int c, int ldz, double* ws, int dtype, void* stream);
int upa_bn_finalize(const double* ws, long npix, int c, float momentum, float* mean, float* var, float* running_mean,
                    float* running_var, void* stream);
/* Training forward of Conv (conv.py:177-186 in train mode) up to the normalisation: z = conv2d(x) - no bias, no activation - AND the
 * batch statistics of z (mean, biased var, running update as upa_bn_finalize) in one call.  On the kernels that have a statistics
 * epilogue the per-channel sums are left by the convolution's own workgroups (of the rounded values they store; one f32 row per pixel
 * tile, added in a fixed order by a combine launch): z is not read back.  Other shapes: upa_conv2d_bias_act + upa_bn_stats +
 * upa_bn_finalize.  w_packed as upa_conv2d_bias_act; ws as above (upa_channel_reduce_workspace_bytes(cout)). */
int upa_conv2d_bn_stats(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, void* z, int cout, int ldz, int k,
                        int stride, int pad, float momentum, float* mean, float* var, float* running_mean, float* running_var,
                        double* ws, int dtype, const upa_opts* opts, void* stream);
/* ... followed by upa_bn_act_fwd (below) in the same call: the whole training forward of a Conv, one call per layer. */
int upa_conv2d_bn_act_fwd(const void* x, int n, int h, int w, int cin, int ldx, const void* w_packed, void* z, int cout, int ldz, int k,
                          int stride, int pad, float momentum, float* mean, float* var, float* running_mean, float* running_var,
                          const float* gamma, const float* beta, float eps, int act, void* y, int ldy, const void* residual, int ldr,
                          double* ws, int dtype, const upa_opts* opts, void* stream);
/* y = act(gamma * (z - mean) / sqrt(var + eps) + beta) (+ residual) */
int upa_bn_act_fwd(const void* z, long npix, int c, int ldz, const float* mean, const float* var, const float* gamma,
                   const float* beta, float eps, int act, void* y, int ldy, const void* residual, int ldr, int dtype,
                   void* stream);
/* Backward of the above (z saved from the forward): dgamma, dbeta (f32, optionally accumulated) and dz.
 * ws: upa_channel_reduce_workspace_bytes(c). */
int upa_bn_act_bwd(const void* z, const void* dy, long npix, int c, int ldz, int lddy, const float* mean, const float* var,
                   const float* gamma, const float* beta, float eps, int act, void* dz, int lddz, float* dgamma, float* dbeta,
                   int accumulate, double* ws, int dtype, void* stream);
/* The whole backward of a training-mode Conv with stride 1 in one call: upa_bn_act_bwd (dgamma / dbeta accumulated, dz written), the
 * weight gradient (accumulated into dw_oihw; on side_stream if not NULL, ordered behind dz by an event), and - unless w_packed_t is NULL -
 * the data gradient dx (+)= conv(dz, w_packed_t) with w_packed_t = the transposed + flipped packing (upa_pack_conv_weight_dev,
 * transpose_flip 1).  x: the layer's input view (n, h, w, cin); z / dy / dz: output-side views (n, oh, ow, cout). */
int upa_conv_bn_act_bwd(const void* x, int n, int h, int w, int cin, int ldx, const void* z, const void* dy, int cout, int ldz, int lddy,
                        const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, void* dz,
                        int lddz, float* dgamma, float* dbeta, double* ws, float* dw_oihw, void* wgrad_ws, size_t wgrad_ws_bytes,
                        void* side_stream, const void* w_packed_t, void* dx, int lddx, int accumulate_dx, int k, int pad, int dtype,
                        const upa_opts* opts, void* stream);
/* out[c] (+)= sum over rows of z[:, c]  (bias gradient of the plain nn.Conv2d head outputs). ws as above. */
int upa_channel_sum(const void* z, long npix, int c, int ldz, float* out, int accumulate, double* ws, int dtype, void* stream);
/* dW[co][ci][kh][kw] (OIHW f32, optionally accumulated) = sum_{n,oy,ox} dz[n,oy,ox,co] * x[n,oy*s+kh-p,ox*s+kw-p,ci]
 * on exact-f32 MFMA; k in {1, 3}.  Workgroups store partial blocks into the caller's workspace and a second kernel sums
 * them in a fixed order (deterministic, and no contended atomics). */
size_t upa_conv2d_wgrad_workspace_bytes(int cin, int cout, int k);
int upa_conv2d_wgrad(const void* x, int n, int h, int w, int cin, int ldx, const void* dz, int cout, int lddz, float* dw_oihw,
                     int k, int stride, int pad, int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Data gradient of a 3x3 stride-2 pad-1 conv in one launch: dx (n,h,w,cin) (+)= the four 2x2 phase correlations over dz (n,oh,ow,cout),
 * phase_w_packed = upa_dgrad_s2_phase_weights packed as a (cout -> 4 cin, k 2) conv; the convolution's epilogue writes dx's interleaved
 * pixels itself.  UPA_EUNSUPPORTED (nothing launched) outside the fused form: run the phase conv + upa_interleave2x. */
int upa_conv2d_dgrad_s2(const void* dz, int n, int oh, int ow, int cout, int lddz, const void* phase_w_packed, void* dx, int h, int w,
                        int cin, int lddx, int accumulate, int dtype, const upa_opts* opts, void* stream);
/* dst (n,h,w,c) = zero-inserted src (n,oh,ow,c): dst[y,x] = src[y/2,x/2] for even y, x (data gradient of stride 2). */
int upa_dilate2x(const void* src, int n, int oh, int ow, int c, int lds, void* dst, int h, int w, int ldd, int dtype,
                 void* stream);
/* Data gradient of a 3x3 stride-2 pad-1 conv by output parity: four 2x2 stride-1 correlations over dz (1/2/2/4 live taps)
 * instead of a 9-tap one over the zero-inserted dz.  upa_dgrad_s2_phase_weights writes V[phase = 2*py+px][ci][co][2][2] (f32,
 * "OIHW" with O = cin); each V[phase] is packed with upa_pack_conv_weight_dev(cout' = cin, cin' = cout, k = 2) and run as
 * upa_conv2d_bias_act(dz, k 2, stride 1, pad 1) into a phase map of (oh+1, ow+1) pixels; upa_interleave2x scatters
 * dx[2i+py][2j+px] (+)= phase[py][px][i+1][j+1]. */
int upa_dgrad_s2_phase_weights(const float* w_oihw, int cout, int cin, float* v, void* stream);
int upa_interleave2x(const void* t00, const void* t01, const void* t10, const void* t11, int n, int oh1, int ow1, int c, int ldt,
                     void* dx, int h, int w, int lddx, int accumulate, int dtype, void* stream);
/* dx (n,h,w,c) (+)= 2x2 block sums of dy (n,2h,2w,c): backward of nn.Upsample(scale 2, nearest). */
int upa_upsample2x_bwd(const void* dy, int n, int h, int w, int c, int lddy, void* dx, int lddx, int accumulate, int dtype,
                       void* stream);
/* Backward of nn.MaxPool2d(k, stride, pad): dy goes to the first maximum of each window (torch's index rule). */
size_t upa_maxpool2d_bwd_workspace_bytes(int n, int h, int w, int c, int k, int stride, int pad);
int upa_maxpool2d_bwd(const void* x, const void* dy, int n, int h, int w, int c, int ldx, int lddy, int k, int stride, int pad,
                      void* dx, int lddx, int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* *out (+)= sum g[i]^2 (f64): squared gradient norm for clip_grad_norm_ (trainer.py:676).  Fixed summation order (block
 * partials in `workspace`, upa_sumsq_workspace_bytes() bytes, then one folding workgroup): bit-reproducible. */
size_t upa_sumsq_workspace_bytes(void);
int upa_sumsq(const float* g, long n, double* out, int accumulate, void* workspace, void* stream);
/* clip_grad_norm_(max_norm) + SGD(nesterov, weight decay) + ModelEMA update over a flat parameter segment
 * (engine/trainer.py:674-682, :891-950; utils/torch_utils.py:632-646).  ema may be NULL.  ema_d_dev (nullable): device
 * float that overrides ema_d - lets a captured hipGraph of the step read the per-step EMA decay. */
int upa_sgd_nesterov_ema(float* p, float* g, float* momentum_buf, float* ema, long n, const double* grad_sumsq, float max_norm,
                         float lr, float momentum, float weight_decay, int first_step, float ema_d, const float* ema_d_dev,
                         int zero_grad, void* stream);
/* The same under an AMP GradScaler (engine/trainer.py:301-302, 429, 676-679; torch/amp/grad_scaler.py), kept entirely on the device:
 * scaler_state = 4 floats {loss scale, growth tracker, found_inf of the last update, 0}.  g holds gradients of loss * scale (see
 * upa_detection_loss_scaled), grad_sumsq the squared norm of those scaled gradients.  unscale_ + clip_grad_norm_ + scaler.step: the
 * norm and the update use g / scale; if grad_sumsq is inf / NaN the step leaves p and the momentum buffer untouched (gradients are
 * still zeroed and the EMA still updates, as the reference's optimizer_step does).  upa_grad_scaler_update = scaler.update(): an
 * overflowing step multiplies the scale by backoff_factor and clears the tracker, growth_interval clean steps in a row multiply it by
 * growth_factor.  No host synchronisation anywhere (the reference's scaler.step() reads found_inf on the host). */
int upa_sgd_nesterov_ema_scaled(float* p, float* g, float* momentum_buf, float* ema, long n, const double* grad_sumsq, float max_norm,
                                float lr, float momentum, float weight_decay, int first_step, float ema_d, const float* ema_d_dev,
                                int zero_grad, const float* scaler_state, void* stream);
int upa_grad_scaler_update(float* scaler_state, const double* grad_sumsq, float growth_factor, float backoff_factor,
                           int growth_interval, void* stream);
int upa_ema_update(float* ema, const float* v, long n, float d, const float* d_dev, void* stream);
/* dst view = src view converted between f32 and bf16 (head maps enter the loss as f32; c, strides multiples of 8). */
int upa_cast_view(const void* src, int src_dtype, int lds, void* dst, int dst_dtype, int ldd, long npix, int c, void* stream);
/* v8DetectionLoss forward + gradient wrt the raw head maps.  feats[l] / grads[l]: NHWC f32 rows [(b,y,x)][4*reg_max+nc]
 * with row stride lds[l]; gt: (b, max_gt, 5) rows (cls, x1, y1, x2, y2) in pixels, n_gt[b] valid rows; max_gt = the row
 * capacity per image, any value in [1, 1024] (the reference pads to counts.max(), utils/loss.py:445-461).
 * loss_items = (box, cls, dfl) as the reference reports them; gradients are those of loss.sum() * grad_scale
 * (grad_scale = world_size, engine/trainer.py:424-425). */
size_t upa_detection_loss_workspace_bytes(int b, int n_anchors, int max_gt);
int upa_detection_loss(const float* const* feats, float* const* grads, const int* hs, const int* ws, const int* lds,
                       const float* strides, int n_levels, int b, int nc, int reg_max, const float* gt, const int* n_gt,
                       int max_gt, float gain_box, float gain_cls, float gain_dfl, float grad_scale, float* loss_items,
                       void* workspace, size_t workspace_bytes, void* stream);
/* ... with the gradients multiplied by *grad_scale_dev (device memory; NULL = 1) as well: scaler.scale(loss).backward(),
 * engine/trainer.py:429.  loss_items stay unscaled. */
int upa_detection_loss_scaled(const float* const* feats, float* const* grads, const int* hs, const int* ws, const int* lds,
                              const float* strides, int n_levels, int b, int nc, int reg_max, const float* gt, const int* n_gt,
                              int max_gt, float gain_box, float gain_cls, float gain_dfl, float grad_scale,
                              const float* grad_scale_dev, float* loss_items, void* workspace, size_t workspace_bytes, void* stream);

/* ---- HIP graph helpers (capture a launch sequence once, replay per batch) -------------------------------------
 * upa_graph_begin / _end bracket a stream capture of upa_* launches; upa_graph_launch replays the instantiated graph.
 * Two rules for graphs that run concurrently with other graphs (several steps in flight on separate streams), both found
 * as intermittent "Memory access fault by GPU" aborts on ROCm 7.2 ("Two rules for graphs in flight"):
 *  1. NO MEMSET NODES: nothing inside a captured region may call hipMemsetAsync - with several graphs in flight a memset
 *     node now and then left its target stale.  Every upa_* entry that needs zeroed counters zeroes them with a kernel
 *     (upa_zero_words, csrc/common.h); callers must do the same in their own captured code.
 *  2. NEVER REPLAY ON THE NULL STREAM: upa_graph_launch(exec, 0) is legal but replaying graphs with parallel branches on the
 *     null stream and other graphs on created streams raised the fault rate from rare to most processes.  The Python
 *     wrapper (engine/runtime.HipGraph.replay) therefore REDIRECTS a launch requested on the null stream to the graph's own
 *     capture stream, ordered before and after by stream waits - callers of this C API should pass a created stream. */
int upa_graph_begin(void* stream);
int upa_graph_end(void* stream, void** graph_exec_out);
int upa_graph_launch(void* graph_exec, void* stream);
int upa_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif
