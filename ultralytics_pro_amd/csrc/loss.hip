// YOLOv8 detection training loss on the GPU: TaskAlignedAssigner + SlideLoss BCE + CIoU + DFL, forward AND the gradient
// with respect to the raw head maps, replacing v8DetectionLoss.__call__ + autograd (ultralytics/utils/loss.py:471-528,
// :21-46, :308-360; utils/tal.py:12-316; utils/metrics.py:77-150).
//
// Tensors: head maps of level l are NHWC f32 rows [(b, y, x)][4*reg_max + nc] (row stride ld); gradients are written in
// the same layout.  Ground truth is padded per image: gt[b][g] = (cls, x1, y1, x2, y2) in pixels, n_gt[b] valid rows.
// Pipeline (all sizes tiny next to the convolutions - B*A = 32*8400 anchors, <= 32 boxes per image):
//   1. loss_decode_kernel      per anchor: DFL expectation -> box (grid units), kept for the later stages
//   2. tal_topk_kernel         per (image, gt): alignment metric of every anchor inside the box, 10 best anchors
//   3. tal_resolve_kernel      per image: anchors claimed by several gts go to the highest overlap, per-gt maxima,
//                              target score of every positive anchor, target-score sum
//   4. loss_grad_kernel        per anchor: the three loss terms and d(loss)/d(head map)
// Ties of the top-k (equal metrics) resolve to the lower anchor index; torch.topk leaves that order unspecified, and
// the only tie that occurs in practice (metric 0 outside every box) is masked out afterwards (tal.py:141).
#include "common.h"

namespace {

constexpr int REG = 16;
constexpr int TOPK = 10;
constexpr int MAXG_CAP = 1024;  // largest gt-row capacity per image (LDS arrays of the resolve kernel); the capacity of a
                                // call is the runtime `L.maxg` = rows per image of the padded gt tensor

struct LossLevels {
  const float* feat[3];
  float* grad[3];
  int h[3], w[3], ld[3];
  float stride[3];
  int a0[3];   // first anchor index of the level
  int nl, A, B, nc;
  int maxg;    // gt rows per image in `gt` / `cand` (the reference pads to counts.max(), loss.py:445-461)
};

__device__ __forceinline__ void anchor_decode(const LossLevels& L, int a, int& lvl, int& ay, int& ax) {
  lvl = 0;
  if (L.nl > 1 && a >= L.a0[1]) lvl = 1;
  if (L.nl > 2 && a >= L.a0[2]) lvl = 2;
  const int r = a - L.a0[lvl];
  ay = r / L.w[lvl];
  ax = r - ay * L.w[lvl];
}

// CIoU of (b1 = first argument, b2 = second), xyxy, exactly the operation order of metrics.py:118-142
__device__ __forceinline__ float ciou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  const float eps = 1e-7f;
  const float w1 = ax2 - ax1, h1 = ay2 - ay1 + eps, w2 = bx2 - bx1, h2 = by2 - by1 + eps;
  const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = iw * ih;
  const float uni = w1 * h1 + w2 * h2 - inter + eps;
  const float iou = inter / uni;
  const float cw = fmaxf(ax2, bx2) - fminf(ax1, bx1), ch = fmaxf(ay2, by2) - fminf(ay1, by1);
  const float c2 = cw * cw + ch * ch + eps;
  const float dx = bx1 + bx2 - ax1 - ax2, dy = by1 + by2 - ay1 - ay2;
  const float rho2 = (dx * dx + dy * dy) / 4.f;
  const float dat = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = 0.40528473456935109f * (dat * dat);  // 4 / pi^2
  const float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

// ---- 1. decode: pred box (grid units, xyxy) per anchor ------------------------------------------------------------
__global__ void loss_decode_kernel(const LossLevels L, float* pbox /* [B][A][4] */) {
  const long total = (long)L.B * L.A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    int lvl, ay, ax;
    anchor_decode(L, a, lvl, ay, ax);
    const float* row = L.feat[lvl] + ((size_t)(b * L.h[lvl] + ay) * L.w[lvl] + ax) * L.ld[lvl];
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float m = row[s * REG];
      for (int k = 1; k < REG; ++k) m = fmaxf(m, row[s * REG + k]);
      float sum = 0.f, e = 0.f;
      for (int k = 0; k < REG; ++k) {
        const float ex = expf(row[s * REG + k] - m);
        sum += ex;
        e += ex * (float)k;
      }
      d[s] = e / sum;
    }
    const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f;
    float* o = pbox + i * 4;
    o[0] = cx - d[0]; o[1] = cy - d[1]; o[2] = cx + d[2]; o[3] = cy + d[3];
  }
}

// ---- 2. per (image, gt): top-k anchors by alignment metric ---------------------------------------------------------
// metric = sigmoid(cls logit of the gt class)^0.5 * CIoU(gt, pred * stride).clamp(0)^6 for anchors whose centre lies
// strictly inside the gt box (tal.py:146-178, 271-291); 0 elsewhere.
__global__ __launch_bounds__(256) void tal_topk_kernel(const LossLevels L, const float* pbox, const float* gt /* [B][maxg][5] */,
                                                        const int* n_gt, int* cand /* [B][maxg][TOPK] anchor or -1 */) {
  extern __shared__ float s_metric[];  // [A]: the alignment metric of every anchor for this gt (computed once)
  const int b = blockIdx.x / L.maxg, g = blockIdx.x % L.maxg;
  int* out = cand + ((size_t)b * L.maxg + g) * TOPK;
  if (g >= n_gt[b]) {
    if (threadIdx.x < TOPK) out[threadIdx.x] = -1;
    return;
  }
  const float* gb = gt + ((size_t)b * L.maxg + g) * 5;
  const int cls = (int)gb[0];
  const float gx1 = gb[1], gy1 = gb[2], gx2 = gb[3], gy2 = gb[4];
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  const int tid = threadIdx.x;
  for (int a = tid; a < L.A; a += 256) {
    int lvl, ay, ax;
    anchor_decode(L, a, lvl, ay, ax);
    const float st = L.stride[lvl];
    const float px = ((float)ax + 0.5f) * st, py = ((float)ay + 0.5f) * st;
    const float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
    float metric = 0.f;
    if (dmin > 1e-9f) {
      const float* pb = pbox + ((size_t)b * L.A + a) * 4;
      float ov = ciou(gx1, gy1, gx2, gy2, pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st);
      ov = fmaxf(ov, 0.f);
      const float logit = L.feat[lvl][((size_t)(b * L.h[lvl] + ay) * L.w[lvl] + ax) * L.ld[lvl] + 4 * REG + cls];
      const float sc = 1.0f / (1.0f + expf(-logit));
      const float o2 = ov * ov;
      metric = sqrtf(sc) * (o2 * o2 * o2);
    }
    s_metric[a] = metric;
  }
  __syncthreads();
  for (int k = 0; k < TOPK; ++k) {
    // best (metric desc, anchor asc) among the remaining anchors; taken ones are marked -1
    float best = -1.f;
    int besti = 0x7fffffff;
    for (int a = tid; a < L.A; a += 256) {
      const float m = s_metric[a];
      if (m > best) { best = m; besti = a; }  // ascending a: the first maximum wins
    }
    s_val[tid] = best; s_idx[tid] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const float v2 = s_val[tid + o];
        const int i2 = s_idx[tid + o];
        if (v2 > s_val[tid] || (v2 == s_val[tid] && i2 < s_idx[tid])) { s_val[tid] = v2; s_idx[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      out[k] = s_idx[0];
      s_metric[s_idx[0]] = -1.f;
    }
    __syncthreads();
  }
}

// ---- 3. per image: resolve assignments ---------------------------------------------------------------------------
// fg anchors = top-k candidates that lie inside their gt (tal.py:141); an anchor claimed by more than one gt is given to
// the gt with the largest overlap among ALL gts (argmax over the full overlaps row, tal.py:305-311; overlaps are zero
// outside (in-box & valid)).  Then per gt: max alignment / max overlap over its anchors; per positive anchor:
// target score = align * pos_overlap / (pos_align + eps)   (tal.py:118-124).
struct Assign {  // one per anchor
  int gt;        // assigned gt or -1
  float score;   // target score (normalised alignment metric)
};

__device__ float anchor_gt_metric(const LossLevels& L, const float* pbox, const float* gb, int b, int a, float* overlap) {
  int lvl, ay, ax;
  anchor_decode(L, a, lvl, ay, ax);
  const float st = L.stride[lvl];
  const float px = ((float)ax + 0.5f) * st, py = ((float)ay + 0.5f) * st;
  const float dmin = fminf(fminf(px - gb[1], py - gb[2]), fminf(gb[3] - px, gb[4] - py));
  *overlap = 0.f;
  if (!(dmin > 1e-9f)) return 0.f;
  const float* pb = pbox + ((size_t)b * L.A + a) * 4;
  const float ov = fmaxf(ciou(gb[1], gb[2], gb[3], gb[4], pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st), 0.f);
  *overlap = ov;
  const float logit = L.feat[lvl][((size_t)(b * L.h[lvl] + ay) * L.w[lvl] + ax) * L.ld[lvl] + 4 * REG + (int)gb[0]];
  const float sc = 1.0f / (1.0f + expf(-logit));
  const float o2 = ov * ov;
  return sqrtf(sc) * (o2 * o2 * o2);
}

__global__ __launch_bounds__(256) void tal_resolve_kernel(const LossLevels L, const float* pbox, const float* gt, const int* n_gt,
                                                           const int* cand, Assign* asg /* [B][A] */, int* count /* [B][A] scratch */,
                                                           double* tss /* [1] */, int* n_fg /* [1] */) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ng = n_gt[b];
  Assign* A_ = asg + (size_t)b * L.A;
  int* cnt = count + (size_t)b * L.A;
  for (int a = tid; a < L.A; a += 256) { A_[a].gt = -1; A_[a].score = 0.f; cnt[a] = 0; }
  __syncthreads();
  // count claims (inside-box candidates only); a gt lists an anchor at most once
  for (int i = tid; i < ng * TOPK; i += 256) {
    const int g = i / TOPK;
    const int a = cand[((size_t)b * L.maxg + g) * TOPK + (i - g * TOPK)];
    if (a < 0) continue;
    float ov;
    const float* gb = gt + ((size_t)b * L.maxg + g) * 5;
    anchor_gt_metric(L, pbox, gb, b, a, &ov);
    int lvl, ay, ax;
    anchor_decode(L, a, lvl, ay, ax);
    const float st = L.stride[lvl];
    const float px = ((float)ax + 0.5f) * st, py = ((float)ay + 0.5f) * st;
    const float dmin = fminf(fminf(px - gb[1], py - gb[2]), fminf(gb[3] - px, gb[4] - py));
    if (dmin > 1e-9f) {
      atomicAdd(&cnt[a], 1);
      atomicMax(&A_[a].gt, g);  // provisional (exact when the count stays 1)
    }
  }
  __syncthreads();
  // multiply claimed anchors: argmax of the overlaps over all gts (first maximum)
  for (int a = tid; a < L.A; a += 256) {
    if (cnt[a] > 1) {
      float best = -1.f;
      int bg = 0;
      for (int g = 0; g < ng; ++g) {
        float ov;
        anchor_gt_metric(L, pbox, gt + ((size_t)b * L.maxg + g) * 5, b, a, &ov);
        if (ov > best) { best = ov; bg = g; }
      }
      A_[a].gt = bg;
    }
  }
  __syncthreads();
  // per gt maxima over its anchors
  __shared__ float pos_align[MAXG_CAP], pos_ov[MAXG_CAP];
  __shared__ unsigned s_align[MAXG_CAP], s_ov[MAXG_CAP];  // non-negative floats compare like unsigned ints
  for (int g = tid; g < L.maxg; g += 256) { s_align[g] = 0u; s_ov[g] = 0u; }
  __syncthreads();
  for (int a = tid; a < L.A; a += 256) {
    const int g = A_[a].gt;
    if (g < 0) continue;
    float ov;
    const float al = anchor_gt_metric(L, pbox, gt + ((size_t)b * L.maxg + g) * 5, b, a, &ov);
    A_[a].score = al;  // alignment metric for now
    atomicMax(&s_align[g], __float_as_uint(al));
    atomicMax(&s_ov[g], __float_as_uint(ov));
  }
  __syncthreads();
  for (int g = tid; g < L.maxg; g += 256) { pos_align[g] = __uint_as_float(s_align[g]); pos_ov[g] = __uint_as_float(s_ov[g]); }
  __syncthreads();
  double local = 0.0;
  int nf = 0;
  for (int a = tid; a < L.A; a += 256) {
    const int g = A_[a].gt;
    if (g < 0) continue;
    const float sc = A_[a].score * pos_ov[g] / (pos_align[g] + 1e-9f);
    A_[a].score = sc;
    local += (double)sc;
    ++nf;
  }
  __shared__ double red[256];
  __shared__ int redn[256];
  red[tid] = local; redn[tid] = nf;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[tid] += red[tid + o]; redn[tid] += redn[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { atomicAdd(tss, red[0]); atomicAdd(n_fg, redn[0]); }
}

// ---- 4a. classification term, one thread per (anchor, class): SlideLoss(BCEWithLogits) (loss.py:21-46, auto_iou = 0.5)
__global__ __launch_bounds__(256) void loss_cls_kernel(const LossLevels L, const float* gt, const Assign* asg, const double* tss_p,
                                                        double* out, float gain_cls, float grad_scale, const float* gs_dev) {
  if (gs_dev) grad_scale *= *gs_dev;  // AMP loss scale, kept on the device (GradScaler.scale(loss).backward())
  const float tss = fmaxf((float)*tss_p, 1.f);
  const float gs = grad_scale * (float)L.B / tss;
  double l_cls = 0.0;
  const long total = (long)L.B * L.A * L.nc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long i = idx / L.nc;
    const int c = (int)(idx - i * L.nc);
    const int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    int lvl, ay, ax;
    anchor_decode(L, a, lvl, ay, ax);
    const size_t roff = ((size_t)(b * L.h[lvl] + ay) * L.w[lvl] + ax) * L.ld[lvl] + 4 * REG + c;
    const Assign as = asg[i];
    const int tcls = as.gt >= 0 ? (int)gt[((size_t)b * L.maxg + as.gt) * 5] : -1;
    const float x = L.feat[lvl][roff];
    const float t = c == tcls ? as.score : 0.f;
    float wgt;
    if (t <= 0.4f) wgt = 1.0f;
    else if (t < 0.5f) wgt = 1.6487212707001282f;  // exp(1 - 0.5)
    else wgt = expf(-(t - 1.0f));
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));  // max(x,0) - x*t + log(1 + exp(-|x|))
    l_cls += (double)(bce * wgt);
    const float sg = 1.0f / (1.0f + expf(-x));
    L.grad[lvl][roff] = gain_cls * gs * wgt * (sg - t);
  }
  __shared__ double red[256];
  red[threadIdx.x] = l_cls;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&out[1], red[0]);
}

// ... four classes of an anchor per thread (nc % 4 == 0, 16-byte aligned rows): one 16-byte load and store instead of four
// scalar ones, the anchor's assignment read once per four classes, 32-bit index arithmetic, one exponential per element
// (sigmoid and softplus both from e = exp(-|x|)), hardware transcendentals.  The element-per-thread form above took 144 us for yolov8s' 21.5 M
// scores (two 64-bit divisions and three exponentials per element) against a 21 us traffic floor.
__global__ __launch_bounds__(256) void loss_cls4_kernel(const LossLevels L, const float* gt, const Assign* asg, const double* tss_p,
                                                         double* out, float gain_cls, float grad_scale, const float* gs_dev) {
  if (gs_dev) grad_scale *= *gs_dev;
  const float tss = fmaxf((float)*tss_p, 1.f);
  const float gs = gain_cls * grad_scale * (float)L.B / tss;
  double l_cls = 0.0;
  const unsigned nc4 = (unsigned)L.nc >> 2;
  const unsigned total = (unsigned)L.B * (unsigned)L.A * nc4;
  // U independent elements per trip: all their loads (scores, assignment, then the assigned class) are issued before the
  // arithmetic - with one 16-byte load in flight per thread the kernel moved 2.3 TB/s
  constexpr int U = 4;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned idx0 = blockIdx.x * blockDim.x + threadIdx.x; idx0 < total; idx0 += U * stride) {
    size_t roff[U]; int c0[U], lvl[U]; unsigned bb[U]; bool live[U];
    f32x4 x4[U]; Assign as[U]; int tcls[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned idx = idx0 + u * stride;
      live[u] = idx < total;
      const unsigned i = live[u] ? idx / nc4 : 0;
      c0[u] = live[u] ? (int)(idx - i * nc4) * 4 : 0;
      bb[u] = i / (unsigned)L.A;
      const int a = (int)(i - bb[u] * (unsigned)L.A);
      lvl[u] = 0;
      if (L.nl > 1 && a >= L.a0[1]) lvl[u] = 1;
      if (L.nl > 2 && a >= L.a0[2]) lvl[u] = 2;
      roff[u] = ((size_t)bb[u] * (L.h[lvl[u]] * L.w[lvl[u]]) + (a - L.a0[lvl[u]])) * L.ld[lvl[u]] + 4 * REG + c0[u];
      as[u] = asg[i];
      x4[u] = *reinterpret_cast<const f32x4*>(L.feat[lvl[u]] + roff[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) tcls[u] = as[u].gt >= 0 ? (int)gt[((size_t)bb[u] * L.maxg + as[u].gt) * 5] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      f32x4 g4;
      float lsum = 0.f;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float x = x4[u][e4];
        const float t = c0[u] + e4 == tcls[u] ? as[u].score : 0.f;
        float wgt;
        if (t <= 0.4f) wgt = 1.0f;
        else if (t < 0.5f) wgt = 1.6487212707001282f;  // exp(1 - 0.5)
        else wgt = __expf(-(t - 1.0f));
        // hardware exp2 / log2 / rcp (1-2 ulp): ex is in (0, 1], so log(1 + ex) is off by at most 6e-8 absolute where 1 + ex rounds
        const float ex = __expf(-fabsf(x));
        const float bce = fmaxf(x, 0.f) - x * t + __logf(1.0f + ex);   // max(x,0) - x*t + log(1 + exp(-|x|))
        lsum += bce * wgt;
        const float sg = __fdividef(x >= 0.f ? 1.0f : ex, 1.0f + ex);  // sigmoid(x)
        g4[e4] = gs * wgt * (sg - t);
      }
      l_cls += (double)lsum;
      *reinterpret_cast<f32x4*>(L.grad[lvl[u]] + roff[u]) = g4;
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = l_cls;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&out[1], red[0]);
}

// ---- 4. losses and gradients ---------------------------------------------------------------------------------------
// out[0..2] += box, cls, dfl sums (before the gains and the / target_scores_sum); gradients carry
// gain * batch_size * grad_scale / max(tss, 1).
__global__ __launch_bounds__(256) void loss_grad_kernel(const LossLevels L, const float* pbox, const float* gt, const Assign* asg,
                                                         const double* tss_p, double* out, float gain_box, float gain_cls,
                                                         float gain_dfl, float grad_scale, const float* gs_dev) {
  if (gs_dev) grad_scale *= *gs_dev;
  const float tss = fmaxf((float)*tss_p, 1.f);
  double l_box = 0.0, l_cls = 0.0, l_dfl = 0.0;
  const long total = (long)L.B * L.A;
  const float gs = grad_scale * (float)L.B / tss;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    int lvl, ay, ax;
    anchor_decode(L, a, lvl, ay, ax);
    const size_t roff = ((size_t)(b * L.h[lvl] + ay) * L.w[lvl] + ax) * L.ld[lvl];
    const float* row = L.feat[lvl] + roff;
    float* grow = L.grad[lvl] + roff;
    const Assign as = asg[i];
    // ---- box + DFL for positive anchors
    if (as.gt < 0) {
      for (int k = 0; k < 4 * REG; ++k) grow[k] = 0.f;
      continue;
    }
    const float st = L.stride[lvl];
    const float* gb = gt + ((size_t)b * L.maxg + as.gt) * 5;
    const float tx1 = gb[1] / st, ty1 = gb[2] / st, tx2 = gb[3] / st, ty2 = gb[4] / st;
    const float wgt = as.score;  // target_scores.sum(-1): one class per anchor
    const float* pb = pbox + i * 4;
    const float px1 = pb[0], py1 = pb[1], px2 = pb[2], py2 = pb[3];
    // CIoU(pred, target) and its gradient wrt the pred box (alpha constant, metrics.py:139-141)
    const float eps = 1e-7f;
    const float w1 = px2 - px1, h1 = py2 - py1 + eps, w2 = tx2 - tx1, h2 = ty2 - ty1 + eps;
    const float ix1 = fmaxf(px1, tx1), iy1 = fmaxf(py1, ty1), ix2 = fminf(px2, tx2), iy2 = fminf(py2, ty2);
    const float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float cx1 = fminf(px1, tx1), cy1 = fminf(py1, ty1), cx2 = fmaxf(px2, tx2), cy2 = fmaxf(py2, ty2);
    const float cw = cx2 - cx1, ch = cy2 - cy1;
    const float c2 = cw * cw + ch * ch + eps;
    const float ddx = tx1 + tx2 - px1 - px2, ddy = ty1 + ty2 - py1 - py2;
    const float rho2 = (ddx * ddx + ddy * ddy) / 4.f;
    const float at1 = atanf(w1 / h1), at2 = atanf(w2 / h2);
    const float dat = at2 - at1;
    const float v = 0.40528473456935109f * (dat * dat);
    const float alpha = v / (v - iou + (1.f + eps));
    const float ciou_v = iou - (rho2 / c2 + v * alpha);
    l_box += (double)((1.0f - ciou_v) * wgt);
    // d ciou / d(px1, py1, px2, py2)
    float dI[4], dU[4], dC2[4], dRho[4], dV[4];
    // inter = iw*ih: d iw/d px1 = -(px1 > tx1) if iw > 0 ...   (clamp(0) passes gradient only where positive; at ties
    // torch's maximum/minimum split the gradient evenly - measure-zero, taken as the pred side here)
    const float giw = iw > 0.f ? 1.f : 0.f, gih = ih > 0.f ? 1.f : 0.f;
    const float d_ix1 = px1 >= tx1 ? 1.f : 0.f, d_iy1 = py1 >= ty1 ? 1.f : 0.f;
    const float d_ix2 = px2 <= tx2 ? 1.f : 0.f, d_iy2 = py2 <= ty2 ? 1.f : 0.f;
    dI[0] = -giw * d_ix1 * ih; dI[1] = -gih * d_iy1 * iw; dI[2] = giw * d_ix2 * ih; dI[3] = gih * d_iy2 * iw;
    // union = w1*h1 + w2*h2 - inter + eps
    dU[0] = -h1 - dI[0]; dU[1] = -w1 - dI[1]; dU[2] = h1 - dI[2]; dU[3] = w1 - dI[3];
    // c2 = cw^2 + ch^2
    const float d_cx1 = px1 <= tx1 ? 1.f : 0.f, d_cy1 = py1 <= ty1 ? 1.f : 0.f;
    const float d_cx2 = px2 >= tx2 ? 1.f : 0.f, d_cy2 = py2 >= ty2 ? 1.f : 0.f;
    dC2[0] = -2.f * cw * d_cx1; dC2[1] = -2.f * ch * d_cy1; dC2[2] = 2.f * cw * d_cx2; dC2[3] = 2.f * ch * d_cy2;
    dRho[0] = -ddx / 2.f; dRho[1] = -ddy / 2.f; dRho[2] = -ddx / 2.f; dRho[3] = -ddy / 2.f;
    // v = k*(at2-at1)^2, at1 = atan(w1/h1): d at1/d w1 = h1/(w1^2+h1^2), d at1/d h1 = -w1/(w1^2+h1^2)
    const float den = w1 * w1 + h1 * h1;
    const float dv_dat1 = -2.f * 0.40528473456935109f * dat;
    const float dat1_dw = h1 / den, dat1_dh = -w1 / den;
    dV[0] = dv_dat1 * (-dat1_dw); dV[1] = dv_dat1 * (-dat1_dh); dV[2] = dv_dat1 * dat1_dw; dV[3] = dv_dat1 * dat1_dh;
    float dbox[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d_iou = (dI[q] * uni - inter * dU[q]) / (uni * uni);
      const float d_pen = (dRho[q] * c2 - rho2 * dC2[q]) / (c2 * c2);
      const float d_ciou = d_iou - (d_pen + alpha * dV[q]);
      dbox[q] = gain_box * gs * wgt * (-d_ciou);  // d loss_box / d pred box coordinate
    }
    // pred box = (cx - l, cy - t, cx + r, cy + b): d/d(l,t,r,b) = (-dx1, -dy1, +dx2, +dy2)
    const float ddist[4] = {-dbox[0], -dbox[1], dbox[2], dbox[3]};
    // target ltrb for DFL (tal.py:379-382, loss.py:314-325)
    const float acx = (float)ax + 0.5f, acy = (float)ay + 0.5f;
    const float tl4[4] = {acx - tx1, acy - ty1, tx2 - acx, ty2 - acy};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* lg = row + s * REG;
      float m = lg[0];
      for (int k = 1; k < REG; ++k) m = fmaxf(m, lg[k]);
      float sum = 0.f, e = 0.f;
      float pr[REG];
      for (int k = 0; k < REG; ++k) { pr[k] = expf(lg[k] - m); sum += pr[k]; e += pr[k] * (float)k; }
      const float dist = e / sum;
      const float lse = m + logf(sum);
      float tgt = fminf(fmaxf(tl4[s], 0.f), (float)(REG - 1) - 0.01f);
      const int tl = (int)tgt;
      const float wl = (float)(tl + 1) - tgt, wr = 1.f - wl;
      // CE(tl)*wl + CE(tr)*wr, mean over the 4 sides, * weight
      l_dfl += (double)(((lse - lg[tl]) * wl + (lse - lg[tl + 1]) * wr) * 0.25f * wgt);
      const float gd = gain_dfl * gs * wgt * 0.25f;
      for (int k = 0; k < REG; ++k) {
        const float p_ = pr[k] / sum;
        // DFL expectation gradient: d dist / d logit_k = p_k (k - dist)
        float gk = ddist[s] * p_ * ((float)k - dist);
        gk += gd * ((wl + wr) * p_ - (k == tl ? wl : 0.f) - (k == tl + 1 ? wr : 0.f));
        grow[s * REG + k] = gk;
      }
    }
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = l_box; red[1][threadIdx.x] = l_cls; red[2][threadIdx.x] = l_dfl;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicAdd(&out[threadIdx.x], red[threadIdx.x][0]);
}

__global__ void loss_finish_kernel(const double* sums, const double* tss, float gain_box, float gain_cls, float gain_dfl,
                                   float* items /* [3] box, cls, dfl (the reference's loss_items) */) {
  const float t = fmaxf((float)*tss, 1.f);
  items[0] = (float)sums[0] / t * gain_box;
  items[1] = (float)sums[1] / t * gain_cls;
  items[2] = (float)sums[2] / t * gain_dfl;
}

}  // namespace

extern "C" size_t upa_detection_loss_workspace_bytes(int b, int a, int max_gt) {
  // pbox (B*A*4 f32) + cand (B*max_gt*TOPK i32) + assign (B*A*8) + count (B*A i32) + scalars (tss, sums[3] f64, n_fg)
  return (size_t)b * a * 16 + (size_t)b * max_gt * TOPK * 4 + (size_t)b * a * 8 + (size_t)b * a * 4 + 64;
}

extern "C" int upa_detection_loss_scaled(const float* const* feats, float* const* grads, const int* hs, const int* ws, const int* lds_,
                                         const float* strides, int n_levels, int b, int nc, int reg_max, const float* gt, const int* n_gt,
                                         int max_gt, float gain_box, float gain_cls, float gain_dfl, float grad_scale,
                                         const float* grad_scale_dev, float* loss_items, void* workspace, size_t workspace_bytes,
                                         void* stream);
extern "C" int upa_detection_loss(const float* const* feats, float* const* grads, const int* hs, const int* ws, const int* lds_,
                                  const float* strides, int n_levels, int b, int nc, int reg_max, const float* gt, const int* n_gt,
                                  int max_gt, float gain_box, float gain_cls, float gain_dfl, float grad_scale, float* loss_items,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return upa_detection_loss_scaled(feats, grads, hs, ws, lds_, strides, n_levels, b, nc, reg_max, gt, n_gt, max_gt, gain_box, gain_cls,
                                   gain_dfl, grad_scale, nullptr, loss_items, workspace, workspace_bytes, stream);
}

// ... with the gradients multiplied by a scale read from DEVICE memory as well (the AMP GradScaler's loss scale, engine/trainer.py:429:
// the scale changes on the device when a step overflows - the host never reads it); the loss items are unscaled.
extern "C" int upa_detection_loss_scaled(const float* const* feats, float* const* grads, const int* hs, const int* ws, const int* lds_,
                                         const float* strides, int n_levels, int b, int nc, int reg_max, const float* gt, const int* n_gt,
                                         int max_gt, float gain_box, float gain_cls, float gain_dfl, float grad_scale,
                                         const float* grad_scale_dev, float* loss_items, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  UPA_CHECK_ARG(feats && grads && hs && ws && lds_ && strides && gt && n_gt && loss_items && workspace, "detection_loss: null pointer");
  UPA_CHECK_ARG(n_levels >= 1 && n_levels <= 3 && reg_max == REG && nc >= 1, "detection_loss: unsupported shape (levels <= 3, "
                "reg_max 16)");
  UPA_CHECK_ARG(max_gt >= 1 && max_gt <= MAXG_CAP, "detection_loss: max_gt (gt rows per image) must be in [1, %d]", MAXG_CAP);
  LossLevels L{};
  L.nl = n_levels; L.B = b; L.nc = nc; L.maxg = max_gt;
  int a = 0;
  for (int l = 0; l < n_levels; ++l) {
    L.feat[l] = feats[l]; L.grad[l] = grads[l]; L.h[l] = hs[l]; L.w[l] = ws[l]; L.ld[l] = lds_[l]; L.stride[l] = strides[l];
    L.a0[l] = a;
    a += hs[l] * ws[l];
  }
  L.A = a;
  UPA_CHECK_ARG(workspace_bytes >= upa_detection_loss_workspace_bytes(b, a, max_gt), "detection_loss: workspace too small");
  char* wsb = (char*)workspace;
  double* scal = (double*)wsb;                // [0] tss, [1..3] sums, then n_fg
  int* n_fg = (int*)(scal + 4);
  float* pbox = (float*)(wsb + 64);
  int* cand = (int*)(pbox + (size_t)b * a * 4);
  Assign* asg = (Assign*)(cand + (size_t)b * max_gt * TOPK);
  int* count = (int*)(asg + (size_t)b * a);
  hipStream_t s = (hipStream_t)stream;
  upa_zero_words(wsb, 16, s);  // not hipMemsetAsync: see upa_zero_words (common.h)
  const long total = (long)b * a;
  const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(loss_decode_kernel, dim3(grid), dim3(256), 0, s, L, pbox);
  {
    // the per-gt metric row of all anchors lives in dynamic LDS: raise the kernel's limit to the whole CU (the 64 KB default
    // ends at imgsz ~ 900) and refuse shapes past it (A * 4 bytes + 2 KB static <= 160 KB: imgsz up to ~1400 square)
    if (hipError_t e = upa_full_lds<tal_topk_kernel>(); e != hipSuccess) {
      upa_set_error("detection_loss: cannot raise the LDS limit of tal_topk: %s", hipGetErrorString(e));
      return UPA_ELAUNCH;
    }
    UPA_CHECK_ARG((size_t)a * sizeof(float) + 2048 <= 160 * 1024, "detection_loss: %d anchors do not fit the LDS metric row", a);
  }
  hipLaunchKernelGGL(tal_topk_kernel, dim3(b * max_gt), dim3(256), (size_t)a * sizeof(float), s, L, pbox, gt, n_gt, cand);
  hipLaunchKernelGGL(tal_resolve_kernel, dim3(b), dim3(256), 0, s, L, pbox, gt, n_gt, cand, asg, count, scal, n_fg);
  const long tot_c = total * nc;
  bool vec4 = nc % 4 == 0 && tot_c / 4 < (1L << 31);
  for (int l = 0; l < L.nl; ++l)
    vec4 = vec4 && L.ld[l] % 4 == 0 && ((uintptr_t)L.feat[l] & 15) == 0 && ((uintptr_t)L.grad[l] & 15) == 0;
  if (vec4) {
    const long t4 = tot_c / 4;
    hipLaunchKernelGGL(loss_cls4_kernel, dim3((int)((t4 + 255) / 256 > 1024 ? 1024 : (t4 + 255) / 256)), dim3(256), 0, s, L, gt, asg,
                       scal, scal + 1, gain_cls, grad_scale, grad_scale_dev);
  } else {
    hipLaunchKernelGGL(loss_cls_kernel, dim3((int)((tot_c + 255) / 256 > 8192 ? 8192 : (tot_c + 255) / 256)), dim3(256), 0, s, L, gt, asg,
                       scal, scal + 1, gain_cls, grad_scale, grad_scale_dev);
  }
  hipLaunchKernelGGL(loss_grad_kernel, dim3(grid), dim3(256), 0, s, L, pbox, gt, asg, scal, scal + 1, gain_box, gain_cls, gain_dfl,
                     grad_scale, grad_scale_dev);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, s, scal + 1, scal, gain_box, gain_cls, gain_dfl, loss_items);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
