// Pairwise IoU matrix for validation: box_iou(box1 (N,4), box2 (M,4)) -> (N,M), xyxy, eps in the denominator.
// Replaces ultralytics/utils/metrics.py:54-74 as used by DetectionValidator._process_batch (models/yolo/detect/val.py:286).
// One lane per (i, j) pair, j fastest (coalesced row writes); f32 arithmetic in the reference's operation order.
#include "common.h"
#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void box_iou_kernel(const float* b1, const float* b2, float* out, int N, int M, float eps) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * M) return;
  const int i = (int)(idx / M), j = (int)(idx - (long)i * M);
  const float ax1 = b1[i * 4], ay1 = b1[i * 4 + 1], ax2 = b1[i * 4 + 2], ay2 = b1[i * 4 + 3];
  const float bx1 = b2[j * 4], by1 = b2[j * 4 + 1], bx2 = b2[j * 4 + 2], by2 = b2[j * 4 + 3];
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = w * h;
  out[idx] = inter / ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter + eps);
}

extern "C" int upa_box_iou(const float* box1, int n, const float* box2, int m, float eps, float* out, void* stream) {
  UPA_CHECK_ARG(out && (n == 0 || box1) && (m == 0 || box2), "box_iou: null pointer");
  if (n == 0 || m == 0) return UPA_OK;
  const long total = (long)n * m;
  hipLaunchKernelGGL(box_iou_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, box1, box2, out,
                     n, m, eps);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// scale_boxes + clip_boxes (utils/ops.py:102-178), in place on the first 4 floats of each row, same op order.
__global__ __launch_bounds__(256) void scale_boxes_kernel(float* rows, long n, int rs, float gain, float px, float py,
                                                          int padding, float w0, float h0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float* b = rows + i * rs;
  float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
  if (padding) { x1 -= px; y1 -= py; x2 -= px; y2 -= py; }
  x1 /= gain; y1 /= gain; x2 /= gain; y2 /= gain;
  b[0] = fminf(fmaxf(x1, 0.f), w0);
  b[1] = fminf(fmaxf(y1, 0.f), h0);
  b[2] = fminf(fmaxf(x2, 0.f), w0);
  b[3] = fminf(fmaxf(y2, 0.f), h0);
}

extern "C" int upa_scale_boxes(float* rows, long n, int row_stride, float gain, float pad_x, float pad_y, int padding,
                               float w0, float h0, void* stream) {
  UPA_CHECK_ARG(n == 0 || rows, "scale_boxes: null pointer");
  UPA_CHECK_ARG(row_stride >= 4 && gain > 0.f, "scale_boxes: bad row stride / gain");
  if (n == 0) return UPA_OK;
  hipLaunchKernelGGL(scale_boxes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n,
                     row_stride, gain, pad_x, pad_y, padding, w0, h0);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// match_predictions for a whole batch (engine/validator.py:267-308, the non-scipy branch, through
// DetectionValidator._process_batch, models/yolo/detect/val.py:274-288): the true-positive matrix (n_det, 10 IoU
// thresholds) of every image, computed on the fixed-shape NMS outputs without a host round trip.
// The reference, per threshold: candidate pairs (label, detection) with IoU >= thr and equal class, sorted by IoU
// descending; per detection the first pair (its best label); the survivors re-ordered by detection index
// (np.unique(..., return_index=True)); per label the first of those, i.e. the SMALLEST detection index among the detections
// whose best label it is.  A detection's best label does not depend on the threshold, so one pass finds it; the
// per-(label, threshold) minimum detection index is an LDS atomicMin.  IoU as utils/metrics.py:54-74 (eps 1e-7, same
// operation order as upa_box_iou).  Exact IoU ties (identical boxes of one class cannot both survive NMS) go to the larger
// label index - what the reversed ascending sort gives for the stable small-array case.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int MP_NT = 10;

struct MatchThr {
  float v[MP_NT];
};

__global__ __launch_bounds__(256) void match_predictions_kernel(const float* det, const int* counts, int max_det, const float* gt,
                                                                const int* ngt, int max_gt, MatchThr thr, float eps,
                                                                unsigned char* tp) {
  extern __shared__ int s_min[];  // [max_gt][MP_NT] smallest detection index claiming (label, threshold)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = min(counts[b], max_det), M = min(ngt[b], max_gt);
  for (int i = tid; i < M * MP_NT; i += 256) s_min[i] = 0x7fffffff;
  __syncthreads();
  const float* D = det + (size_t)b * max_det * 6;
  const float* G = gt + (size_t)b * max_gt * 5;
  unsigned char* T = tp + (size_t)b * max_det * MP_NT;
  // rounds of 256 detections; a detection keeps (best label, best IoU) in registers between the two phases
  for (int base = 0; base < max_det; base += 256) {
    const int d = base + tid;
    int bl = -1;
    float bi = -1.f;
    if (d < N) {
      const float bx1 = D[d * 6], by1 = D[d * 6 + 1], bx2 = D[d * 6 + 2], by2 = D[d * 6 + 3], cls = D[d * 6 + 5];
      for (int l = 0; l < M; ++l) {
        if (G[l * 5] != cls) continue;
        const float ax1 = G[l * 5 + 1], ay1 = G[l * 5 + 2], ax2 = G[l * 5 + 3], ay2 = G[l * 5 + 4];
        const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
        const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
        const float inter = w * h;
        const float iou = inter / ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter + eps);
        if (iou >= bi) { bi = iou; bl = l; }
      }
      if (bl >= 0)
        for (int k = 0; k < MP_NT; ++k)
          if (bi >= thr.v[k]) atomicMin(&s_min[bl * MP_NT + k], d);
    }
    __syncthreads();
    // rows of this round can only be decided once every detection with a smaller index has claimed its label: detections
    // are visited in increasing rounds, and a later round can only LOWER no minimum below an index of this round
    if (d < max_det)
      for (int k = 0; k < MP_NT; ++k) T[d * MP_NT + k] = (d < N && bl >= 0 && bi >= thr.v[k] && s_min[bl * MP_NT + k] == d) ? 1 : 0;
    __syncthreads();
  }
}
}  // namespace

extern "C" int upa_match_predictions(const float* det, const int* counts, int b, int max_det, const float* gt, const int* ngt,
                                     int max_gt, const float* iou_thresholds, int n_thr, unsigned char* tp, void* stream) {
  UPA_CHECK_ARG(det && counts && gt && ngt && iou_thresholds && tp && b > 0 && max_det > 0 && max_gt > 0,
                "match_predictions: bad args");
  UPA_CHECK_ARG(n_thr == MP_NT, "match_predictions: %d IoU thresholds (torch.linspace(0.5, 0.95, 10), detect/val.py:59)", MP_NT);
  UPA_CHECK_ARG((size_t)max_gt * MP_NT * 4 <= 160 * 1024 - 1024, "match_predictions: max_gt too large for the LDS table");
  MatchThr thr;
  for (int k = 0; k < MP_NT; ++k) thr.v[k] = iou_thresholds[k];
  if (hipError_t e = upa_full_lds<match_predictions_kernel>(); e != hipSuccess) return UPA_ELAUNCH;
  hipLaunchKernelGGL(match_predictions_kernel, dim3((unsigned)b), dim3(256), (size_t)max_gt * MP_NT * 4, (hipStream_t)stream, det,
                     counts, max_det, gt, ngt, max_gt, thr, 1e-7f, tp);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}
