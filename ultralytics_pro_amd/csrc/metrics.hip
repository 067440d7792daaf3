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
