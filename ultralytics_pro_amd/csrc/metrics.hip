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
