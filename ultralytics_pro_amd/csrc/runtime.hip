// Library plumbing: error string, version, HIP-graph capture helpers (hipGraph replay instead of a tracing compiler:
// the Python layer executor - BaseModel._predict_once, ultralytics/nn/tasks.py:1046-1085 - is walked once under
// stream capture and replayed per batch).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void upa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* upa_last_error(void) { return g_err; }

__global__ void upa_zero_words_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}
extern "C" int upa_version(void) { return 2; }  // 2: upa_opts argument on the dispatching entry points (round 3)
extern "C" size_t upa_opts_size(void) { return sizeof(upa_opts); }

// Device -> pinned host copy as a KERNEL (the device writes the host-mapped allocation through its unified address): a step
// captured into a hipGraph can hand its detections to the host with no memcpy node (with several graphs of the step in flight
// the runtime's memset / memcpy nodes have misbehaved, see upa_zero_words) and no extra host call per step.
__global__ void upa_copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int upa_copy_to_host(const void* src_dev, void* dst_pinned, size_t bytes, void* stream) {
  UPA_CHECK_ARG(src_dev && dst_pinned && bytes % 4 == 0 && ((uintptr_t)src_dev % 4) == 0 && ((uintptr_t)dst_pinned % 4) == 0,
                "copy_to_host: pointers and size must be 4-byte aligned");
  if (bytes == 0) return UPA_OK;
  const long n = (long)(bytes / 4);
  long blocks = (n + 255) / 256;
  if (blocks > 64) blocks = 64;  // a few hundred KB over the host link: more workgroups only add launch ramp
  hipLaunchKernelGGL(upa_copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)src_dev,
                     (unsigned*)dst_pinned, n);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

// One workgroup per image: the count, then only the rows that hold detections (the rest of the fixed-shape buffer never crosses the
// host link).  Rows past the count keep whatever the host buffer held - the count says how many are valid, as on the device.
__global__ void upa_results_to_host_kernel(const unsigned* __restrict__ rows, const int* __restrict__ counts, int max_rows,
                                           int row_words, unsigned* __restrict__ dst_rows, int* __restrict__ dst_counts) {
  const int b = blockIdx.x;
  int c = counts[b];
  if (threadIdx.x == 0) dst_counts[b] = c;
  c = c < 0 ? 0 : (c > max_rows ? max_rows : c);
  const long base = (long)b * max_rows * row_words;
  for (int i = threadIdx.x; i < c * row_words; i += blockDim.x) dst_rows[base + i] = rows[base + i];
}
extern "C" int upa_results_to_host(const void* rows_dev, const int* counts_dev, int batch, int max_rows, int row_bytes,
                                   void* rows_pinned, int* counts_pinned, void* stream) {
  UPA_CHECK_ARG(rows_dev && counts_dev && rows_pinned && counts_pinned && batch > 0 && max_rows > 0 && row_bytes > 0 && row_bytes % 4 == 0 &&
                    ((uintptr_t)rows_dev % 4) == 0 && ((uintptr_t)rows_pinned % 4) == 0,
                "results_to_host: bad args (rows of whole 4-byte words)");
  hipLaunchKernelGGL(upa_results_to_host_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const unsigned*)rows_dev,
                     counts_dev, max_rows, row_bytes / 4, (unsigned*)rows_pinned, counts_pinned);
  UPA_LAUNCH_CHECK();
  return UPA_OK;
}

extern "C" int upa_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { upa_set_error("graph_begin: %s", hipGetErrorString(e)); return UPA_ELAUNCH; }
  return UPA_OK;
}

extern "C" int upa_graph_end(void* stream, void** graph_exec_out) {
  UPA_CHECK_ARG(graph_exec_out, "graph_end: null out pointer");
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || !graph) { upa_set_error("graph_end: %s", hipGetErrorString(e)); return UPA_ELAUNCH; }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { upa_set_error("graph_instantiate: %s", hipGetErrorString(e)); return UPA_ELAUNCH; }
  *graph_exec_out = (void*)exec;
  return UPA_OK;
}

extern "C" int upa_graph_launch(void* graph_exec, void* stream) {
  UPA_CHECK_ARG(graph_exec, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) { upa_set_error("graph_launch: %s", hipGetErrorString(e)); return UPA_ELAUNCH; }
  return UPA_OK;
}

extern "C" int upa_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return UPA_OK;
}
