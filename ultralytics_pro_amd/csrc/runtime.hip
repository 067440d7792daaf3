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
extern "C" int upa_version(void) { return 1; }

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
