// Runtime glue of libfocoos_amd.so: ABI/version/error strings, device probe, hipGraph capture/replay
// and HIP-event timing on the caller's stream.
#include <string.h>

#include "conv_common.h"

#include <stdlib.h>

int fx_tune(const char* env_name, int default_value) {
  const char* v = getenv(env_name);
  return (v && *v) ? atoi(v) : default_value;
}

extern "C" int fx_abi_version(void) { return FX_ABI_VERSION; }

#ifndef FX_BUILD_FLAGS
#define FX_BUILD_FLAGS 0
#endif
extern "C" int fx_build_flags(void) { return FX_BUILD_FLAGS | (FX_FP16 ? 2 : 0); }

extern "C" const char* fx_error_string(int code) {
  switch (code) {
    case FX_OK: return "ok";
    case FX_ERR_INVALID_ARGUMENT: return "invalid argument (null pointer, misaligned pointer, or inconsistent shape/stride)";
    case FX_ERR_LAUNCH: return "kernel launch failed (hipGetLastError != hipSuccess)";
    case FX_ERR_UNSUPPORTED: return "unsupported configuration for the gfx950 kernels";
    case FX_ERR_RUNTIME: return "HIP runtime call failed";
    default: return "unknown error code";
  }
}

extern "C" int fx_device_info(int device, int* cu_count, char* arch_name, int arch_name_len) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FX_ERR_RUNTIME;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? FX_OK : FX_ERR_UNSUPPORTED;
}

extern "C" int fx_graph_begin(fx_stream_t stream) {
  return hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal) == hipSuccess ? FX_OK : FX_ERR_RUNTIME;
}

// Fork / join of a side stream (also inside a capture: the side stream joins the capture through the event dependency, so
// the captured graph keeps the two launch sequences as independent branches the GPU may run concurrently).
extern "C" int fx_stream_fork(fx_stream_t main_stream, fx_stream_t side_stream) {
  FX_CHECK_ARG(side_stream);
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return FX_ERR_RUNTIME;
  int rc = FX_OK;
  if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(main_stream)) != hipSuccess) rc = FX_ERR_RUNTIME;
  if (rc == FX_OK && hipStreamWaitEvent(reinterpret_cast<hipStream_t>(side_stream), ev, 0) != hipSuccess) rc = FX_ERR_RUNTIME;
  (void)hipEventDestroy(ev);
  return rc;
}

extern "C" int fx_stream_join(fx_stream_t main_stream, fx_stream_t side_stream) { return fx_stream_fork(side_stream, main_stream); }

extern "C" int fx_graph_end(fx_stream_t stream, void** graph_exec_out) {
  FX_CHECK_ARG(graph_exec_out);
  hipGraph_t graph = nullptr;
  if (hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &graph) != hipSuccess || !graph) return FX_ERR_RUNTIME;
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return FX_ERR_RUNTIME;
  *graph_exec_out = exec;
  return FX_OK;
}

extern "C" int fx_graph_launch(void* graph_exec, fx_stream_t stream) {
  FX_CHECK_ARG(graph_exec);
  return hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? FX_OK
                                                                                                                      : FX_ERR_RUNTIME;
}

extern "C" int fx_graph_destroy(void* graph_exec) {
  if (!graph_exec) return FX_OK;
  return hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? FX_OK : FX_ERR_RUNTIME;
}

extern "C" int fx_graph_time(void* graph_exec, fx_stream_t stream_, int iters, float* ms_avg) {
  FX_CHECK_ARG(graph_exec && iters > 0 && ms_avg);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return FX_ERR_RUNTIME;
  int rc = FX_OK;
  (void)hipEventRecord(e0, stream);
  for (int i = 0; i < iters; ++i)
    if (hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), stream) != hipSuccess) rc = FX_ERR_RUNTIME;
  (void)hipEventRecord(e1, stream);
  if (hipEventSynchronize(e1) != hipSuccess) rc = FX_ERR_RUNTIME;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = FX_ERR_RUNTIME;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_avg = ms / (float)iters;
  return rc;
}
