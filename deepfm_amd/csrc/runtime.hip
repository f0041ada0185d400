// Error plumbing and device queries of the C ABI.
#include "common.h"

namespace dfm {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace dfm

__global__ void dfm_empty_kernel() {}

extern "C" {

// timing calibration only (bench.py): an empty kernel launched like every other entry point
int dfm_debug_empty_launch(dfm_stream_t stream) {
  hipLaunchKernelGGL(dfm_empty_kernel, dim3(1), dim3(64), 0, dfm::as_stream(stream));
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// ---- graph plumbing -------------------------------------------------------------------------------
// While `stream` is being captured: the node the next captured operation would depend on, i.e. the node
// of the operation enqueued last (call it right after the launch whose node is wanted).
int dfm_graph_last_node(dfm_stream_t stream, void** node_out) {
  DFM_REQUIRE(node_out, "null argument");
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t n = 0;
  DFM_HIP_TRY(hipStreamGetCaptureInfo_v2(dfm::as_stream(stream), &status, &id, &graph, &deps, &n));
  DFM_REQUIRE(status == hipStreamCaptureStatusActive, "stream is not being captured");
  DFM_REQUIRE(n == 1 && deps, "expected exactly one dependency node, found %zu", n);
  *node_out = deps[0];
  return DFM_OK;
}

int dfm_abi_version(void) { return DFM_ABI_VERSION; }

const char* dfm_last_error(void) { return dfm::last_error_buf(); }

int dfm_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  DFM_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  DFM_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DFM_OK;
}

}  // extern "C"
