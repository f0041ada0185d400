// Shared host/device helpers for the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/deepfm_hip.h"

namespace dfm {

constexpr int kWave = 64;  // gfx950 wavefront

// ---- error plumbing ------------------------------------------------------------------
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define DFM_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return ::dfm::fail(DFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                               \
  } while (0)

#define DFM_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return ::dfm::fail(DFM_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// Launch check that does not synchronise (graph-capture safe).
#define DFM_LAUNCH_CHECK() DFM_HIP_TRY(hipGetLastError())

static inline hipStream_t as_stream(dfm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Per-call pointer table passed by value in the kernel-argument segment.
struct PtrTable {
  const void* p[DFM_MAX_FIELDS];
};
struct GradTable {
  dfm_field_grad g[DFM_MAX_FIELDS];
};

// ---- device helpers ------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// id range guard shared by every gather: never fault, flag instead.
__device__ __forceinline__ int64_t checked_id(int64_t id, int vocab, int32_t* error_flag) {
  if (id < 0 || id >= vocab) {
    if (error_flag) atomicOr(error_flag, 1);
    return 0;
  }
  return id;
}

}  // namespace dfm
