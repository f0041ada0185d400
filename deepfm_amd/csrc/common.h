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

// Sum over aligned groups of W consecutive lanes (W = 2, 4, 8, 16: inside one 16-lane DPP row), result in
// every lane, as four v_add_f32_dpp — no LDS.  __shfl_xor lowers to ds_bpermute_b32 + a full lgkmcnt(0)
// wait per step (~500 cycles for a 16-lane butterfly, measured in the CIN epilogue); this is ~30.
// Bitwise equal to the xor butterfly x += shfl_xor(x, 1); ... 2; 4; 8: after the quad steps every lane
// of a quad holds the quad's sum, so "the partner quad / half" is all a mirror has to deliver.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int W>
__device__ __forceinline__ float group_sum(float x) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "group_sum: 1..16 lanes");
  if (W >= 2) x = dpp_add<0xB1>(x);     // quad_perm:[1,0,3,2]
  if (W >= 4) x = dpp_add<0x4E>(x);     // quad_perm:[2,3,0,1]
  if (W >= 8) x = dpp_add<0x141>(x);    // row_half_mirror
  if (W >= 16) x = dpp_add<0x140>(x);   // row_mirror
  return x;
}

// id range guard shared by every gather: never fault, flag instead.
__device__ __forceinline__ int64_t checked_id(int64_t id, int vocab, int32_t* error_flag) {
  if (id < 0 || id >= vocab) {
    if (error_flag) atomicOr(error_flag, 1);
    return 0;
  }
  return id;
}

// Optional extra terms of attn_block_mfma_bwd's d x (attention_mfma.hip); all null: none.
struct AttnGradTail {
  const float* g_flat;         // (B, >= F*D) rows at stride ld_flat
  int64_t ld_flat;
  const float* g_fm;           // (B)
  const float* fm_sum;         // (B, D)
};

}  // namespace dfm
