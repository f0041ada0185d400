// BCEWithLogitsLoss (mean reduction) forward + gradient in one pass
// (reference deepfm/training/trainer.py:59, 221: nn.BCEWithLogitsLoss()):
//   loss = mean( max(z,0) - z*y + log1p(exp(-|z|)) ),   d loss / d z = (sigmoid(z) - y) / B
// Stage 1 writes d z and one partial sum per workgroup; stage 2 adds the partials in a fixed
// order (bitwise reproducible) and divides by B.
#include "common.h"

using namespace dfm;

namespace {
constexpr int kThreads = 256;
constexpr int kPerThread = 4;
inline int64_t bce_blocks(int64_t n) { return (n + kThreads * kPerThread - 1) / (kThreads * kPerThread); }
}

__global__ __launch_bounds__(kThreads) void bce_fwd_bwd(const float* __restrict__ z, const float* __restrict__ y,
                                                        int64_t n, float inv_n, float* __restrict__ dz,
                                                        float* __restrict__ partial) {
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * kPerThread;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      const float zi = z[i], yi = y[i];
      const float e = expf(-fabsf(zi));
      acc += fmaxf(zi, 0.f) - zi * yi + log1pf(e);
      const float sig = zi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      dz[i] = (sig - yi) * inv_n;
    }
  }
  __shared__ float wsum[kThreads / kWave];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(kThreads) void bce_finalize(const float* __restrict__ partial, int n_partials,
                                                         float inv_n, float* __restrict__ loss) {
  __shared__ float wsum[kThreads / kWave];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += kThreads) acc += partial[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_n;
}

extern "C" size_t dfm_bce_workspace_bytes(int64_t n) { return sizeof(float) * static_cast<size_t>(bce_blocks(n > 0 ? n : 1)); }

extern "C" int dfm_bce_with_logits(const float* d_logits, const float* d_labels, int64_t n, float* d_loss,
                                   float* d_g_logits, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_logits && d_labels && d_loss && d_g_logits && d_workspace, "null argument");
  DFM_REQUIRE(n > 0, "empty batch");
  hipStream_t st = as_stream(stream);
  const int64_t blocks = bce_blocks(n);
  float* partial = static_cast<float*>(d_workspace);
  const float inv_n = 1.f / static_cast<float>(n);
  hipLaunchKernelGGL(bce_fwd_bwd, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, d_logits, d_labels, n,
                     inv_n, d_g_logits, partial);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_finalize, dim3(1), dim3(kThreads), 0, st, partial, static_cast<int>(blocks), inv_n, d_loss);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
