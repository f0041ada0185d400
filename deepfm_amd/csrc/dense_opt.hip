// Dense-parameter tail of the training step on ONE flat fp32 buffer (DENSE-field Linears,
// DNN, heads, CIN, attention weights): L2 term, global gradient norm, clip and Adam —
// reference trainer.py:224-237 (get_l2_reg_loss base.py:78-83, clip_grad_norm_, Adam).
//   prepare : g[i] += 2*l2*p[i] for i < n_l2 (the embedding parameters);  partial |g|^2
//   finalize: total = sum(partials) in fixed order;  clip = min(1, max_norm/(sqrt(total)+1e-6))
//   adam    : torch.optim.Adam update with g*clip
#include "tail_bodies.h"

using namespace dfm;

namespace {
constexpr int kPrepBlock = 256;
constexpr int kPrepPerThread = tail::kPrepPerThread;  // elements per thread
inline int64_t prep_blocks(int64_t n) {
  const int64_t per_block = static_cast<int64_t>(kPrepBlock) * kPrepPerThread;
  return (n + per_block - 1) / per_block;
}
}  // namespace

__global__ __launch_bounds__(kPrepBlock) void dense_prepare_kernel(
    float* __restrict__ g, const float* __restrict__ p, int64_t n, int64_t n_l2, float l2,
    float* __restrict__ partial) {
  tail::dense_prepare_body(blockIdx.x, g, p, n, n_l2, l2, partial);
}

__global__ __launch_bounds__(1024) void norm_finalize_kernel(const float* __restrict__ partial, int n,
                                                             float max_norm, float* __restrict__ sq_out,
                                                             float* __restrict__ clip_out,
                                                             int32_t* __restrict__ step_tick,
                                                             int64_t* __restrict__ seed_tick) {
  __shared__ float wsum[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += partial[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += wsum[i];
    sq_out[0] = tot;
    if (clip_out) {
      float c = 1.f;
      if (max_norm > 0.f) c = fminf(1.f, max_norm / (sqrtf(tot) + 1e-6f));
      clip_out[0] = c;
    }
    // the step's single-workgroup kernel doubles as its clock: Adam's step count (read by the
    // update kernels that follow) and the dropout seed (read by the next step) advance here
    if (step_tick) step_tick[0] += 1;
    if (seed_tick) seed_tick[0] += 1;
  }
}

__global__ __launch_bounds__(256) void dense_adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ g,
                                                         int64_t n, const float* __restrict__ clip_coef,
                                                         float lr, float b1, float b2, float eps,
                                                         const int32_t* __restrict__ step_ptr, float* __restrict__ g_zero) {
  tail::dense_adam_body(blockIdx.x, p, m, v, g, n, clip_coef, lr, b1, b2, eps, step_ptr, g_zero);
}

extern "C" int64_t dfm_dense_num_partials(int64_t n) { return n > 0 ? prep_blocks(n) : 0; }

extern "C" int dfm_dense_grad_prepare(float* d_g, const float* d_p, int64_t n, int64_t n_l2, float l2,
                                      float* d_partials, dfm_stream_t stream) {
  DFM_REQUIRE(n >= 0 && n_l2 >= 0 && n_l2 <= n, "bad sizes");
  if (n == 0) return DFM_OK;
  DFM_REQUIRE(d_g && d_p && d_partials, "null argument");
  hipLaunchKernelGGL(dense_prepare_kernel, dim3(static_cast<unsigned>(prep_blocks(n))), dim3(kPrepBlock), 0,
                     as_stream(stream), d_g, d_p, n, n_l2, l2, d_partials);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_grad_norm_finalize(const float* d_partials, int64_t num_partials, float max_norm,
                                      float* d_sq_norm, float* d_clip_coef, int32_t* d_step_tick,
                                      int64_t* d_seed_tick, dfm_stream_t stream) {
  DFM_REQUIRE(d_partials && d_sq_norm, "null argument");
  DFM_REQUIRE(num_partials >= 0 && num_partials < (int64_t(1) << 31), "bad partial count");
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d_partials,
                     static_cast<int>(num_partials), max_norm, d_sq_norm, d_clip_coef, d_step_tick, d_seed_tick);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_dense_adam(float* d_p, float* d_m, float* d_v, float* d_g, int64_t n,
                              const float* d_clip_coef, float lr, float beta1, float beta2, float eps,
                              const int32_t* d_step, int zero_grad, dfm_stream_t stream) {
  DFM_REQUIRE(n >= 0, "bad size");
  if (n == 0) return DFM_OK;
  DFM_REQUIRE(d_p && d_m && d_v && d_g && d_step, "null argument");
  hipLaunchKernelGGL(dense_adam_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_p, d_m, d_v, d_g, n, d_clip_coef, lr, beta1, beta2, eps, d_step,
                     zero_grad ? d_g : static_cast<float*>(nullptr));
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
