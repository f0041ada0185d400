// Fused BatchNorm1d(train) -> ReLU -> Dropout for the DNN tower
// (reference deepfm/models/layers/dnn.py:45-55: Linear -> BatchNorm1d -> act -> Dropout).
// (autograd form of the tower; the Linear GEMMs are dfm_gemm_f32) — three small launches forward
// and three backward instead of ~12 elementwise/reduction launches per layer:
//   fwd : partial column sums over 16-row slices -> finalize (mean, rstd, running statistics
//         like nn.BatchNorm1d: unbiased variance, momentum, num_batches_tracked) ->
//         y = gamma*(z-mean)*rstd + beta, a = dropout(relu(y))
//   bwd : dy = g * mask/(1-p) * [y>0]; partial sums of dy, dy*zhat -> finalize (means, d gamma,
//         d beta) -> dz = gamma*rstd*(dy - mean(dy) - zhat*mean(dy*zhat))
// Dropout is counter based: keep(i) = hash(seed, salt, i) >= p*2^32, with the seed read from
// device memory so the same mask is rebuilt in the backward and a captured graph sees a new
// seed every replay.  z is (M, N) row-major, N <= 4096.
#include "common.h"
#include "dropout.h"

using namespace dfm;

namespace {
constexpr int kThreads = 256;
constexpr int kRowsPerSlice = 16;

inline int slices_for(int64_t M) { return static_cast<int>((M + kRowsPerSlice - 1) / kRowsPerSlice); }
inline uint32_t thresh_for(float p) { return dropout_thresh(p); }
}  // namespace

// Three launches each way: partial column sums over 16-row slices (one workgroup per slice,
// coalesced), a tiny finalize that adds the slices in a fixed order, a fully parallel apply.

// partial[s][0][c] = sum (z - shift), partial[s][1][c] = sum (z - shift)^2, shift = z[0][c]
// (sums about row 0 so that var = E[(z-shift)^2] - E[z-shift]^2 does not cancel)
__global__ __launch_bounds__(kThreads) void bn_fwd_partials(const float* __restrict__ z, int64_t M, int N,
                                                            float* __restrict__ partial) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsPerSlice;
  const int64_t r1 = r0 + kRowsPerSlice < M ? r0 + kRowsPerSlice : M;
  for (int c = threadIdx.x; c < N; c += kThreads) {
    const float shift = z[c];
    float s = 0.f, q = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float v = z[r * N + c] - shift;
      s += v;
      q = fmaf(v, v, q);
    }
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * N + c] = s;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * N + c] = q;
  }
}


// Column sums of the two partial planes: a workgroup owns 16 columns; 16 "slice lanes" per
// column each add every 16th slice in order, then lane 0 adds the 16 lane sums in order
// (fixed association -> bitwise reproducible), instead of one thread walking all slices.
constexpr int kFinCols = 16, kFinLanes = 16;
__device__ __forceinline__ bool column_sums(const float* __restrict__ partial, int N, int slices, int* col,
                                            float* s_out, float* q_out) {
  __shared__ float red[2][kFinLanes][kFinCols];
  const int cl = threadIdx.x % kFinCols, sl = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cl;
  float s = 0.f, q = 0.f;
  if (c < N) {
    for (int i = sl; i < slices; i += kFinLanes) {
      s += partial[(static_cast<int64_t>(i) * 2 + 0) * N + c];
      q += partial[(static_cast<int64_t>(i) * 2 + 1) * N + c];
    }
  }
  red[0][sl][cl] = s;
  red[1][sl][cl] = q;
  __syncthreads();
  if (sl != 0 || c >= N) return false;
  s = 0.f; q = 0.f;
  for (int i = 0; i < kFinLanes; ++i) { s += red[0][i][cl]; q += red[1][i][cl]; }
  *col = c; *s_out = s; *q_out = q;
  return true;
}

__global__ __launch_bounds__(kThreads) void bn_fwd_finalize(
    const float* __restrict__ z, int64_t M, int N, int slices, const float* __restrict__ partial,
    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t* __restrict__ num_batches,
    float momentum, float eps, float* __restrict__ mean_rstd) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) num_batches[0] += 1;
  int c;
  float s, q;
  if (!column_sums(partial, N, slices, &c, &s, &q)) return;
  const float ms = s / static_cast<float>(M);
  const float mu = z[c] + ms;
  float var = q / static_cast<float>(M) - ms * ms;      // biased, as BN normalises
  var = var < 0.f ? 0.f : var;
  mean_rstd[c] = mu;
  mean_rstd[N + c] = rsqrtf(var + eps);
  if (running_mean) {
    const float unbiased = M > 1 ? var * static_cast<float>(M) / static_cast<float>(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(kThreads) void bn_relu_dropout_fwd(
    const float* __restrict__ z, int64_t total, int N, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, uint32_t thresh, float inv_keep,
    const int64_t* __restrict__ seed_ptr, int salt, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int64_t seed = seed_ptr ? seed_ptr[0] : 0;
  const int c = static_cast<int>(idx % N);
  const float y = fmaf(gamma[c], (z[idx] - mean_rstd[c]) * mean_rstd[N + c], beta[c]);
  out[idx] = fmaxf(y, 0.f) * drop_scale(seed, salt, idx, thresh, inv_keep);
}

__device__ __forceinline__ float bwd_dy(float g, float zv, float mu, float rs, float ga, float be, int64_t seed,
                                        int salt, int64_t idx, uint32_t thresh, float inv_keep, float* zhat) {
  const float zh = (zv - mu) * rs;
  *zhat = zh;
  const float y = fmaf(ga, zh, be);
  return y > 0.f ? g * drop_scale(seed, salt, idx, thresh, inv_keep) : 0.f;
}

__global__ __launch_bounds__(kThreads) void bn_bwd_partials(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int64_t M, int N, uint32_t thresh,
    float inv_keep, const int64_t* __restrict__ seed_ptr, int salt, float* __restrict__ partial) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsPerSlice;
  const int64_t r1 = r0 + kRowsPerSlice < M ? r0 + kRowsPerSlice : M;
  const int64_t seed = seed_ptr ? seed_ptr[0] : 0;
  for (int c = threadIdx.x; c < N; c += kThreads) {
    const float mu = mean_rstd[c], rs = mean_rstd[N + c], ga = gamma[c], be = beta[c];
    float s = 0.f, q = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t idx = r * N + c;
      float zh;
      const float dy = bwd_dy(g[idx], z[idx], mu, rs, ga, be, seed, salt, idx, thresh, inv_keep, &zh);
      s += dy;
      q = fmaf(dy, zh, q);
    }
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * N + c] = s;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * N + c] = q;
  }
}

// means[0][c] = mean(dy), means[1][c] = mean(dy * zhat); d beta += sum dy, d gamma += sum dy*zhat
__global__ __launch_bounds__(kThreads) void bn_bwd_finalize(const float* __restrict__ partial, int64_t M, int N,
                                                            int slices, float* __restrict__ means,
                                                            float* __restrict__ d_gamma, float* __restrict__ d_beta) {
  int c;
  float s, q;
  if (!column_sums(partial, N, slices, &c, &s, &q)) return;
  means[c] = s / static_cast<float>(M);
  means[N + c] = q / static_cast<float>(M);
  d_beta[c] += s;
  d_gamma[c] += q;
}

__global__ __launch_bounds__(kThreads) void bn_relu_dropout_bwd(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int64_t total, int N,
    const float* __restrict__ means, uint32_t thresh, float inv_keep, const int64_t* __restrict__ seed_ptr,
    int salt, float* __restrict__ dz) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int64_t seed = seed_ptr ? seed_ptr[0] : 0;
  const int c = static_cast<int>(idx % N);
  const float rs = mean_rstd[N + c];
  float zh;
  const float dy = bwd_dy(g[idx], z[idx], mean_rstd[c], rs, gamma[c], beta[c], seed, salt, idx, thresh, inv_keep, &zh);
  dz[idx] = gamma[c] * rs * (dy - means[c] - zh * means[N + c]);
}

extern "C" size_t dfm_bn_workspace_bytes(int64_t batch, int features) {
  return sizeof(float) * (2 * static_cast<size_t>(slices_for(batch)) + 2) * features;
}

extern "C" int dfm_bn_relu_dropout_forward(const float* d_z, int64_t batch, int features, const float* d_gamma,
                                           const float* d_beta, float* d_running_mean, float* d_running_var,
                                           int64_t* d_num_batches, float momentum, float eps, float p_drop,
                                           const int64_t* d_seed, int salt, float* d_out, float* d_mean_rstd,
                                           void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_z && d_gamma && d_beta && d_out && d_mean_rstd && d_workspace, "null argument");
  DFM_REQUIRE(batch > 0 && features > 0 && features <= 65536, "bad shape");
  DFM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  DFM_REQUIRE(p_drop == 0.f || d_seed, "dropout needs a device seed");
  hipStream_t st = as_stream(stream);
  const int slices = slices_for(batch);
  float* partial = static_cast<float*>(d_workspace);
  hipLaunchKernelGGL(bn_fwd_partials, dim3(slices), dim3(kThreads), 0, st, d_z, batch, features, partial);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_fwd_finalize, dim3((features + kFinCols - 1) / kFinCols), dim3(kThreads), 0, st, d_z, batch,
                     features, slices, partial, d_running_mean, d_running_var, d_num_batches, momentum, eps,
                     d_mean_rstd);
  DFM_LAUNCH_CHECK();
  const int64_t total = batch * features;
  hipLaunchKernelGGL(bn_relu_dropout_fwd, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, d_z, total, features, d_mean_rstd, d_gamma, d_beta, thresh_for(p_drop),
                     1.f / (1.f - p_drop), d_seed, salt, d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_bn_relu_dropout_backward(const float* d_g_out, const float* d_z, const float* d_mean_rstd,
                                            const float* d_gamma, const float* d_beta, int64_t batch,
                                            int features, float p_drop, const int64_t* d_seed, int salt,
                                            float* d_g_z, float* d_g_gamma, float* d_g_beta, void* d_workspace,
                                            dfm_stream_t stream) {
  DFM_REQUIRE(d_g_out && d_z && d_mean_rstd && d_gamma && d_beta && d_g_z && d_g_gamma && d_g_beta && d_workspace,
              "null argument");
  DFM_REQUIRE(batch > 0 && features > 0 && features <= 65536, "bad shape");
  hipStream_t st = as_stream(stream);
  const int slices = slices_for(batch);
  float* partial = static_cast<float*>(d_workspace);
  float* means = partial + 2 * static_cast<size_t>(slices) * features;
  const uint32_t th = thresh_for(p_drop);
  const float inv_keep = 1.f / (1.f - p_drop);
  hipLaunchKernelGGL(bn_bwd_partials, dim3(slices), dim3(kThreads), 0, st, d_g_out, d_z, d_mean_rstd, d_gamma,
                     d_beta, batch, features, th, inv_keep, d_seed, salt, partial);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize, dim3((features + kFinCols - 1) / kFinCols), dim3(kThreads), 0, st, partial,
                     batch, features, slices, means, d_g_gamma, d_g_beta);
  DFM_LAUNCH_CHECK();
  const int64_t total = batch * features;
  hipLaunchKernelGGL(bn_relu_dropout_bwd, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, d_g_out, d_z, d_mean_rstd, d_gamma, d_beta, total, features, means, th,
                     inv_keep, d_seed, salt, d_g_z);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
