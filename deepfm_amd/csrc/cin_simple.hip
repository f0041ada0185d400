// CIN layer kernels, general shapes, exact fp32 (reference deepfm/models/layers/cin.py:66-105).
//
//   Z[b,h*F+f,d] = hidden[b,h,d] * x0[b,f,d]           (cin.py:84-87)
//   Y[b,c,d]     = relu(bias[c] + sum_k W[c,k] Z[b,k,d]) (cin.py:90-91)
//   out[b,col+c] = sum_d Y[b,c,d]  for the "direct" channels (cin.py:93-102)
//
// These kernels never materialise Z (the reference writes 398-654 MB of it per layer):
// one workgroup owns one sample, keeps x0[b] and hidden[b] in LDS and forms the products
// on the fly.  They accept any F, D, layer size and split layout; the MFMA kernels in
// cin_mfma.hip take over for the shapes they support.
//
// Memory layout shared by both paths: Y_i is stored (B, C_i, D); the hidden input of layer
// i+1 is the slice Y_i[:, direct_i:, :] (or all of Y_i without split), addressed in place
// through (pointer, sample stride).
#include "common.h"

using namespace dfm;

namespace {
constexpr int kThreads = 256;
}

// grid = B; dynamic LDS = (F + H + C) * D floats
__global__ __launch_bounds__(kThreads) void cin_fwd_simple(
    const float* __restrict__ x0, const float* __restrict__ hidden, int64_t hidden_stride,
    const float* __restrict__ W, const float* __restrict__ bias, int F, int H, int C, int D,
    int direct, float* __restrict__ Y, float* __restrict__ out, int out_stride, int out_col) {
  extern __shared__ float lds[];
  float* xs = lds;            // F*D
  float* hs = xs + F * D;     // H*D
  float* ys = hs + H * D;     // C*D
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < F * D; i += kThreads) xs[i] = x0[b * F * D + i];
  for (int i = tid; i < H * D; i += kThreads) hs[i] = hidden[b * hidden_stride + i];
  __syncthreads();
  const int K = H * F;
  for (int o = tid; o < C * D; o += kThreads) {
    const int c = o / D, d = o % D;
    const float* wrow = W + static_cast<int64_t>(c) * K;
    float acc = bias[c];
    for (int h = 0; h < H; ++h) {
      const float hv = hs[h * D + d];
      const float* w = wrow + h * F;
      for (int f = 0; f < F; ++f) acc = fmaf(w[f], hv * xs[f * D + d], acc);
    }
    const float y = fmaxf(acc, 0.f);
    ys[o] = y;
    Y[(b * C + c) * D + d] = y;
  }
  __syncthreads();
  for (int c = tid; c < direct; c += kThreads) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += ys[c * D + d];
    out[b * out_stride + out_col + c] = s;
  }
}

// Per-sample backward of one layer.  grid = B; dynamic LDS = (F + H + C) * D floats.
//   dY = relu'(Y) * (g_out broadcast over d on direct channels  +  d_next on the channels that
//        fed the next layer);  stored to dY_out for the weight-gradient kernel
//   G[k,d] = sum_c W[c,k] dY[c,d]
//   d_hidden[h,d]  = sum_f x0[f,d] G[(h,f),d]     -> d_hidden_out (or += d_x0 when layer 0)
//   d_x0[f,d]     += sum_h hidden[h,d] G[(h,f),d]
__global__ __launch_bounds__(kThreads) void cin_bwd_simple(
    const float* __restrict__ x0, const float* __restrict__ hidden, int64_t hidden_stride,
    const float* __restrict__ W, const float* __restrict__ Y, int F, int H, int C, int D,
    int direct, int next_off, int next_count, const float* __restrict__ g_out, int out_stride,
    int out_col, const float* __restrict__ d_next, float* __restrict__ dY_out,
    float* __restrict__ d_hidden_out, float* __restrict__ d_x0, int layer0) {
  extern __shared__ float lds[];
  float* xs = lds;
  float* hs = xs + F * D;
  float* gs = hs + H * D;  // dY, C*D
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < F * D; i += kThreads) xs[i] = x0[b * F * D + i];
  for (int i = tid; i < H * D; i += kThreads) hs[i] = hidden[b * hidden_stride + i];
  for (int o = tid; o < C * D; o += kThreads) {
    const int c = o / D, d = o % D;
    float g = 0.f;
    if (c < direct) g = g_out[b * out_stride + out_col + c];
    if (d_next && c >= next_off && c < next_off + next_count)
      g += d_next[(b * next_count + (c - next_off)) * D + d];
    g = Y[(b * C + c) * D + d] > 0.f ? g : 0.f;
    gs[o] = g;
    dY_out[(b * C + c) * D + d] = g;
  }
  __syncthreads();
  const int K = H * F;
  // pass A: gradient w.r.t. the hidden input
  for (int o = tid; o < H * D; o += kThreads) {
    const int h = o / D, d = o % D;
    float dh = 0.f;
    for (int f = 0; f < F; ++f) {
      const int k = h * F + f;
      float gsum = 0.f;
      for (int c = 0; c < C; ++c) gsum = fmaf(W[static_cast<int64_t>(c) * K + k], gs[c * D + d], gsum);
      dh = fmaf(xs[f * D + d], gsum, dh);
    }
    if (layer0) d_x0[(b * F + h) * D + d] += dh;       // hidden_0 is x0 itself (H == F)
    else d_hidden_out[(b * H + h) * D + d] = dh;
  }
  // pass B: gradient w.r.t. x0 (same thread owns element (f,d) in both passes when layer0)
  for (int o = tid; o < F * D; o += kThreads) {
    const int f = o / D, d = o % D;
    float dx = 0.f;
    for (int h = 0; h < H; ++h) {
      const int k = h * F + f;
      float gsum = 0.f;
      for (int c = 0; c < C; ++c) gsum = fmaf(W[static_cast<int64_t>(c) * K + k], gs[c * D + d], gsum);
      dx = fmaf(hs[h * D + d], gsum, dx);
    }
    d_x0[(b * F + f) * D + d] += dx;
  }
}

// Weight gradient partials: dW[c,k] = sum_{b,d} dY[b,c,d] hidden[b,h,d] x0[b,f,d] over the
// samples of one batch slice.  grid = (ceil(C*K/256), slices); partial (slices, C*K).
__global__ __launch_bounds__(kThreads) void cin_wgrad_simple(
    const float* __restrict__ x0, const float* __restrict__ hidden, int64_t hidden_stride,
    const float* __restrict__ dY, int64_t B, int F, int H, int C, int D, int slices,
    float* __restrict__ partial) {
  const int K = H * F;
  const int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (o >= static_cast<int64_t>(C) * K) return;
  const int c = static_cast<int>(o / K), k = static_cast<int>(o % K);
  const int h = k / F, f = k % F;
  const int64_t per = (B + slices - 1) / slices;
  const int64_t b0 = blockIdx.y * per, b1 = b0 + per < B ? b0 + per : B;
  float acc = 0.f;
  for (int64_t b = b0; b < b1; ++b) {
    const float* g = dY + (b * C + c) * D;
    const float* hv = hidden + b * hidden_stride + h * D;
    const float* xv = x0 + (b * F + f) * D;
    for (int d = 0; d < D; ++d) acc = fmaf(g[d], hv[d] * xv[d], acc);
  }
  partial[static_cast<int64_t>(blockIdx.y) * C * K + o] = acc;
}

// dW[o] = sum_s partial[s][o] (fixed order);  grid over C*K
__global__ __launch_bounds__(kThreads) void cin_wgrad_reduce(const float* __restrict__ partial,
                                                             int64_t n, int slices,
                                                             float* __restrict__ dW) {
  const int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (o >= n) return;
  float acc = 0.f;
  for (int s = 0; s < slices; ++s) acc += partial[static_cast<int64_t>(s) * n + o];
  dW[o] += acc;
}

// db[c] = sum_{b,d} dY[b,c,d].  Stage 1: each workgroup sweeps a batch slice with fully
// coalesced reads (a sample's (C, D) block is contiguous) and leaves C partial sums; stage 2
// adds the slices in a fixed order.  Dynamic LDS = C*D floats.
__global__ __launch_bounds__(kThreads) void cin_bias_partial(const float* __restrict__ dY, int64_t B,
                                                             int C, int D, int slices,
                                                             float* __restrict__ partial) {
  extern __shared__ float tmp[];
  const int CD = C * D;
  const int64_t per = (B + slices - 1) / slices;
  const int64_t b0 = blockIdx.x * per, b1 = b0 + per < B ? b0 + per : B;
  for (int e = threadIdx.x; e < CD; e += kThreads) {
    float acc = 0.f;
    for (int64_t b = b0; b < b1; ++b) acc += dY[b * CD + e];
    tmp[e] = acc;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kThreads) {
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += tmp[c * D + d];
    partial[static_cast<int64_t>(blockIdx.x) * C + c] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void cin_bias_reduce(const float* __restrict__ partial, int C,
                                                            int slices, float* __restrict__ db) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int s = 0; s < slices; ++s) acc += partial[static_cast<int64_t>(s) * C + c];
  db[c] += acc;
}

namespace dfm {
constexpr int kWgradSlices = 8;

int cin_simple_forward_layer(const float* x0, const float* hidden, int64_t hidden_stride,
                             const float* W, const float* bias, int64_t B, int F, int H, int C, int D,
                             int direct, float* Y, float* out, int out_stride, int out_col,
                             hipStream_t st) {
  const size_t lds = sizeof(float) * static_cast<size_t>(F + H + C) * D;
  DFM_REQUIRE(lds <= 160 * 1024, "CIN layer too large for one workgroup's LDS (%zu bytes)", lds);
  if (lds > 64 * 1024)
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwd_simple),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(cin_fwd_simple, dim3(static_cast<unsigned>(B)), dim3(kThreads), lds, st, x0, hidden,
                     hidden_stride, W, bias, F, H, C, D, direct, Y, out, out_stride, out_col);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

constexpr int kBiasSlices = 256;
size_t cin_bias_grad_workspace_bytes(int C) { return sizeof(float) * kBiasSlices * static_cast<size_t>(C); }

int cin_bias_grad_launch(const float* dY, int64_t B, int C, int D, float* db, float* partial, hipStream_t st) {
  const int slices = B < kBiasSlices ? static_cast<int>(B) : kBiasSlices;
  hipLaunchKernelGGL(cin_bias_partial, dim3(slices), dim3(kThreads), sizeof(float) * C * D, st, dY, B, C, D,
                     slices, partial);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(cin_bias_reduce, dim3((C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, partial, C,
                     slices, db);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int cin_simple_backward_layer(const float* x0, const float* hidden, int64_t hidden_stride,
                              const float* W, const float* Y, int64_t B, int F, int H, int C, int D,
                              int direct, int next_off, int next_count, const float* g_out,
                              int out_stride, int out_col, const float* d_next, float* dY,
                              float* d_hidden_out, float* d_x0, int layer0, float* dW, float* db,
                              float* partial, hipStream_t st) {
  const size_t lds = sizeof(float) * static_cast<size_t>(F + H + C) * D;
  DFM_REQUIRE(lds <= 160 * 1024, "CIN layer too large for one workgroup's LDS (%zu bytes)", lds);
  if (lds > 64 * 1024)
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bwd_simple),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(cin_bwd_simple, dim3(static_cast<unsigned>(B)), dim3(kThreads), lds, st, x0, hidden,
                     hidden_stride, W, Y, F, H, C, D, direct, next_off, next_count, g_out, out_stride,
                     out_col, d_next, dY, d_hidden_out, d_x0, layer0);
  DFM_LAUNCH_CHECK();
  const int64_t n = static_cast<int64_t>(C) * H * F;
  const int slices = B < kWgradSlices ? static_cast<int>(B) : kWgradSlices;
  hipLaunchKernelGGL(cin_wgrad_simple, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads), slices),
                     dim3(kThreads), 0, st, x0, hidden, hidden_stride, dY, B, F, H, C, D, slices, partial);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(cin_wgrad_reduce, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, partial, n, slices, dW);
  DFM_LAUNCH_CHECK();
  // the weight-gradient partials are consumed by now: reuse the buffer for the bias partials
  return cin_bias_grad_launch(dY, B, C, D, db, partial, st);
}
}  // namespace dfm
