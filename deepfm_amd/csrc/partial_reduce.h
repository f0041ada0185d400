// Fixed-order reductions of per-workgroup partial sums, shared by the stand-alone finish kernels of
// gemm_skinny.hip (weight gradients) and attention_core.hip (LayerNorm d gamma / d beta) and by the one launch
// that finishes several of them together (dfm_partials_finish, gemm_skinny.hip): the three 5-7 us launches an
// attention block's backward ended in are one.
#pragma once

#include "common.h"

namespace dfm {
namespace partials {

// out[e] (+)= sum_blocks partial[block][e]  (fixed order); the last n1 entries go to db.
// 64 elements per workgroup of 256 threads, four threads per element: thread (e, g) adds blocks g, g+4, g+8, ...
// and the four sums are added in the order g = 0,1,2,3.  `blk`: workgroup index inside this job.
__device__ __forceinline__ void wgrad_reduce_body(int blk, const float* __restrict__ partial, int blocks, int n1n2,
                                                  int n1, int N2, float* __restrict__ dW, int64_t ldw,
                                                  float* __restrict__ db, int accumulate, float (*part)[64]) {
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blk * 64 + el;
  const int total = n1n2 + n1;
  float acc = 0.f;
  if (e < total) {
    int s = g;
    for (; s + 28 < blocks; s += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[static_cast<int64_t>(s + 4 * u) * total + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; s < blocks; s += 4) acc += partial[static_cast<int64_t>(s) * total + e];
  }
  part[g][el] = acc;
  __syncthreads();
  if (g != 0 || e >= total) return;
  acc = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
  if (e < n1n2) {
    float* dst = dW + static_cast<int64_t>(e / N2) * ldw + e % N2;
    *dst = accumulate ? *dst + acc : acc;
  } else if (db) {
    db[e - n1n2] = accumulate ? db[e - n1n2] + acc : acc;
  }
}

// d gamma[d] += sum_blocks partial[block][0][d], d beta[d] += sum_blocks partial[block][1][d]: one workgroup of 256
// threads per column d: thread t adds rows t, t + 256, ... and the 256 sums are combined by a fixed binary tree.
__device__ __forceinline__ void layernorm_finalize_body(int d, const float* __restrict__ partial, int blocks, int D,
                                                        float* __restrict__ d_gamma, float* __restrict__ d_beta,
                                                        float (*red)[256]) {
  const int t = threadIdx.x;
  float sg = 0.f, sb = 0.f;
  for (int i = t; i < blocks; i += 256) {
    sg += partial[(static_cast<int64_t>(i) * 2 + 0) * D + d];
    sb += partial[(static_cast<int64_t>(i) * 2 + 1) * D + d];
  }
  red[0][t] = sg;
  red[1][t] = sb;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; }
    __syncthreads();
  }
  if (t == 0) {
    d_gamma[d] += red[0][0];
    d_beta[d] += red[1][0];
  }
}

}  // namespace partials
}  // namespace dfm
