// Exact-fp32 GEMM on the gfx950 matrix cores for the DNN tower's Linear layers
// (reference dnn.py:45-47: nn.Linear forward/backward = aten::addmm / mm).
//
//   C[m,n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n])
// Each operand is "K-contiguous" (element (r,k) at base[r*ld + k]) or "K-strided" (element (r,k)
// at base[k*ld + r]); the three GEMMs of a Linear layer are then this one kernel, no transposes:
//   forward  z  = x W^T + b : A = x  (KC),  B = W (KC)
//   d input  dx = dz W      : A = dz (KC),  B = W (K-strided: element (k_in, n) at W[n*K + k_in])
//   d weight dW += dz^T x   : A = dz (K-strided over the batch), B = x (K-strided), accumulate
//
// v_mfma_f32_32x32x2_f32: fp32 inputs, fp32 accumulate — bit-for-bit a k-ordered fmaf chain
// (MI355X_MICROARCH.md, matrix cores), 64 FLOP/clk/SIMD = the fp32 vector peak, so nothing is
// rounded to bf16 and the 1e-4 parity bar is untouched.  The shapes are small (batch 4096 x
// <= 1024), so the tile is small too: a workgroup of 8 waves owns a 64 x 64 output tile (four
// 32 x 32 MFMA tiles x two k-halves; gemm_core.h has the tile loop, shared with the fused tower
// kernels of tower.hip).  Small outputs with a long reduction (dW: reduce over the batch) are
// split over workgroup rows into slabs that are added in a fixed order.
#include "gemm_core.h"

using namespace dfm;
using namespace dfm::gemm;

// grid (tiles_n, tiles_m, splits)
template <bool A_KC, bool B_KC, bool A_FAST, bool B_FAST>
__global__ __launch_bounds__(kThreads) void gemm_f32_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
    int64_t ldc, int M, int N, int K, const float* __restrict__ bias, int accumulate, int k_per_split,
    float* __restrict__ slabs) {
  __shared__ Smem sm;
  const TilePos pos;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = blockIdx.z * k_per_split;
  const int ke = kb + k_per_split < K ? kb + k_per_split : K;
  f32x16 acc = {};
  mainloop<A_KC, B_KC, A_FAST, B_FAST>(A, lda, B, ldb, M, N, m0, n0, kb, ke, sm, pos, acc);
  if (pos.khalf == 1) return;
  const int n = n0 + pos.col();
  if (n >= N) return;
  if (slabs) {
    float* sl = slabs + static_cast<int64_t>(blockIdx.z) * M * N;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + pos.row(reg);
      if (m < M) sl[static_cast<int64_t>(m) * N + n] = acc[reg];
    }
    return;
  }
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + pos.row(reg);
    if (m < M) {
      float* dst = C + static_cast<int64_t>(m) * ldc + n;
      const float v = acc[reg] + bv;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

// C[m,n] = (accumulate ? C : 0) + bias[n] + sum_s slabs[s][m][n]   (fixed order)
__global__ __launch_bounds__(256) void gemm_f32_reduce(const float* __restrict__ slabs, int splits, int M, int N,
                                                       float* __restrict__ C, int64_t ldc,
                                                       const float* __restrict__ bias, int accumulate) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= static_cast<int64_t>(M) * N) return;
  const int m = static_cast<int>(i / N), n = static_cast<int>(i % N);
  // fixed summation order; loads issued 8 at a time so they overlap instead of chaining
  float acc = 0.f;
  int sp = 0;
  for (; sp + 8 <= splits; sp += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[static_cast<int64_t>(sp + u) * M * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += t[u];
  }
  for (; sp < splits; ++sp) acc += slabs[static_cast<int64_t>(sp) * M * N + i];
  float* dst = C + static_cast<int64_t>(m) * ldc + n;
  *dst = (accumulate ? *dst : 0.f) + (bias ? bias[n] : 0.f) + acc;
}

namespace dfm {   // gemm_skinny.hip
bool gemm_rows_try(const float* A, int64_t lda, const float* W, int64_t ldw, bool w_kc, float* C, int64_t ldc,
                   int64_t M, int N, int K, const float* bias, int accumulate, hipStream_t st);
bool gemm_wgrad_supported(int64_t M, int N1, int N2);
size_t gemm_wgrad_workspace_bytes(int64_t M, int N1, int N2);
bool gemm_wgrad_try(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int N1, int N2,
                    float* dW, int64_t ldw, float* db, int accumulate, void* workspace, hipStream_t st, bool defer);
}  // namespace dfm

namespace {
int pick_splits(int m, int n, int k) {
  const int64_t tiles = static_cast<int64_t>((m + BM - 1) / BM) * ((n + BN - 1) / BN);
  if (tiles >= 96 || k <= 8 * BK) return 1;      // only small outputs with a long reduction (dW)
  int64_t s = (256 + tiles - 1) / tiles;         // aim for >= 256 workgroups (1024 waves)
  const int64_t max_s = k / (4 * BK);            // at least 4 k-tiles per split
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : static_cast<int>(s);
}
}  // namespace

extern "C" size_t dfm_gemm_workspace_bytes(int m, int n, int k) {
  const int s = pick_splits(m, n, k);
  size_t bytes = s > 1 ? sizeof(float) * static_cast<size_t>(s) * m * n : 0;
  // (the caller does not say which operand layout it will use: cover the streamed weight gradient too)
  if (dfm::gemm_wgrad_supported(k, m, n)) {
    const size_t w = dfm::gemm_wgrad_workspace_bytes(k, m, n);
    bytes = w > bytes ? w : bytes;
  }
  return bytes;
}

extern "C" int dfm_gemm_f32(const float* d_a, int64_t lda, int a_k_contiguous, const float* d_b, int64_t ldb,
                            int b_k_contiguous, float* d_c, int64_t ldc, int m, int n, int k,
                            const float* d_bias, int accumulate, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_a && d_b && d_c, "null argument");
  DFM_REQUIRE(m > 0 && n > 0 && k > 0, "bad shape");
  // many rows against a tiny weight (the attention projections and their gradients): gemm_skinny.hip
  if (a_k_contiguous && dfm::gemm_rows_try(d_a, lda, d_b, ldb, b_k_contiguous != 0, d_c, ldc, m, n, k, d_bias,
                                           accumulate, as_stream(stream))) {
    DFM_LAUNCH_CHECK();
    return DFM_OK;
  }
  if (!a_k_contiguous && !b_k_contiguous && !d_bias &&
      dfm::gemm_wgrad_try(d_a, lda, d_b, ldb, k, m, n, d_c, ldc, nullptr, accumulate, d_workspace,
                          as_stream(stream), false)) {
    DFM_LAUNCH_CHECK();
    return DFM_OK;
  }
  int splits = d_workspace ? pick_splits(m, n, k) : 1;
  int k_per_split = ((k + splits - 1) / splits + BK - 1) / BK * BK;
  splits = (k + k_per_split - 1) / k_per_split;
  float* slabs = splits > 1 ? static_cast<float*>(d_workspace) : nullptr;
  const dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM, splits), block(kThreads);
  hipStream_t st = as_stream(stream);
  // branch-free tile loads need all-or-nothing 16-byte pieces (see load_slice)
  const bool af = operand_fast(d_a, lda, a_k_contiguous != 0, m, k);
  const bool bf = operand_fast(d_b, ldb, b_k_contiguous != 0, n, k);
#define DFM_GEMM_LAUNCH(AK, BK_, AF, BF)                                                                    \
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, AF, BF>), grid, block, 0, st, d_a, lda, d_b, ldb, d_c, ldc, m, \
                     n, k, d_bias, accumulate, k_per_split, slabs)
#define DFM_GEMM_FAST(AK, BK_)                         \
  do {                                                 \
    if (af && bf) DFM_GEMM_LAUNCH(AK, BK_, true, true); \
    else DFM_GEMM_LAUNCH(AK, BK_, false, false);       \
  } while (0)
  if (a_k_contiguous && b_k_contiguous) DFM_GEMM_FAST(true, true);
  else if (a_k_contiguous) DFM_GEMM_FAST(true, false);
  else if (b_k_contiguous) DFM_GEMM_FAST(false, true);
  else DFM_GEMM_FAST(false, false);
#undef DFM_GEMM_FAST
#undef DFM_GEMM_LAUNCH
  DFM_LAUNCH_CHECK();
  if (slabs) {
    const int64_t total = static_cast<int64_t>(m) * n;
    hipLaunchKernelGGL(gemm_f32_reduce, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, slabs,
                       splits, m, n, d_c, ldc, d_bias, accumulate);
    DFM_LAUNCH_CHECK();
  }
  return DFM_OK;
}
