// Exact-fp32 MFMA GEMMs for the "many rows, tiny weight" shapes of the attention projections
// (attention.py:95-97, :115 and their autograd: B*F = 159 744 rows against 32 ... 192 columns).
// These are HBM-bound (the QKV projection reads 20 MB and writes 123 MB for 2 GFLOP); the tiled
// kernel of gemm_f32.hip spends them in per-tile prologues (one 32-deep slice per 64 x 64 tile).
// Here nothing is staged through LDS and there is no workgroup barrier:
//
//   rows kernel    C[M,N] (+)= A[M,K] W^T (+ bias)       N <= 192, K <= 192, N*K <= 6144
//     The whole weight lives in REGISTERS as MFMA B fragments (<= 96 VGPRs), loaded once per wave.
//     A wave walks 32-row tiles of A: one float4 load per lane and 8 k, straight into the A-fragment
//     layout (the k order inside a v_mfma_f32_32x32x2 chain is free as long as A and B agree: lane
//     half h takes k = 8j + 4h .. 8j + 4h + 3, so the loads are 16-byte row pieces), N/32
//     independent accumulators, 128-byte coalesced row stores.
//   weight-gradient kernel   dW[N1,N2] (+)= G[M,N1]^T X[M,N2],  db[N1] (+)= sum_m G[m,:]
//     both operands stream (coalesced 128-byte row pieces, the reduction index m is the MFMA k);
//     every wave keeps the whole dW in accumulators over its row chunks, the four waves of a
//     workgroup are added through LDS, the per-workgroup partials by a second kernel in a fixed
//     order (bitwise reproducible).  db comes from the A fragments with VALU adds — no ones-column
//     GEMM.
// (One workgroup per CU with a second register set for the next tile / row chunk was slower: beyond
// 256 VGPRs the extra registers are AGPRs, which loads and VALU reach only through copies —
// weight gradient 59 -> 167 us, K = 192 rows kernel 55 -> 69 us.)
#include "common.h"
#include "partial_reduce.h"

using namespace dfm;

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;
constexpr int kMaxResidentWaves = 2 * 256 * kWaves;   // 2 workgroups per CU
constexpr int kWgradWaves = 2 * 256 * kWaves;         // weight gradient: two workgroups per CU

// ---- rows kernel -----------------------------------------------------------------------------
template <int NT, int K>
__global__ __launch_bounds__(kWaves * 64, 2) void gemm_rows_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw, int w_kc,
    float* __restrict__ C, int64_t ldc, int64_t M, const float* __restrict__ bias, int accumulate,
    int tiles_per_wave) {
  constexpr int KH = K / 2;                        // k values per lane (its half of every k pair)
  const int lane = lane_id(), wave = wave_id_uniform();
  const int r = lane & 31, hf = lane >> 5;
  const int64_t wid = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  const int64_t ntiles = (M + 31) / 32;
  // B fragments: b[t][i] = W(n = 32 t + r, k = kmap(i)), kmap(i) = 8 (i / 4) + 4 hf + i % 4
  float b[NT][KH];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 32 + r;
    if (w_kc) {
#pragma unroll
      for (int j = 0; j < K / 8; ++j) {
        const float4 v = ld4(W + n * ldw + 8 * j + 4 * hf);
        b[t][4 * j] = v.x; b[t][4 * j + 1] = v.y; b[t][4 * j + 2] = v.z; b[t][4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < KH; ++i) b[t][i] = W[(8 * (i >> 2) + 4 * hf + (i & 3)) * ldw + n];
    }
  }
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[t * 32 + r] : 0.f;

  // A fragments of a tile: one float4 per lane and 8 k (rows clamped: a dead row only reaches a dead
  // output row).  With K <= 64 the next tile's fragments are requested before this tile's MFMAs.
  constexpr bool PREFETCH = KH <= 32;
  auto load_a = [&](int64_t tile, float (&dst)[KH]) __attribute__((always_inline)) {
    const int64_t row = tile * 32 + r;
    const float* arow = A + (row < M ? row : M - 1) * lda + 4 * hf;
#pragma unroll
    for (int j = 0; j < K / 8; ++j) {
      const float4 v = ld4(arow + 8 * j);
      dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
    }
  };
  const int64_t tile0 = wid * tiles_per_wave;
  float a[KH], an[KH];
  if (PREFETCH && tile0 < ntiles) load_a(tile0, an);
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int64_t tile = tile0 + it;                         // wave-uniform
    if (tile >= ntiles) break;
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < KH; ++i) a[i] = an[i];
      if (it + 1 < tiles_per_wave && tile + 1 < ntiles) load_a(tile + 1, an);
    } else {
      load_a(tile, a);
    }
    // column tiles in groups of at most three: 48 accumulator registers live at a time (six tiles
    // at once needed 324 VGPRs: one wave per SIMD, or spills at two)
    constexpr int TG = NT > 3 ? 3 : NT;
    float* cbase = C + (tile * 32 + 4 * hf) * ldc + r;
    const int64_t rlim = M - (tile * 32 + 4 * hf);             // row offset (q & 3) + 8 (q >> 2) must stay below
#pragma unroll
    for (int g0 = 0; g0 < NT; g0 += TG) {
      f32x16 acc[TG];
#pragma unroll
      for (int t = 0; t < TG; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = bv[g0 + t];
      }
#pragma unroll
      for (int i = 0; i < KH; ++i) {
#pragma unroll
        for (int t = 0; t < TG; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[g0 + t][i], acc[t], 0, 0, 0);
      }
      // accumulator register q: row (q & 3) + 8 (q >> 2) + 4 hf of the tile, column r
      if (accumulate) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ro = (q & 3) + 8 * (q >> 2);
          if (ro < rlim) {
#pragma unroll
            for (int t = 0; t < TG; ++t) cbase[ro * ldc + (g0 + t) * 32] += acc[t][q];
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ro = (q & 3) + 8 * (q >> 2);
          if (ro < rlim) {
#pragma unroll
            for (int t = 0; t < TG; ++t) cbase[ro * ldc + (g0 + t) * 32] = acc[t][q];
          }
        }
      }
    }
  }
}

// ---- weight-gradient kernel ----------------------------------------------------------------------
// partial[block][N1*N2 + N1]: the workgroup's dW (row-major N1 x N2) followed by its db
template <int T1, int T2>
__device__ __forceinline__ void gemm_wgrad_body(
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, int64_t M,
    float* __restrict__ partial, int chunks_per_wave, int blk, float* __restrict__ red) {
  constexpr int N1 = T1 * 32, N2 = T2 * 32;
  constexpr int PS = N1 * N2 + N1;                 // red: 2 * PS floats (two waves' partials at a time, <= 50 KB)
  const int lane = lane_id(), wave = wave_id_uniform();
  const int r = lane & 31, hf = lane >> 5;
  const int64_t wid = static_cast<int64_t>(blk) * kWaves + wave;
  const int64_t nchunks = (M + 31) / 32;
  f32x16 acc[T1][T2];
  float colsum[T1];
#pragma unroll
  for (int t1 = 0; t1 < T1; ++t1) {
    colsum[t1] = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < T2; ++t2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t1][t2][q] = 0.f;
    }
  }
  // (Requesting chunk it + 1 before chunk it is multiplied — two operand sets — was tried for <1, 2>: 32.5 -> 40.4 us;
  // the loads of one chunk already fill the wave's queue, and the second set costs a wave of occupancy.)
  for (int it = 0; it < chunks_per_wave; ++it) {
    const int64_t chunk = wid * chunks_per_wave + it;        // wave-uniform
    if (chunk >= nchunks) break;
    const int64_t m0 = chunk * 32 + hf;
    // k pair i of the chunk = rows m0 + 2 i (lane half 0) and m0 + 2 i + 1 (half 1); lane r = column
    float a[T1][16], bq[T2][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + 2 * i;
      const bool ok = m < M;
      const int64_t mc = ok ? m : M - 1;
#pragma unroll
      for (int t1 = 0; t1 < T1; ++t1) {
        const float v = G[mc * ldg + t1 * 32 + r];
        a[t1][i] = ok ? v : 0.f;
      }
#pragma unroll
      for (int t2 = 0; t2 < T2; ++t2) {
        const float v = X[mc * ldx + t2 * 32 + r];
        bq[t2][i] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int t1 = 0; t1 < T1; ++t1) {
        colsum[t1] += a[t1][i];
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2)
          acc[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t1][i], bq[t2][i], acc[t1][t2], 0, 0, 0);
      }
    }
  }
  // db: the two lane halves saw alternate rows
#pragma unroll
  for (int t1 = 0; t1 < T1; ++t1) colsum[t1] += __shfl_xor(colsum[t1], 32, kWave);
  // element (n1 = 32 t1 + row(q), n2 = 32 t2 + r) of this wave's partial dW
  auto put = [&](float* dst) {
#pragma unroll
    for (int t1 = 0; t1 < T1; ++t1) {
#pragma unroll
      for (int t2 = 0; t2 < T2; ++t2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n1 = t1 * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf;
          dst[n1 * N2 + t2 * 32 + r] = acc[t1][t2][q];
        }
      }
      if (hf == 0) dst[N1 * N2 + t1 * 32 + r] = colsum[t1];
    }
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int t1 = 0; t1 < T1; ++t1) {
#pragma unroll
      for (int t2 = 0; t2 < T2; ++t2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n1 = t1 * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf;
          acc[t1][t2][q] += src[n1 * N2 + t2 * 32 + r];
        }
      }
      colsum[t1] += src[N1 * N2 + t1 * 32 + r];
    }
  };
  // fixed order: (wave 0 + wave 2) + (wave 1 + wave 3)
  static_assert(kWaves == 4, "the reduction tree below is written for four waves");
  if (wave >= 2) put(red + (wave - 2) * PS);
  __syncthreads();
  if (wave < 2) add(red + wave * PS);
  __syncthreads();
  if (wave == 1) put(red);
  __syncthreads();
  if (wave == 0) {
    add(red);
    put(partial + static_cast<int64_t>(blk) * PS);
  }
}

template <int T1, int T2>
__global__ __launch_bounds__(kWaves * 64, 2) void gemm_wgrad_kernel(
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, int64_t M,
    float* __restrict__ partial, int chunks_per_wave) {
  __shared__ float red[2 * (T1 * 32 * T2 * 32 + T1 * 32)];
  gemm_wgrad_body<T1, T2>(G, ldg, X, ldx, M, partial, chunks_per_wave, blockIdx.x, red);
}

// Two weight gradients over the same rows in ONE launch (an attention block's dW_qkv and dW_out): workgroups
// [0, blocks) run the first, [blocks, 2 blocks) the second — one ramp and one tail instead of two.
template <int T1A, int T2A, int T1B, int T2B>
__global__ __launch_bounds__(kWaves * 64, 2) void gemm_wgrad_pair_kernel(
    const float* __restrict__ GA, int64_t ldgA, const float* __restrict__ XA, int64_t ldxA, float* __restrict__ partialA,
    const float* __restrict__ GB, int64_t ldgB, const float* __restrict__ XB, int64_t ldxB, float* __restrict__ partialB,
    int64_t M, int chunks_per_wave, int blocks) {
  constexpr int PSA = T1A * 32 * T2A * 32 + T1A * 32, PSB = T1B * 32 * T2B * 32 + T1B * 32;
  __shared__ float red[2 * (PSA > PSB ? PSA : PSB)];
  if (static_cast<int>(blockIdx.x) < blocks)
    gemm_wgrad_body<T1A, T2A>(GA, ldgA, XA, ldxA, M, partialA, chunks_per_wave, blockIdx.x, red);
  else
    gemm_wgrad_body<T1B, T2B>(GB, ldgB, XB, ldxB, M, partialB, chunks_per_wave, blockIdx.x - blocks, red);
}

// out[e] (+)= sum_blocks partial[block][e]  (fixed order); the last N1 entries go to db (partial_reduce.h).
__global__ __launch_bounds__(256) void gemm_wgrad_reduce(const float* __restrict__ partial, int blocks, int n1n2,
                                                         int n1, int N2, float* __restrict__ dW, int64_t ldw,
                                                         float* __restrict__ db, int accumulate) {
  __shared__ float part[4][64];
  partials::wgrad_reduce_body(blockIdx.x, partial, blocks, n1n2, n1, N2, dW, ldw, db, accumulate, part);
}

// several finishes in one launch: job j owns workgroups [first_block[j], first_block[j + 1])
constexpr int kMaxPartialJobs = 8;
struct PartialJobs {
  const float* partial[kMaxPartialJobs];
  float* out_w[kMaxPartialJobs];
  float* out_b[kMaxPartialJobs];
  int64_t ldw[kMaxPartialJobs];
  int kind[kMaxPartialJobs], blocks[kMaxPartialJobs], n1[kMaxPartialJobs], n2[kMaxPartialJobs], accumulate[kMaxPartialJobs];
  int first_block[kMaxPartialJobs + 1];
  int count;
};
__global__ __launch_bounds__(256) void partials_finish_kernel(PartialJobs jobs) {
  __shared__ float red[2][256];
  int j = 0;
  while (j + 1 < jobs.count && static_cast<int>(blockIdx.x) >= jobs.first_block[j + 1]) ++j;
  const int blk = blockIdx.x - jobs.first_block[j];
  if (jobs.kind[j] == 0)
    partials::wgrad_reduce_body(blk, jobs.partial[j], jobs.blocks[j], jobs.n1[j] * jobs.n2[j], jobs.n1[j], jobs.n2[j],
                                jobs.out_w[j], jobs.ldw[j], jobs.out_b[j], jobs.accumulate[j],
                                reinterpret_cast<float (*)[64]>(&red[0][0]));
  else
    partials::layernorm_finalize_body(blk, jobs.partial[j], jobs.blocks[j], jobs.n1[j], jobs.out_w[j], jobs.out_b[j], red);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int wgrad_blocks(int64_t M, int* chunks_per_wave) {
  const int64_t nchunks = (M + 31) / 32;
  const int64_t cpw = (nchunks + kWgradWaves - 1) / kWgradWaves;
  *chunks_per_wave = static_cast<int>(cpw);
  const int64_t waves = (nchunks + cpw - 1) / cpw;
  return static_cast<int>((waves + kWaves - 1) / kWaves);
}
}  // namespace

namespace dfm {
constexpr int64_t kSkinnyMinRows = 8192;

// C (+)= A W^T (+ bias) on the rows kernel; false if the shape is not one it takes
bool gemm_rows_try(const float* A, int64_t lda, const float* W, int64_t ldw, bool w_kc, float* C, int64_t ldc,
                   int64_t M, int N, int K, const float* bias, int accumulate, hipStream_t st) {
  if (M < kSkinnyMinRows || !aligned16(A) || lda % 4 != 0) return false;
  if (w_kc && (!aligned16(W) || ldw % 4 != 0)) return false;
  const int64_t ntiles = (M + 31) / 32;
  const int64_t tpw = (ntiles + kMaxResidentWaves - 1) / kMaxResidentWaves;
  const int64_t waves = (ntiles + tpw - 1) / tpw;
  const dim3 grid(static_cast<unsigned>((waves + kWaves - 1) / kWaves)), block(kWaves * 64);
#define DFM_ROWS(NT_, K_)                                                                                      \
  if (N == NT_ * 32 && K == K_) {                                                                              \
    hipLaunchKernelGGL((gemm_rows_kernel<NT_, K_>), grid, block, 0, st, A, lda, W, ldw, w_kc ? 1 : 0, C, ldc, M, \
                       bias, accumulate, static_cast<int>(tpw));                                               \
    return true;                                                                                               \
  }
  if (N == 192 && K == 32) {
    // six column tiles need > 256 VGPRs with the whole weight resident (one wave per SIMD, or spills):
    // two launches over three tiles each; A (the small operand of this shape) is read twice
    for (int half = 0; half < 2; ++half)
      hipLaunchKernelGGL((gemm_rows_kernel<3, 32>), grid, block, 0, st, A, lda, w_kc ? W + 96 * half * ldw : W + 96 * half,
                         ldw, w_kc ? 1 : 0, C + 96 * half, ldc, M, bias ? bias + 96 * half : nullptr, accumulate,
                         static_cast<int>(tpw));
    return true;
  }
  DFM_ROWS(1, 64) DFM_ROWS(1, 192) DFM_ROWS(2, 32)                       // Cfg4: out, d x, d o (qkv above)
  DFM_ROWS(3, 16) DFM_ROWS(1, 32) DFM_ROWS(1, 96) DFM_ROWS(3, 32) DFM_ROWS(2, 64) DFM_ROWS(1, 16)
#undef DFM_ROWS
  return false;
}

size_t gemm_wgrad_workspace_bytes(int64_t M, int N1, int N2) {
  int cpw;
  return sizeof(float) * static_cast<size_t>(wgrad_blocks(M, &cpw)) * (static_cast<size_t>(N1) * N2 + N1);
}

bool gemm_wgrad_supported(int64_t M, int N1, int N2) {
  if (M < kSkinnyMinRows) return false;
  const int t1 = N1 / 32, t2 = N2 / 32;
  if (N1 % 32 || N2 % 32) return false;
  return (t1 == 6 && t2 == 1) || (t1 == 1 && t2 == 2) || (t1 == 1 && t2 == 1) || (t1 == 2 && t2 == 1) ||
         (t1 == 3 && t2 == 1) || (t1 == 2 && t2 == 2) || (t1 == 1 && t2 == 3);
}

// dW (+)= G^T X, db (+)= column sums of G (db may be null); false if the shape is not taken
bool gemm_wgrad_try(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int N1, int N2,
                    float* dW, int64_t ldw, float* db, int accumulate, void* workspace, hipStream_t st, bool defer) {
  if (!workspace || !gemm_wgrad_supported(M, N1, N2)) return false;
  int cpw;
  const int blocks = wgrad_blocks(M, &cpw);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(blocks), block(kWaves * 64);
#define DFM_WG(T1_, T2_)                                                                                    \
  if (N1 == T1_ * 32 && N2 == T2_ * 32)                                                                     \
    hipLaunchKernelGGL((gemm_wgrad_kernel<T1_, T2_>), grid, block, 0, st, G, ldg, X, ldx, M, partial, cpw);
  DFM_WG(6, 1) DFM_WG(1, 2) DFM_WG(1, 1) DFM_WG(2, 1) DFM_WG(3, 1) DFM_WG(2, 2) DFM_WG(1, 3)
#undef DFM_WG
  if (defer) return true;               // the partials stay in the workspace: dfm_partials_finish
  const int total = N1 * N2 + N1;
  hipLaunchKernelGGL(gemm_wgrad_reduce, dim3((total + 63) / 64), dim3(256), 0, st, partial, blocks, N1 * N2, N1,
                     N2, dW, ldw, db, accumulate);
  return true;
}
// the (6, 1) + (1, 2) pair (attention_dim 64, embed_dim 32: dW_qkv and dW_out of configuration 4) in one launch
bool gemm_wgrad_pair_try(const float* GA, int64_t ldgA, const float* XA, int64_t ldxA, int n1A, int n2A, void* wsA,
                         const float* GB, int64_t ldgB, const float* XB, int64_t ldxB, int n1B, int n2B, void* wsB,
                         int64_t M, hipStream_t st) {
  if (!wsA || !wsB || !gemm_wgrad_supported(M, n1A, n2A) || !gemm_wgrad_supported(M, n1B, n2B)) return false;
  if (!(n1A == 192 && n2A == 32 && n1B == 32 && n2B == 64)) return false;
  int cpw;
  const int blocks = wgrad_blocks(M, &cpw);
  hipLaunchKernelGGL((gemm_wgrad_pair_kernel<6, 1, 1, 2>), dim3(2 * blocks), dim3(kWaves * 64), 0, st, GA, ldgA, XA, ldxA,
                     static_cast<float*>(wsA), GB, ldgB, XB, ldxB, static_cast<float*>(wsB), M, cpw, blocks);
  return true;
}
int gemm_wgrad_partial_blocks(int64_t M) {
  int cpw;
  return wgrad_blocks(M, &cpw);
}
}  // namespace dfm

extern "C" size_t dfm_weight_grad_workspace_bytes(int64_t rows, int n1, int n2) {
  return dfm::gemm_wgrad_supported(rows, n1, n2) ? dfm::gemm_wgrad_workspace_bytes(rows, n1, n2) : 0;
}

extern "C" int dfm_weight_grad_f32(const float* d_g, int64_t ldg, const float* d_x, int64_t ldx, int64_t rows,
                                   int n1, int n2, float* d_dw, int64_t lddw, float* d_db, int accumulate,
                                   void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_g && d_x && d_dw, "null argument");
  DFM_REQUIRE(rows > 0 && n1 > 0 && n2 > 0, "bad shape");
  if (!dfm::gemm_wgrad_try(d_g, ldg, d_x, ldx, rows, n1, n2, d_dw, lddw, d_db, accumulate, d_workspace,
                           as_stream(stream), false))
    return fail(DFM_ERR_UNSUPPORTED, "dfm_weight_grad_f32: no kernel for rows=%lld n1=%d n2=%d (workspace %p)",
                (long long)rows, n1, n2, d_workspace);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// The streamed pass of dfm_weight_grad_f32 alone: per-workgroup partials [blocks][n1 * n2 + n1] stay in the workspace
// for dfm_partials_finish (kind 0 with blocks = dfm_weight_grad_partial_blocks(rows)).
extern "C" int dfm_weight_grad_partial_blocks(int64_t rows) { return rows > 0 ? dfm::gemm_wgrad_partial_blocks(rows) : 0; }
extern "C" int dfm_weight_grad_partials_f32(const float* d_g, int64_t ldg, const float* d_x, int64_t ldx, int64_t rows,
                                            int n1, int n2, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_g && d_x && d_workspace, "null argument");
  DFM_REQUIRE(rows > 0 && n1 > 0 && n2 > 0, "bad shape");
  if (!dfm::gemm_wgrad_try(d_g, ldg, d_x, ldx, rows, n1, n2, nullptr, 0, nullptr, 0, d_workspace, as_stream(stream), true))
    return fail(DFM_ERR_UNSUPPORTED, "dfm_weight_grad_partials_f32: no kernel for rows=%lld n1=%d n2=%d",
                (long long)rows, n1, n2);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_partials_finish(const dfm_partial_job* jobs, int count, dfm_stream_t stream) {
  DFM_REQUIRE(jobs && count >= 1 && count <= kMaxPartialJobs, "1 to %d jobs", kMaxPartialJobs);
  PartialJobs pj = {};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const dfm_partial_job& j = jobs[i];
    DFM_REQUIRE(j.kind == 0 || j.kind == 1, "job %d: kind 0 (weight gradient) or 1 (LayerNorm)", i);
    DFM_REQUIRE(j.partial && j.out_w && j.blocks > 0 && j.n1 > 0, "job %d: incomplete", i);
    DFM_REQUIRE(j.kind == 0 ? (j.n2 > 0 && j.ldw >= j.n2) : (j.out_b != nullptr), "job %d: incomplete", i);
    pj.partial[i] = j.partial; pj.out_w[i] = j.out_w; pj.out_b[i] = j.out_b; pj.ldw[i] = j.ldw;
    pj.kind[i] = j.kind; pj.blocks[i] = j.blocks; pj.n1[i] = j.n1; pj.n2[i] = j.n2; pj.accumulate[i] = j.accumulate;
    pj.first_block[i] = blocks;
    blocks += j.kind == 0 ? (j.n1 * j.n2 + j.n1 + 63) / 64 : j.n1;
  }
  pj.first_block[count] = blocks;
  pj.count = count;
  hipLaunchKernelGGL(partials_finish_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), pj);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// Two dfm_weight_grad_partials_f32 passes over the same rows in one launch; DFM_ERR_UNSUPPORTED for a pair of shapes
// without a joint kernel (the caller then makes the two calls).
extern "C" int dfm_weight_grad_partials_pair_f32(const float* d_g_a, int64_t ldg_a, const float* d_x_a, int64_t ldx_a,
                                                 int n1_a, int n2_a, void* d_workspace_a, const float* d_g_b,
                                                 int64_t ldg_b, const float* d_x_b, int64_t ldx_b, int n1_b, int n2_b,
                                                 void* d_workspace_b, int64_t rows, dfm_stream_t stream) {
  DFM_REQUIRE(d_g_a && d_x_a && d_workspace_a && d_g_b && d_x_b && d_workspace_b, "null argument");
  DFM_REQUIRE(rows > 0, "bad shape");
  if (!dfm::gemm_wgrad_pair_try(d_g_a, ldg_a, d_x_a, ldx_a, n1_a, n2_a, d_workspace_a, d_g_b, ldg_b, d_x_b, ldx_b, n1_b,
                                n2_b, d_workspace_b, rows, as_stream(stream)))
    return fail(DFM_ERR_UNSUPPORTED, "dfm_weight_grad_partials_pair_f32: no joint kernel for (%d, %d) + (%d, %d)", n1_a,
                n2_a, n1_b, n2_b);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
