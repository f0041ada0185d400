// Fused DNN tower + head for the training step (reference deepfm/models/layers/dnn.py:45-55:
// [Linear -> BatchNorm1d -> ReLU -> Dropout] * n; deepfm.py:30-42: first_order + fm + Linear(dnn);
// trainer.py:59,221: BCEWithLogitsLoss).
//
// On an MI355X a dependent kernel costs ~4.5 us whatever it does, and the tower's GEMMs are
// 5-30 us each, so the tower is organised around FEW launches, not around FLOPs:
//
//   forward, per layer   linear_bn_fwd        z = x W^T + b on the exact-fp32 MFMA tile loop
//                                             (gemm_core.h); the epilogue reduces per-column
//                                             (mean, M2) over each 32-row MFMA tile, and the LAST
//                                             workgroup of a column tile (device-scope counter)
//                                             merges them in a fixed order (Chan) -> mean, rstd,
//                                             running statistics.  No separate statistics pass.
//                        bn_relu_dropout_apply a = dropout(relu(gamma*(z-mean)*rstd + beta))
//   head                 head_bce             logit = (fo + fm) + (a.w + b); BCE loss; d logit;
//                                             d w, d b; g = d logit * w pushed through the last
//                                             BatchNorm's ReLU/dropout mask (dy) with its column
//                                             sums (same last-workgroup reduction)
//   backward, per layer  bn_bwd_apply         dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat))
//                        linear_bwd           ONE launch for both GEMMs of a Linear backward:
//                                             dW += dz^T x (batch split into slabs, the last
//                                             workgroup of an output tile adds the slabs in order)
//                                             and dx = dz W, whose epilogue is either the NEXT
//                                             (lower) layer's BatchNorm mask + column sums, or the
//                                             FM backward g_fm*(S - e) added in place (layer 1).
//
// Every reduction has a fixed association (no floating-point read-modify-write atomics): results
// are bitwise reproducible run to run.  Counters are self-cleaning (the last workgroup resets them).
#include "dropout.h"
#include "gemm_core.h"

using namespace dfm;
using namespace dfm::gemm;

namespace {

constexpr int kFinLanes = kThreads / BN;   // 8 partial-lanes per column in a last-workgroup reduction

// Cross-workgroup hand-off without cache-wide fences.  gfx950 has one L2 per XCD and they are not
// coherent with each other: an agent-scope fence (__threadfence) writes back and invalidates the
// whole L2 of the issuing XCD, and doing that once per workgroup destroys the operand reuse of
// every GEMM tile running beside it (measured: 3-6x slower kernels).  Instead, exactly the values
// that cross workgroups (per-tile partial sums, dW slabs, the arrival counters) are written and
// read with agent-scope relaxed atomics — sc1 stores write through to memory, sc1 loads bypass the
// non-coherent cache levels — and ordering comes from "all my stores have completed" (s_waitcnt)
// + workgroup barrier + the counter increment by one thread.
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// "Am I the last workgroup of my group to get here?"  Every thread of the workgroup calls it
// after its st_agent writes; a true return means all the group's st_agent writes are in memory.
__device__ __forceinline__ bool last_block_of(int* counter, int group_size) {
  __shared__ int s_last;
  __builtin_amdgcn_s_waitcnt(0);     // this wave's stores have completed (vmcnt/lgkmcnt 0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == group_size - 1;
    if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// Sum of `mine` over the kFinLanes partial-lanes of column c (fixed order), returned to all of them.
__device__ __forceinline__ float column_total(float mine, float* red, int c, int pl) {
  red[pl * BN + c] = mine;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < kFinLanes; ++i) tot += red[i * BN + c];
  __syncthreads();
  return tot;
}

struct BnBwd {            // device view of dfm_bn_bwd
  const float* z;
  const float* mean_rstd;
  const float* gamma;
  const float* beta;
  float* dy;
  float* means;
  float* g_gamma;
  float* g_beta;
  const int64_t* seed;
  float* partial;         // [T][2][N]
  int* counters;          // [tiles_n]
  uint32_t thresh;
  float inv_keep;
  int salt;
};

// Epilogue shared by head_bce and linear_bwd: the workgroup's writer waves hold g(m, n) for a
// 32 x 32 tile; push it through the BatchNorm's ReLU/dropout mask, store dy, and leave per-tile
// column sums of dy and dy*xhat in bn.partial.  N = features of the BatchNorm layer.
__device__ __forceinline__ void bn_mask_tile(const BnBwd& bn, const f32x16& g, const TilePos& pos, int m0, int n0,
                                             int M, int N) {
  const int n = n0 + pos.col();
  const bool okn = n < N;
  const int nc = okn ? n : 0;
  const float mu = bn.mean_rstd[nc], rs = bn.mean_rstd[N + nc], ga = bn.gamma[nc], be = bn.beta[nc];
  const int64_t seed = bn.seed ? bn.seed[0] : 0;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + pos.row(reg);
    if (m < M && okn) {
      const int64_t idx = static_cast<int64_t>(m) * N + n;
      const float zh = (bn.z[idx] - mu) * rs;
      const float y = fmaf(ga, zh, be);
      const float dy = y > 0.f ? g[reg] * drop_scale(seed, bn.salt, idx, bn.thresh, bn.inv_keep) : 0.f;
      bn.dy[idx] = dy;
      s1 += dy;
      s2 = fmaf(dy, zh, s2);
    }
  }
  s1 += __shfl_xor(s1, 32, kWave);
  s2 += __shfl_xor(s2, 32, kWave);
  if (pos.hf == 0 && okn && m0 + pos.wm < M) {
    const int t = (m0 + pos.wm) / 32;
    st_agent(bn.partial + (static_cast<int64_t>(t) * 2 + 0) * N + n, s1);
    st_agent(bn.partial + (static_cast<int64_t>(t) * 2 + 1) * N + n, s2);
  }
}

// Last workgroup of a column tile: means, d gamma, d beta from the per-tile sums (fixed order).
__device__ __forceinline__ void bn_mask_finalize(const BnBwd& bn, int n0, int M, int N, float* red) {
  const int T = (M + 31) / 32;
  const int c = threadIdx.x & (BN - 1), pl = threadIdx.x / BN;
  const int n = n0 + c;
  const bool ok = n < N;
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
    for (int t = pl; t < T; t += kFinLanes) {
      s1 += ld_agent(bn.partial + (static_cast<int64_t>(t) * 2 + 0) * N + n);
      s2 += ld_agent(bn.partial + (static_cast<int64_t>(t) * 2 + 1) * N + n);
    }
  }
  s1 = column_total(s1, red, c, pl);
  s2 = column_total(s2, red, c, pl);
  if (pl == 0 && ok) {
    bn.means[n] = s1 / static_cast<float>(M);
    bn.means[N + n] = s2 / static_cast<float>(M);
    bn.g_beta[n] += s1;
    bn.g_gamma[n] += s2;
  }
}

}  // namespace

// =====================================================================================
// forward: z = x W^T + b with BatchNorm batch statistics.  grid tiles_m * tiles_n (n fastest)
// =====================================================================================
template <bool FAST>
__global__ __launch_bounds__(kThreads) void linear_bn_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ z, int M, int N, int K, int tiles_n, float* __restrict__ partial,
    int* __restrict__ counters, float* __restrict__ mean_rstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ num_batches, float momentum, float eps) {
  __shared__ Smem sm;
  const TilePos pos;
  const int lt = xcd_logical_index(blockIdx.x, gridDim.x);
  const int tn = lt % tiles_n, tiles_m = gridDim.x / tiles_n;
  const int m0 = (lt / tiles_n) * BM, n0 = tn * BN;
  f32x16 acc = {};
  mainloop<true, true, FAST, FAST>(x, ldx, w, K, M, N, m0, n0, 0, K, sm, pos, acc);
  if (pos.khalf == 0) {
    const int n = n0 + pos.col();
    const bool okn = n < N;
    const float bv = (bias && okn) ? bias[n] : 0.f;
    const int cnt_i = M - (m0 + pos.wm) < 32 ? M - (m0 + pos.wm) : 32;   // valid rows of this MFMA tile
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + pos.row(reg);
      v[reg] = acc[reg] + bv;
      if (m < M && okn) {
        z[static_cast<int64_t>(m) * N + n] = v[reg];
        s += v[reg];
      }
    }
    s += __shfl_xor(s, 32, kWave);
    const float mean_t = cnt_i > 0 ? s / static_cast<float>(cnt_i) : 0.f;
    float q = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + pos.row(reg);
      const float d = v[reg] - mean_t;
      if (m < M) q = fmaf(d, d, q);
    }
    q += __shfl_xor(q, 32, kWave);
    if (pos.hf == 0 && okn && cnt_i > 0) {
      const int t = (m0 + pos.wm) / 32;
      st_agent(partial + (static_cast<int64_t>(t) * 2 + 0) * N + n, mean_t);
      st_agent(partial + (static_cast<int64_t>(t) * 2 + 1) * N + n, q);
    }
  }
  if (!last_block_of(counters + tn, tiles_m)) return;
  // ---- merge the tiles' (count, mean, M2) of this column tile: mean first, then M2 about it ----
  float* red = &sm.a[0][0];
  const int T = (M + 31) / 32;
  const int c = threadIdx.x & (BN - 1), pl = threadIdx.x / BN;
  const int n = n0 + c;
  const bool ok = n < N;
  float s = 0.f;
  if (ok) {
    for (int t = pl; t < T; t += kFinLanes) {
      const int cnt = M - 32 * t < 32 ? M - 32 * t : 32;
      s = fmaf(static_cast<float>(cnt), ld_agent(partial + (static_cast<int64_t>(t) * 2) * N + n), s);
    }
  }
  const float mu = column_total(s, red, c, pl) / static_cast<float>(M);
  float q = 0.f;
  if (ok) {
    for (int t = pl; t < T; t += kFinLanes) {
      const int cnt = M - 32 * t < 32 ? M - 32 * t : 32;
      const float d = ld_agent(partial + (static_cast<int64_t>(t) * 2) * N + n) - mu;
      q += fmaf(static_cast<float>(cnt) * d, d, ld_agent(partial + (static_cast<int64_t>(t) * 2 + 1) * N + n));
    }
  }
  const float var = column_total(q, red, c, pl) / static_cast<float>(M);   // biased, as BN normalises
  if (pl == 0 && ok) {
    mean_rstd[n] = mu;
    mean_rstd[N + n] = rsqrtf(var + eps);
    if (running_mean) {
      const float unbiased = M > 1 ? var * static_cast<float>(M) / static_cast<float>(M - 1) : var;
      running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mu;
      running_var[n] = (1.f - momentum) * running_var[n] + momentum * unbiased;
    }
  }
  if (tn == 0 && threadIdx.x == 0 && num_batches) num_batches[0] += 1;
}

// a = dropout(relu(gamma * (z - mean) * rstd + beta)), 4 elements per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void bn_relu_dropout_apply_kernel(
    const float* __restrict__ z, int64_t total4, int N, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, uint32_t thresh, float inv_keep,
    const int64_t* __restrict__ seed_ptr, int salt, float* __restrict__ out) {
  const int64_t i4 = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const int64_t idx = i4 * 4;
  const int64_t seed = seed_ptr ? seed_ptr[0] : 0;
  const int c = static_cast<int>(idx % N);
  const float4 zv = ld4(z + idx), mu = ld4(mean_rstd + c), rs = ld4(mean_rstd + N + c), ga = ld4(gamma + c),
               be = ld4(beta + c);
  float4 o;
  o.x = fmaxf(fmaf(ga.x, (zv.x - mu.x) * rs.x, be.x), 0.f) * drop_scale(seed, salt, idx + 0, thresh, inv_keep);
  o.y = fmaxf(fmaf(ga.y, (zv.y - mu.y) * rs.y, be.y), 0.f) * drop_scale(seed, salt, idx + 1, thresh, inv_keep);
  o.z = fmaxf(fmaf(ga.z, (zv.z - mu.z) * rs.z, be.z), 0.f) * drop_scale(seed, salt, idx + 2, thresh, inv_keep);
  o.w = fmaxf(fmaf(ga.w, (zv.w - mu.w) * rs.w, be.w), 0.f) * drop_scale(seed, salt, idx + 3, thresh, inv_keep);
  st4(out + idx, o);
}

// dz = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)); dz may alias dy
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ z, int64_t total4, int N,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ means,
    float* __restrict__ dz) {
  const int64_t i4 = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const int64_t idx = i4 * 4;
  const int c = static_cast<int>(idx % N);
  const float4 d = ld4(dy + idx), zv = ld4(z + idx), mu = ld4(mean_rstd + c), rs = ld4(mean_rstd + N + c),
               ga = ld4(gamma + c), m1 = ld4(means + c), m2 = ld4(means + N + c);
  float4 o;
  o.x = ga.x * rs.x * (d.x - m1.x - (zv.x - mu.x) * rs.x * m2.x);
  o.y = ga.y * rs.y * (d.y - m1.y - (zv.y - mu.y) * rs.y * m2.y);
  o.z = ga.z * rs.z * (d.z - m1.z - (zv.z - mu.z) * rs.z * m2.z);
  o.w = ga.w * rs.w * (d.w - m1.w - (zv.w - mu.w) * rs.w * m2.w);
  st4(dz + idx, o);
}

// =====================================================================================
// head: logits, BCE, d logit, d w, d b and the last BatchNorm's masked gradient
// =====================================================================================
namespace {
constexpr int kHeadThreads = 256;
constexpr int kHeadLPR = 8;                              // lanes per row, one float4 per 32 features
constexpr int kHeadRows = kHeadThreads / kHeadLPR;       // 32 rows per workgroup
constexpr int kHeadMaxChunks = 8;                        // K <= 256
}  // namespace

// partials per workgroup: [3][K] column sums (dy, dy*xhat, dlogit*a) + [2] (loss, dlogit)
template <int CH>   // CH = K / 32 float4 chunks per lane
__global__ __launch_bounds__(kHeadThreads) void head_bce_kernel(
    const float* __restrict__ a, int M, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ fo, const float* __restrict__ fm, const float* __restrict__ labels,
    float* __restrict__ logits, float* __restrict__ loss, float* __restrict__ dlogit, float* __restrict__ g_w,
    float* __restrict__ g_b, float* __restrict__ g_a, BnBwd bn, int has_bn, float* __restrict__ hpart,
    int* __restrict__ counter) {
  constexpr int K = CH * 32;
  constexpr int P = 3 * K + 2;
  __shared__ float red[kHeadThreads / kWave][P];
  const int tid = threadIdx.x, l8 = tid & (kHeadLPR - 1), rl = tid / kHeadLPR;
  const int m = blockIdx.x * kHeadRows + rl;
  const bool live = m < M;
  const int mc = live ? m : M - 1;
  float4 av[CH], wv[CH];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int j = i * 32 + l8 * 4;
    av[i] = ld4(a + static_cast<int64_t>(mc) * K + j);
    wv[i] = ld4(w + j);
    dot = fmaf(av[i].x, wv[i].x, dot); dot = fmaf(av[i].y, wv[i].y, dot);
    dot = fmaf(av[i].z, wv[i].z, dot); dot = fmaf(av[i].w, wv[i].w, dot);
  }
  dot += __shfl_xor(dot, 1, kWave);
  dot += __shfl_xor(dot, 2, kWave);
  dot += __shfl_xor(dot, 4, kWave);
  // (first_order + fm) + (dnn . w + b): the association of deepfm.py:30-42
  const float zl = ((fo ? fo[mc] : 0.f) + (fm ? fm[mc] : 0.f)) + (dot + (b ? b[0] : 0.f));
  const float yl = labels[mc];
  const float e = expf(-fabsf(zl));
  const float li = fmaxf(zl, 0.f) - zl * yl + log1pf(e);
  const float sig = zl >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
  const float dl = live ? (sig - yl) / static_cast<float>(M) : 0.f;
  if (live && l8 == 0) {
    logits[m] = zl;
    dlogit[m] = dl;
  }
  float cs[3][CH][4];            // this lane's contributions to the column sums
  const int64_t seed = (has_bn && bn.seed) ? bn.seed[0] : 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int j = i * 32 + l8 * 4;
    const float gv[4] = {dl * wv[i].x, dl * wv[i].y, dl * wv[i].z, dl * wv[i].w};
    const float aa[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
    float dyv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bn) {
      const float4 zv = ld4(bn.z + static_cast<int64_t>(mc) * K + j), mu = ld4(bn.mean_rstd + j),
                   rs = ld4(bn.mean_rstd + K + j), ga = ld4(bn.gamma + j), be = ld4(bn.beta + j);
      const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, mm[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w},
                  gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float zh = (zz[u] - mm[u]) * rr[u];
        const float y = fmaf(gg[u], zh, bb[u]);
        const int64_t idx = static_cast<int64_t>(mc) * K + j + u;
        dyv[u] = (live && y > 0.f) ? gv[u] * drop_scale(seed, bn.salt, idx, bn.thresh, bn.inv_keep) : 0.f;
        cs[0][i][u] = dyv[u];
        cs[1][i][u] = dyv[u] * zh;
      }
      if (live) st4(bn.dy + static_cast<int64_t>(m) * K + j, make_float4(dyv[0], dyv[1], dyv[2], dyv[3]));
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) cs[0][i][u] = cs[1][i][u] = 0.f;
      if (live && g_a) st4(g_a + static_cast<int64_t>(m) * K + j, make_float4(gv[0], gv[1], gv[2], gv[3]));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) cs[2][i][u] = dl * aa[u];
  }
  // rows of a wave (8 rows x 8 lanes): butterfly over the row bits, then the 4 waves through LDS
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v = cs[p][i][u];
        v += __shfl_xor(v, 8, kWave);
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        if (lane < kHeadLPR) red[wave][p * K + i * 32 + lane * 4 + u] = v;
      }
  float sl = (live && l8 == 0) ? li : 0.f, sd = (l8 == 0) ? dl : 0.f;
#pragma unroll
  for (int msk = 1; msk < kWave; msk <<= 1) {
    sl += __shfl_xor(sl, msk, kWave);
    sd += __shfl_xor(sd, msk, kWave);
  }
  if (lane == 0) { red[wave][3 * K] = sl; red[wave][3 * K + 1] = sd; }
  __syncthreads();
  for (int o = tid; o < P; o += kHeadThreads)
    st_agent(hpart + static_cast<int64_t>(blockIdx.x) * P + o, (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]));
  if (!last_block_of(counter, gridDim.x)) return;
  const int nb = gridDim.x;
  for (int o = tid; o < P; o += kHeadThreads) {
    // fixed order; loads issued 8 at a time so they overlap instead of chaining
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= nb; i += 8) {
      float tq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) tq[u] = ld_agent(hpart + static_cast<int64_t>(i + u) * P + o);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += tq[u];
    }
    for (; i < nb; ++i) s += ld_agent(hpart + static_cast<int64_t>(i) * P + o);
    if (o < K) {
      if (has_bn) { bn.means[o] = s / static_cast<float>(M); bn.g_beta[o] += s; }
    } else if (o < 2 * K) {
      if (has_bn) { bn.means[K + (o - K)] = s / static_cast<float>(M); bn.g_gamma[o - K] += s; }
    } else if (o < 3 * K) {
      g_w[o - 2 * K] += s;
    } else if (o == 3 * K) {
      loss[0] = s / static_cast<float>(M);
    } else {
      if (g_b) g_b[0] += s;
    }
  }
}

// =====================================================================================
// backward of one Linear: dW += dz^T x  and  dx = dz W (+ epilogue), one launch.
//   blocks [0, dw_blocks): dW tiles x splits;  the rest: dx tiles
// =====================================================================================
struct FmBwd {
  const float* g_fm;     // (M)      d loss / d fm value
  const float* fm_sum;   // (M, D)   sum_f e
  const float* e;        // (M, K)   field embeddings
  int dim;
};

template <bool FAST, int EPI>   // EPI 0: plain store, 1: BatchNorm mask of the lower layer, 2: + FM backward
__global__ __launch_bounds__(kThreads) void linear_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ g_w, float* __restrict__ g_x, int M, int N, int K, int dw_tiles_n, int dw_tiles, int splits,
    int k_per_split, float* __restrict__ slabs, int* __restrict__ dw_counters, int dx_tiles_n, BnBwd bn, FmBwd fmb) {
  __shared__ Smem sm;
  const TilePos pos;
  f32x16 acc = {};
  const int bid = xcd_logical_index(blockIdx.x, gridDim.x);
  if (bid < dw_tiles * splits) {
    // ---- dW (N x K) = sum over the batch: A = dz (k-strided), B = x (k-strided) ----
    const int tile = bid % dw_tiles, sp = bid / dw_tiles;
    const int m0 = (tile / dw_tiles_n) * BM, n0 = (tile % dw_tiles_n) * BN;
    const int kb = sp * k_per_split;
    const int ke = kb + k_per_split < M ? kb + k_per_split : M;
    mainloop<false, false, FAST, FAST>(dz, N, x, K, N, K, m0, n0, kb, ke, sm, pos, acc);
    const int n = n0 + pos.col();
    if (pos.khalf == 0 && n < K) {
      float* sl = slabs + static_cast<int64_t>(sp) * N * K;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + pos.row(reg);
        if (m < N) st_agent(sl + static_cast<int64_t>(m) * K + n, acc[reg]);
      }
    }
    if (!last_block_of(dw_counters + tile, splits)) return;
    // the last split of this tile adds the slabs in order: 64 x 64 elements, 8 per thread
    for (int i = threadIdx.x; i < BM * BN; i += kThreads) {
      const int m = m0 + i / BN, nn = n0 + i % BN;
      if (m < N && nn < K) {
        const int64_t off = static_cast<int64_t>(m) * K + nn;
        float s = 0.f;
        int q = 0;
        for (; q + 4 <= splits; q += 4) {
          float tq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) tq[u] = ld_agent(slabs + static_cast<int64_t>(q + u) * N * K + off);
#pragma unroll
          for (int u = 0; u < 4; ++u) s += tq[u];
        }
        for (; q < splits; ++q) s += ld_agent(slabs + static_cast<int64_t>(q) * N * K + off);
        g_w[off] += s;
      }
    }
    return;
  }
  // ---- dx (M x K) = dz W: A = dz (k-contiguous), B = W (k-strided) ----
  const int t = bid - dw_tiles * splits;
  const int m0 = (t / dx_tiles_n) * BM, n0 = (t % dx_tiles_n) * BN;
  mainloop<true, false, FAST, FAST>(dz, N, w, K, M, K, m0, n0, 0, N, sm, pos, acc);
  if (EPI == 1) {
    if (pos.khalf == 0) bn_mask_tile(bn, acc, pos, m0, n0, M, K);
    const int tiles_m = (M + BM - 1) / BM;
    if (!last_block_of(bn.counters + (t % dx_tiles_n), tiles_m)) return;
    bn_mask_finalize(bn, n0, M, K, &sm.a[0][0]);
    return;
  }
  if (pos.khalf == 1) return;
  const int n = n0 + pos.col();
  if (n >= K) return;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + pos.row(reg);
    if (m < M) {
      const int64_t off = static_cast<int64_t>(m) * K + n;
      float v = acc[reg];
      if (EPI == 2)   // d e = d flat + g_fm * (S - e)   (fm.py:18-23 backward)
        v += fmb.g_fm[m] * (fmb.fm_sum[static_cast<int64_t>(m) * fmb.dim + n % fmb.dim] - fmb.e[off]);
      g_x[off] = v;
    }
  }
}

// =====================================================================================
// host side
// =====================================================================================
namespace {
inline int tiles(int n, int t) { return (n + t - 1) / t; }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

// workspace of a BatchNorm column reduction over (m, n): partial [T][2][n] floats, then counters
inline size_t bn_partial_bytes(int64_t m, int n) { return align256(sizeof(float) * 2 * static_cast<size_t>((m + 31) / 32) * n); }
inline size_t bn_counter_bytes(int n) { return align256(sizeof(int) * static_cast<size_t>(tiles(n, BN))); }

int dw_splits(int n_out, int k_in, int64_t m) {
  const int64_t t = static_cast<int64_t>(tiles(n_out, BM)) * tiles(k_in, BN);
  int64_t s = (512 + t - 1) / t;                  // aim for ~2 workgroups per CU from the dW part
  const int64_t max_s = m / (4 * BK) > 0 ? m / (4 * BK) : 1;   // at least 4 k-slices per split
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : static_cast<int>(s);
}

bool fill_bn(const dfm_bn_bwd* h, int64_t m, int n, BnBwd* d) {
  if (!h->z || !h->mean_rstd || !h->gamma || !h->beta || !h->dy || !h->means || !h->g_gamma || !h->g_beta ||
      !h->workspace)
    return false;
  if (!(h->p_drop >= 0.f && h->p_drop < 1.f) || (h->p_drop > 0.f && !h->seed)) return false;
  d->z = h->z; d->mean_rstd = h->mean_rstd; d->gamma = h->gamma; d->beta = h->beta;
  d->dy = h->dy; d->means = h->means; d->g_gamma = h->g_gamma; d->g_beta = h->g_beta;
  d->seed = h->seed;
  d->partial = static_cast<float*>(h->workspace);
  d->counters = reinterpret_cast<int*>(static_cast<char*>(h->workspace) + bn_partial_bytes(m, n));
  d->thresh = dropout_thresh(h->p_drop);
  d->inv_keep = 1.f / (1.f - h->p_drop);
  d->salt = h->salt;
  return true;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

extern "C" size_t dfm_bn_bwd_workspace_bytes(int64_t batch, int features) {
  return bn_partial_bytes(batch, features) + bn_counter_bytes(features);
}

extern "C" size_t dfm_linear_bn_workspace_bytes(int64_t batch, int features) {
  return bn_partial_bytes(batch, features) + bn_counter_bytes(features);
}

extern "C" int dfm_linear_bn_forward(const float* d_x, int64_t ldx, const float* d_w, const float* d_bias,
                                     int64_t batch, int out_features, int in_features, float* d_z,
                                     float* d_mean_rstd, float* d_running_mean, float* d_running_var,
                                     int64_t* d_num_batches, float momentum, float eps, void* d_workspace,
                                     dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w && d_z && d_mean_rstd && d_workspace, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && out_features > 0 && in_features > 0, "bad shape");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  float* partial = static_cast<float*>(d_workspace);
  int* counters = reinterpret_cast<int*>(static_cast<char*>(d_workspace) + bn_partial_bytes(M, N));
  const int tn = tiles(N, BN);
  const dim3 grid(static_cast<unsigned>(tn) * tiles(M, BM));
  const bool fast = operand_fast(d_x, ldx, true, M, K) && operand_fast(d_w, K, true, N, K);
  if (fast)
    hipLaunchKernelGGL(linear_bn_fwd_kernel<true>, grid, dim3(kThreads), 0, as_stream(stream), d_x, ldx, d_w, d_bias,
                       d_z, M, N, K, tn, partial, counters, d_mean_rstd, d_running_mean, d_running_var,
                       d_num_batches, momentum, eps);
  else
    hipLaunchKernelGGL(linear_bn_fwd_kernel<false>, grid, dim3(kThreads), 0, as_stream(stream), d_x, ldx, d_w, d_bias,
                       d_z, M, N, K, tn, partial, counters, d_mean_rstd, d_running_mean, d_running_var,
                       d_num_batches, momentum, eps);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_bn_relu_dropout_apply(const float* d_z, int64_t batch, int features, const float* d_mean_rstd,
                                         const float* d_gamma, const float* d_beta, float p_drop,
                                         const int64_t* d_seed, int salt, float* d_out, dfm_stream_t stream) {
  DFM_REQUIRE(d_z && d_mean_rstd && d_gamma && d_beta && d_out, "null argument");
  DFM_REQUIRE(batch > 0 && features > 0 && features % 4 == 0, "features must be a positive multiple of 4");
  DFM_REQUIRE(aligned16(d_z) && aligned16(d_mean_rstd) && aligned16(d_gamma) && aligned16(d_beta) && aligned16(d_out),
              "pointers must be 16-byte aligned");
  DFM_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || d_seed), "bad dropout arguments");
  const int64_t total4 = batch * features / 4;
  hipLaunchKernelGGL(bn_relu_dropout_apply_kernel, dim3(static_cast<unsigned>((total4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_z, total4, features, d_mean_rstd, d_gamma, d_beta, dropout_thresh(p_drop),
                     1.f / (1.f - p_drop), d_seed, salt, d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_bn_backward_apply(const float* d_dy, const float* d_z, const float* d_mean_rstd,
                                     const float* d_gamma, const float* d_means, int64_t batch, int features,
                                     float* d_dz, dfm_stream_t stream) {
  DFM_REQUIRE(d_dy && d_z && d_mean_rstd && d_gamma && d_means && d_dz, "null argument");
  DFM_REQUIRE(batch > 0 && features > 0 && features % 4 == 0, "features must be a positive multiple of 4");
  DFM_REQUIRE(aligned16(d_dy) && aligned16(d_z) && aligned16(d_mean_rstd) && aligned16(d_gamma) &&
                  aligned16(d_means) && aligned16(d_dz), "pointers must be 16-byte aligned");
  const int64_t total4 = batch * features / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(static_cast<unsigned>((total4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_dy, d_z, total4, features, d_mean_rstd, d_gamma, d_means, d_dz);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" size_t dfm_head_bce_workspace_bytes(int64_t batch, int features) {
  const size_t blocks = static_cast<size_t>((batch + kHeadRows - 1) / kHeadRows);
  return align256(sizeof(float) * blocks * (3 * static_cast<size_t>(features) + 2)) + 256;
}

extern "C" int dfm_head_bce(const float* d_a, int64_t batch, int features, const float* d_w, const float* d_b,
                            const float* d_first_order, const float* d_fm, const float* d_labels, float* d_logits,
                            float* d_loss, float* d_g_logits, float* d_g_w, float* d_g_b, float* d_g_a,
                            const dfm_bn_bwd* bn, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_a && d_w && d_labels && d_logits && d_loss && d_g_logits && d_g_w && d_workspace, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30), "bad batch");
  DFM_REQUIRE(features > 0 && features % 32 == 0 && features <= 32 * kHeadMaxChunks,
              "head features must be a multiple of 32, at most 256");
  DFM_REQUIRE(aligned16(d_a) && aligned16(d_w), "pointers must be 16-byte aligned");
  BnBwd dbn = {};
  if (bn) DFM_REQUIRE(fill_bn(bn, batch, features, &dbn), "incomplete dfm_bn_bwd");
  const int M = static_cast<int>(batch);
  const unsigned blocks = static_cast<unsigned>((batch + kHeadRows - 1) / kHeadRows);
  float* hpart = static_cast<float*>(d_workspace);
  int* counter = reinterpret_cast<int*>(static_cast<char*>(d_workspace) +
                                        align256(sizeof(float) * blocks * (3 * static_cast<size_t>(features) + 2)));
#define DFM_HEAD(CH)                                                                                              \
  hipLaunchKernelGGL(head_bce_kernel<CH>, dim3(blocks), dim3(kHeadThreads), 0, as_stream(stream), d_a, M, d_w, d_b, \
                     d_first_order, d_fm, d_labels, d_logits, d_loss, d_g_logits, d_g_w, d_g_b, d_g_a, dbn,       \
                     bn ? 1 : 0, hpart, counter)
  switch (features / 32) {
    case 1: DFM_HEAD(1); break;
    case 2: DFM_HEAD(2); break;
    case 3: DFM_HEAD(3); break;
    case 4: DFM_HEAD(4); break;
    case 5: DFM_HEAD(5); break;
    case 6: DFM_HEAD(6); break;
    case 7: DFM_HEAD(7); break;
    default: DFM_HEAD(8); break;
  }
#undef DFM_HEAD
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" size_t dfm_linear_backward_workspace_bytes(int64_t batch, int out_features, int in_features) {
  const int s = dw_splits(out_features, in_features, batch);
  return align256(sizeof(float) * static_cast<size_t>(s) * out_features * in_features) +
         align256(sizeof(int) * static_cast<size_t>(tiles(out_features, BM)) * tiles(in_features, BN));
}

extern "C" int dfm_linear_backward(const float* d_dz, int64_t batch, int out_features, const float* d_x,
                                   int in_features, const float* d_w, float* d_g_w, float* d_g_x,
                                   const dfm_bn_bwd* bn_below, const dfm_fm_bwd* fm, void* d_workspace,
                                   dfm_stream_t stream) {
  DFM_REQUIRE(d_dz && d_x && d_w && d_g_w && d_workspace, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && out_features > 0 && in_features > 0, "bad shape");
  DFM_REQUIRE(!(bn_below && fm), "bn_below and fm are exclusive");
  DFM_REQUIRE(bn_below || d_g_x, "d_g_x is required without bn_below");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  BnBwd dbn = {};
  if (bn_below) DFM_REQUIRE(fill_bn(bn_below, batch, K, &dbn), "incomplete dfm_bn_bwd");
  FmBwd dfm_ = {};
  if (fm) {
    DFM_REQUIRE(fm->g_fm && fm->fm_sum && fm->e && fm->dim > 0 && K % fm->dim == 0, "incomplete dfm_fm_bwd");
    dfm_.g_fm = fm->g_fm; dfm_.fm_sum = fm->fm_sum; dfm_.e = fm->e; dfm_.dim = fm->dim;
  }
  const int splits0 = dw_splits(N, K, M);
  const int k_per_split = ((M + splits0 - 1) / splits0 + BK - 1) / BK * BK;
  const int splits = (M + k_per_split - 1) / k_per_split;
  const int dw_tn = tiles(K, BN), dw_t = tiles(N, BM) * dw_tn;
  const int dx_tn = tiles(K, BN), dx_t = tiles(M, BM) * dx_tn;
  float* slabs = static_cast<float*>(d_workspace);
  int* dw_counters = reinterpret_cast<int*>(static_cast<char*>(d_workspace) +
                                            align256(sizeof(float) * static_cast<size_t>(splits0) * N * K));
  const bool fast = operand_fast(d_dz, N, false, N, M) && operand_fast(d_x, K, false, K, M) &&
                    operand_fast(d_dz, N, true, M, N) && operand_fast(d_w, K, false, K, N);
  const dim3 grid(static_cast<unsigned>(dw_t * splits + dx_t));
#define DFM_LBWD(F, E)                                                                                            \
  hipLaunchKernelGGL((linear_bwd_kernel<F, E>), grid, dim3(kThreads), 0, as_stream(stream), d_dz, d_x, d_w, d_g_w, \
                     d_g_x, M, N, K, dw_tn, dw_t, splits, k_per_split, slabs, dw_counters, dx_tn, dbn, dfm_)
  const int epi = bn_below ? 1 : (fm ? 2 : 0);
  if (fast) {
    if (epi == 0) DFM_LBWD(true, 0); else if (epi == 1) DFM_LBWD(true, 1); else DFM_LBWD(true, 2);
  } else {
    if (epi == 0) DFM_LBWD(false, 0); else if (epi == 1) DFM_LBWD(false, 1); else DFM_LBWD(false, 2);
  }
#undef DFM_LBWD
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
