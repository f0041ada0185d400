// Fused DNN tower + head for the training step (reference deepfm/models/layers/dnn.py:45-55:
// [Linear -> BatchNorm1d -> ReLU -> Dropout] * n; deepfm.py:30-42: first_order + fm + Linear(dnn);
// trainer.py:59,221: BCEWithLogitsLoss).
//
// On an MI355X a dependent kernel costs ~4.5 us whatever it does, and the tower's GEMMs are
// 5-30 us each, so the tower is organised around FEW launches, and around launches WITHOUT serial
// tails (no "last workgroup finalises" inside a GEMM: the arrival atomic + two dependent passes over
// memory cost ~10 us per kernel):
//
//   forward, per layer   linear_bn_fwd        z = x W^T + b on the exact-fp32 MFMA tile loop
//                                             (gemm_core.h); the epilogue leaves per-column
//                                             (mean, M2) of each 32-row MFMA tile in a workspace
//                        bn_relu_dropout_apply a workgroup owns 64 columns x 32 rows: it first merges
//                                             the tile statistics of its columns (Chan, fixed order,
//                                             L2-resident) -> mean, rstd (+ running statistics from
//                                             the row-group-0 workgroups), then
//                                             a = dropout(relu(gamma*(z-mean)*rstd + beta))
//   head                 head_bce             logit = (fo + fm) + (a.w + b); BCE terms; d logit;
//                                             g = d logit * w pushed through the last BatchNorm's
//                                             ReLU/dropout mask (dy); per-workgroup partial sums of
//                                             dy, dy*xhat, d logit*a, loss, d logit
//   backward, per layer  bn_bwd_apply         merges the partial sums of its columns (from head_bce or
//                                             from the dx epilogue above it), adds d gamma / d beta
//                                             (and the head's d w, d b, loss), then
//                                             dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat))
//                        linear_bwd           ONE launch for both GEMMs of a Linear backward:
//                                             dW += dz^T x (batch split into slabs, the last
//                                             workgroup of an output tile adds the slabs in order)
//                                             and dx = dz W, whose epilogue is either the NEXT
//                                             (lower) layer's BatchNorm mask + per-tile column sums, or
//                                             the FM backward g_fm*(S - e) added in place (layer 1).
//
// Every reduction has a fixed association (no floating-point read-modify-write atomics): results
// are bitwise reproducible run to run.
#include "dropout.h"
#include "gemm_core.h"
#include "gemm_x6.h"

using namespace dfm;
using namespace dfm::gemm;

namespace {

// No kernel here hands data to another workgroup of the same launch.  Two earlier designs did
// ("the last workgroup of a column tile merges the statistics", "the last batch split adds the
// slabs") and both lost: with __threadfence() every workgroup writes back and invalidates its
// XCD's whole L2 (gfx950 has 8 non-coherent L2s) and the GEMM tiles beside it lose their operand
// reuse (3-6x slower kernels); with agent-scope relaxed atomics (sc1 loads/stores) instead of
// fences the kernels were correct and fast again, but each still ended in a serial tail — arrival
// atomic, then dependent passes over memory — of ~10 us.  Merging partial results in the CONSUMER
// kernel's prologue (L2-resident, fully parallel) costs nothing measurable.

struct BnBwd {            // device view of dfm_bn_bwd
  const float* z;
  const float* mean_rstd;
  const float* gamma;
  const float* beta;
  float* dy;
  float* g_gamma;
  float* g_beta;
  const int64_t* seed;
  float* partial;         // [T][2][N] per-tile column sums of dy, dy*xhat
  uint32_t thresh;
  float inv_keep;
  int salt;
};

// Epilogue shared by linear_bwd: the workgroup's writer waves hold g(m, n) for a 32 x 32 tile; push
// it through the BatchNorm's ReLU/dropout mask, store dy, and leave per-tile column sums of dy and
// dy*xhat in bn.partial (merged by bn_bwd_apply).  N = features of the BatchNorm layer.
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
    bn.partial[(static_cast<int64_t>(t) * 2 + 0) * N + n] = s1;
    bn.partial[(static_cast<int64_t>(t) * 2 + 1) * N + n] = s2;
  }
}

// ---- column-tile workgroups of the apply kernels ------------------------------------------------
// 256 threads = 16 column lanes (one float4 = 4 columns each: 64 columns) x 16 row lanes.
constexpr int kApThreads = 256, kApCols = 64, kApRowLanes = 16, kApRows = 32;

// Per-column totals over the 16 row lanes of two float4 accumulators (fixed order), to every thread.
__device__ __forceinline__ void row_lane_totals(float4& a, float4& b, float (*red)[kApRowLanes][kApCols]) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  st4(&red[0][rl][cl * 4], a);
  st4(&red[1][rl][cl * 4], b);
  __syncthreads();
  float4 ta = make_float4(0.f, 0.f, 0.f, 0.f), tb = ta;
#pragma unroll
  for (int i = 0; i < kApRowLanes; ++i) {
    const float4 x = ld4(&red[0][i][cl * 4]), y = ld4(&red[1][i][cl * 4]);
    ta.x += x.x; ta.y += x.y; ta.z += x.z; ta.w += x.w;
    tb.x += y.x; tb.y += y.y; tb.z += y.z; tb.w += y.w;
  }
  __syncthreads();
  a = ta; b = tb;
}

// Batch statistics of four columns from the merged sums (s1 = sum cnt * (tile mean - shift), s2 = sum of the
// tiles' M2 about the shift), and the running statistics' update.  Shared by bn_relu_dropout_apply_kernel and the
// head kernel that does the same work in its prologue; contraction is OFF here so that both get the same bits
// whatever the surrounding code lets the compiler fuse.
__device__ __forceinline__ void finish_column_stats(const float4& s1, const float4& s2, const float4& shift, int M,
                                                    float eps, float (&mu)[4], float (&rs)[4], float (&var)[4]) {
#pragma clang fp contract(off)
  const float invM = 1.f / static_cast<float>(M);
  const float s1a[4] = {s1.x, s1.y, s1.z, s1.w}, s2a[4] = {s2.x, s2.y, s2.z, s2.w};
  const float sh[4] = {shift.x, shift.y, shift.z, shift.w};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float d = s1a[u] * invM;               // mean - shift
    mu[u] = sh[u] + d;
    var[u] = fmaxf(s2a[u] * invM - d * d, 0.f);  // biased, as BN normalises
    rs[u] = rsqrtf(var[u] + eps);
  }
}
__device__ __forceinline__ void update_running_stats(float* __restrict__ running_mean, float* __restrict__ running_var,
                                                     int c, const float (&mu)[4], const float (&var)[4], int M,
                                                     float momentum) {
#pragma clang fp contract(off)
  const float unb = M > 1 ? static_cast<float>(M) / static_cast<float>(M - 1) : 1.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    running_mean[c + u] = (1.f - momentum) * running_mean[c + u] + momentum * mu[u];
    running_var[c + u] = (1.f - momentum) * running_var[c + u] + momentum * var[u] * unb;
  }
}

}  // namespace

// Epilogue of the forward tile (writer waves): z = acc + bias, per-tile column statistics (tile mean, M2).
__device__ __forceinline__ void fwd_tile_epilogue(const f32x16& acc, const TilePos& pos, int m0, int n0, int M, int N,
                                                  const float* __restrict__ bias, float* __restrict__ z,
                                                  float* __restrict__ partial) {
  const int n = n0 + pos.col();
  const bool okn = n < N;
  const float bv = (bias && okn) ? bias[n] : 0.f;
  const int cnt_i = M - (m0 + pos.wm) < 32 ? M - (m0 + pos.wm) : 32;   // valid rows of this MFMA tile
  float v[16];
  float sum = 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + pos.row(reg);
    v[reg] = acc[reg] + bv;
    if (m < M && okn) {
      z[static_cast<int64_t>(m) * N + n] = v[reg];
      sum += v[reg];
    }
  }
  sum += __shfl_xor(sum, 32, kWave);
  const float mean_t = cnt_i > 0 ? sum / static_cast<float>(cnt_i) : 0.f;
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
    partial[(static_cast<int64_t>(t) * 2 + 0) * N + n] = mean_t;     // tile mean
    partial[(static_cast<int64_t>(t) * 2 + 1) * N + n] = q;          // tile M2 = sum (z - tile mean)^2
  }
}

// =====================================================================================
// forward: z = x W^T + b, per-tile column statistics.  grid tiles_m * tiles_n (n fastest)
// =====================================================================================
template <bool FAST>
__global__ __launch_bounds__(kThreads) void linear_bn_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ z, int M, int N, int K, int tiles_n, float* __restrict__ partial) {
  __shared__ Smem sm;
  const TilePos pos;
  const int lt = xcd_logical_index(blockIdx.x, gridDim.x);
  const int m0 = (lt / tiles_n) * BM, n0 = (lt % tiles_n) * BN;
  f32x16 acc = {};
  mainloop<true, true, FAST, FAST>(x, ldx, w, K, M, N, m0, n0, 0, K, sm, pos, acc);
  if (pos.khalf == 1) return;
  fwd_tile_epilogue(acc, pos, m0, n0, M, N, bias, z, partial);
}

// a = dropout(relu(gamma * (z - mean) * rstd + beta)) for 64 columns x 32 rows per workgroup, after
// merging the (count, mean, M2) of the column's 32-row tiles (Chan's formula about tile 0's mean, so
// nothing cancels).  grid (column tiles, row groups); row group 0 also publishes the statistics.
// PLANES (dfm_tower_set_mode(2)): the workgroup also leaves its 32 x 64 tile of `a` as bf16 x 3 planes in both roles
// (gemm_x6.h) for the GEMMs that consume it; `out` may then be null.  M % 32 == 0 on that path.
template <bool PLANES>
__global__ __launch_bounds__(kApThreads) void bn_relu_dropout_apply_kernel(
    const float* __restrict__ z, int M, int N, const float* __restrict__ partial, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean_rstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ num_batches, float momentum, float eps, uint32_t thresh,
    float inv_keep, const int64_t* __restrict__ seed_ptr, int salt, float* __restrict__ out, Planes pf, Planes ps) {
  __shared__ float red[2][kApRowLanes][kApCols];
  __shared__ __attribute__((aligned(16))) float tile[PLANES ? kTileRows * kTileStride : 4];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * kApCols + cl * 4;
  const bool okc = c < N;                       // N % 4 == 0: a float4 is all in or all out
  const int cc = okc ? c : 0;
  const int T = (M + 31) / 32;
  const float4 m0v = ld4(partial + cc);         // tile 0's means: the shift
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
  for (int t = rl; t < T; t += kApRowLanes) {
    const float cnt = static_cast<float>(M - 32 * t < 32 ? M - 32 * t : 32);
    const float4 mt = ld4(partial + (static_cast<int64_t>(t) * 2 + 0) * N + cc);
    const float4 qt = ld4(partial + (static_cast<int64_t>(t) * 2 + 1) * N + cc);
    const float dx = mt.x - m0v.x, dy = mt.y - m0v.y, dz = mt.z - m0v.z, dw = mt.w - m0v.w;
    s1.x = fmaf(cnt, dx, s1.x); s1.y = fmaf(cnt, dy, s1.y); s1.z = fmaf(cnt, dz, s1.z); s1.w = fmaf(cnt, dw, s1.w);
    s2.x += fmaf(cnt * dx, dx, qt.x); s2.y += fmaf(cnt * dy, dy, qt.y);
    s2.z += fmaf(cnt * dz, dz, qt.z); s2.w += fmaf(cnt * dw, dw, qt.w);
  }
  row_lane_totals(s1, s2, red);
  float mu[4], rs[4], var[4];
  finish_column_stats(s1, s2, m0v, M, eps, mu, rs, var);
  if (blockIdx.y == 0 && rl == 0 && okc) {
    st4(mean_rstd + c, make_float4(mu[0], mu[1], mu[2], mu[3]));
    st4(mean_rstd + N + c, make_float4(rs[0], rs[1], rs[2], rs[3]));
    if (running_mean) update_running_stats(running_mean, running_var, c, mu, var, M, momentum);
    if (blockIdx.x == 0 && cl == 0 && num_batches) num_batches[0] += 1;
  }
  if (!PLANES && !okc) return;
  const int r0 = blockIdx.y * kApRows;
  if (okc) {
    const int64_t seed = seed_ptr ? seed_ptr[0] : 0;
    const float4 ga = ld4(gamma + c), be = ld4(beta + c);
    const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int i = 0; i < kApRows / kApRowLanes; ++i) {
      const int m = r0 + rl + i * kApRowLanes;
      if (m < M) {
        const int64_t idx = static_cast<int64_t>(m) * N + c;
        const float4 zv = ld4(z + idx);
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          o[u] = fmaxf(fmaf(gav[u], (zz[u] - mu[u]) * rs[u], bev[u]), 0.f) * drop_scale(seed, salt, idx + u, thresh, inv_keep);
        if (!PLANES || out) st4(out + idx, make_float4(o[0], o[1], o[2], o[3]));
        if (PLANES) st4(&tile[(rl + i * kApRowLanes) * kTileStride + cl * 4], make_float4(o[0], o[1], o[2], o[3]));
      }
    }
  }
  if (PLANES) {
    __syncthreads();
    emit_planes_from_tile(tile, r0, blockIdx.x * kApCols, M, N, pf, ps);
  }
}

// Sums the head's per-workgroup partials that bn_bwd_apply's column merge does not cover.
struct HeadTail {
  float* g_w;       // (K) += sum_b dlogit_b * a[b, :]
  float* g_b;       // (1) += sum_b dlogit_b     (may be NULL)
  float* loss;      // (1)  = mean BCE
  float* g_b2;      // (1) += the same sum: the bias of another 1-wide Linear added to the logit (may be NULL)
  int enabled;
};

// dz = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)) for 64 columns x 32 rows per workgroup,
// after merging the per-tile column sums of dy and dy*xhat: `partial` has T rows of `stride` floats,
// the two planes at column offsets 0 and off1 (dx epilogue: [T][2][N]; head_bce: [blocks][3K+2]).
// Row group 0 adds d gamma / d beta (and finishes the head's d w, d b, loss).  dz may alias dy.
template <bool PLANES>   // as bn_relu_dropout_apply_kernel: d z also as planes (both roles); dz may then be null
__global__ __launch_bounds__(kApThreads) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ z, int M, int N, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ partial, int T, int stride, int off1,
    float* __restrict__ g_gamma, float* __restrict__ g_beta, HeadTail head, float* __restrict__ dz, Planes pf,
    Planes ps) {
  __shared__ float red[2][kApRowLanes][kApCols];
  __shared__ __attribute__((aligned(16))) float tile[PLANES ? kTileRows * kTileStride : 4];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * kApCols + cl * 4;
  const bool okc = c < N;
  const int cc = okc ? c : 0;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
  for (int t = rl; t < T; t += kApRowLanes) {
    const float4 a = ld4(partial + static_cast<int64_t>(t) * stride + cc);
    const float4 b = ld4(partial + static_cast<int64_t>(t) * stride + off1 + cc);
    s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
    s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
  }
  row_lane_totals(s1, s2, red);
  if (blockIdx.y == 0) {
    if (rl == 0 && okc) {
      const float4 gb = ld4(g_beta + c), gg = ld4(g_gamma + c);
      st4(g_beta + c, make_float4(gb.x + s1.x, gb.y + s1.y, gb.z + s1.z, gb.w + s1.w));
      st4(g_gamma + c, make_float4(gg.x + s2.x, gg.y + s2.y, gg.z + s2.z, gg.w + s2.w));
    }
    if (head.enabled) {     // third plane (d logit * a) for these columns; the two scalars from column tile 0
      float4 s3 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = s3;
#pragma unroll 4
      for (int t = rl; t < T; t += kApRowLanes) {
        const float4 a = ld4(partial + static_cast<int64_t>(t) * stride + 2 * off1 + cc);
        s3.x += a.x; s3.y += a.y; s3.z += a.z; s3.w += a.w;
        if (cl == 0) {      // loss and d logit sums sit behind the three planes
          s4.x += partial[static_cast<int64_t>(t) * stride + 3 * off1];
          s4.y += partial[static_cast<int64_t>(t) * stride + 3 * off1 + 1];
        }
      }
      row_lane_totals(s3, s4, red);
      if (rl == 0 && okc) {
        const float4 gw = ld4(head.g_w + c);
        st4(head.g_w + c, make_float4(gw.x + s3.x, gw.y + s3.y, gw.z + s3.z, gw.w + s3.w));
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        head.loss[0] = s4.x / static_cast<float>(M);
        if (head.g_b) head.g_b[0] += s4.y;
        if (head.g_b2) head.g_b2[0] += s4.y;
      }
    }
  }
  if (!PLANES && !okc) return;
  const int r0 = blockIdx.y * kApRows;
  if (okc) {
    const float invM = 1.f / static_cast<float>(M);
    const float4 muv = ld4(mean_rstd + c), rsv = ld4(mean_rstd + N + c), gav = ld4(gamma + c);
    const float mu[4] = {muv.x, muv.y, muv.z, muv.w}, rs[4] = {rsv.x, rsv.y, rsv.z, rsv.w};
    const float ga[4] = {gav.x, gav.y, gav.z, gav.w};
    const float m1[4] = {s1.x * invM, s1.y * invM, s1.z * invM, s1.w * invM};
    const float m2[4] = {s2.x * invM, s2.y * invM, s2.z * invM, s2.w * invM};
#pragma unroll
    for (int i = 0; i < kApRows / kApRowLanes; ++i) {
      const int m = r0 + rl + i * kApRowLanes;
      if (m < M) {
        const int64_t idx = static_cast<int64_t>(m) * N + c;
        const float4 dv = ld4(dy + idx), zv = ld4(z + idx);
        const float dd[4] = {dv.x, dv.y, dv.z, dv.w}, zz[4] = {zv.x, zv.y, zv.z, zv.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = ga[u] * rs[u] * (dd[u] - m1[u] - (zz[u] - mu[u]) * rs[u] * m2[u]);
        if (!PLANES || dz) st4(dz + idx, make_float4(o[0], o[1], o[2], o[3]));
        if (PLANES) st4(&tile[(rl + i * kApRowLanes) * kTileStride + cl * 4], make_float4(o[0], o[1], o[2], o[3]));
      }
    }
  }
  if (PLANES) {
    __syncthreads();
    emit_planes_from_tile(tile, r0, blockIdx.x * kApCols, M, N, pf, ps);
  }
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

// partials per workgroup (merged by bn_bwd_apply): [3][K] column sums (dy, dy*xhat, dlogit*a),
// then loss and dlogit sums, padded to 3K + 4 floats
// What bn_relu_dropout_apply_kernel needs beyond a BnBwd when the head does its work (FUSE).
struct BnFwdTail {
  const float* partial;     // per-tile column statistics of linear_bn_fwd
  float* mean_rstd;         // written by workgroup 0
  float* running_mean;
  float* running_var;
  int64_t* num_batches;
  float momentum, eps;
};

// FUSE: the last BatchNorm -> ReLU -> Dropout of the tower happens HERE instead of in a
// bn_relu_dropout_apply launch of its own: every workgroup merges the tile statistics of all K columns (what
// each workgroup of that launch did for its 64 columns: the same loads, the same order, the same numbers),
// a = dropout(relu(gamma * (z - mean) * rstd + beta)) is formed in registers from z and never stored — the
// backward wants z, the statistics and the mask, not a.  One launch (~5 us) less per step.
template <int CH, bool FUSE>   // CH = K / 32 float4 chunks per lane
__global__ __launch_bounds__(kHeadThreads) void head_bce_kernel(
    const float* __restrict__ a, int M, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ fo, const float* __restrict__ fm, const float* __restrict__ labels,
    float* __restrict__ logits, float* __restrict__ dlogit, float* __restrict__ g_a, BnBwd bn, int has_bn,
    float* __restrict__ hpart, BnFwdTail ft) {
  constexpr int K = CH * 32;
  constexpr int P = 3 * K + 4;
  __shared__ float red[kHeadThreads / kWave][P];
  __shared__ __attribute__((aligned(16))) float s_mu[FUSE ? K : 4], s_rs[FUSE ? K : 4];
  const int tid = threadIdx.x, l8 = tid & (kHeadLPR - 1), rl = tid / kHeadLPR;
  const int m = blockIdx.x * kHeadRows + rl;
  const bool live = m < M;
  const int mc = live ? m : M - 1;
  if (FUSE) {
    static_assert(kHeadThreads == kApThreads, "the statistics merge uses the apply kernel's thread layout");
    __shared__ float mred[2][kApRowLanes][kApCols];
    const int cl = tid & 15, ml = tid >> 4;
    const int T = (M + 31) / 32;
    for (int c0 = 0; c0 < K; c0 += kApCols) {
      const int c = c0 + cl * 4;
      const bool okc = c < K;
      const int cc = okc ? c : 0;
      const float4 m0v = ld4(ft.partial + cc);
      float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
      for (int t = ml; t < T; t += kApRowLanes) {
        const float cnt = static_cast<float>(M - 32 * t < 32 ? M - 32 * t : 32);
        const float4 mt = ld4(ft.partial + (static_cast<int64_t>(t) * 2 + 0) * K + cc);
        const float4 qt = ld4(ft.partial + (static_cast<int64_t>(t) * 2 + 1) * K + cc);
        const float dx = mt.x - m0v.x, dy = mt.y - m0v.y, dz = mt.z - m0v.z, dw = mt.w - m0v.w;
        s1.x = fmaf(cnt, dx, s1.x); s1.y = fmaf(cnt, dy, s1.y); s1.z = fmaf(cnt, dz, s1.z); s1.w = fmaf(cnt, dw, s1.w);
        s2.x += fmaf(cnt * dx, dx, qt.x); s2.y += fmaf(cnt * dy, dy, qt.y);
        s2.z += fmaf(cnt * dz, dz, qt.z); s2.w += fmaf(cnt * dw, dw, qt.w);
      }
      row_lane_totals(s1, s2, mred);
      float mu[4], rs[4], var[4];
      finish_column_stats(s1, s2, m0v, M, ft.eps, mu, rs, var);
      if (ml == 0 && okc) {
        st4(s_mu + c, make_float4(mu[0], mu[1], mu[2], mu[3]));
        st4(s_rs + c, make_float4(rs[0], rs[1], rs[2], rs[3]));
        if (blockIdx.x == 0) {
          st4(ft.mean_rstd + c, make_float4(mu[0], mu[1], mu[2], mu[3]));
          st4(ft.mean_rstd + K + c, make_float4(rs[0], rs[1], rs[2], rs[3]));
          if (ft.running_mean) update_running_stats(ft.running_mean, ft.running_var, c, mu, var, M, ft.momentum);
          if (c == 0 && ft.num_batches) ft.num_batches[0] += 1;
        }
      }
    }
    __syncthreads();
  }
  const int64_t seed = (has_bn && bn.seed) ? bn.seed[0] : 0;
  float4 av[CH], wv[CH];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int j = i * 32 + l8 * 4;
    if (FUSE) {
      const float4 zv = ld4(bn.z + static_cast<int64_t>(mc) * K + j), mu = ld4(s_mu + j), rs = ld4(s_rs + j),
                   ga = ld4(bn.gamma + j), be = ld4(bn.beta + j);
      const int64_t idx = static_cast<int64_t>(mc) * K + j;
      av[i].x = fmaxf(fmaf(ga.x, (zv.x - mu.x) * rs.x, be.x), 0.f) * drop_scale(seed, bn.salt, idx + 0, bn.thresh, bn.inv_keep);
      av[i].y = fmaxf(fmaf(ga.y, (zv.y - mu.y) * rs.y, be.y), 0.f) * drop_scale(seed, bn.salt, idx + 1, bn.thresh, bn.inv_keep);
      av[i].z = fmaxf(fmaf(ga.z, (zv.z - mu.z) * rs.z, be.z), 0.f) * drop_scale(seed, bn.salt, idx + 2, bn.thresh, bn.inv_keep);
      av[i].w = fmaxf(fmaf(ga.w, (zv.w - mu.w) * rs.w, be.w), 0.f) * drop_scale(seed, bn.salt, idx + 3, bn.thresh, bn.inv_keep);
    } else {
      av[i] = ld4(a + static_cast<int64_t>(mc) * K + j);
    }
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
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int j = i * 32 + l8 * 4;
    const float gv[4] = {dl * wv[i].x, dl * wv[i].y, dl * wv[i].z, dl * wv[i].w};
    const float aa[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
    float dyv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bn) {
      const float4 zv = ld4(bn.z + static_cast<int64_t>(mc) * K + j), mu = ld4(FUSE ? s_mu + j : bn.mean_rstd + j),
                   rs = ld4(FUSE ? s_rs + j : bn.mean_rstd + K + j), ga = ld4(bn.gamma + j), be = ld4(bn.beta + j);
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
  if (lane == 0) {
    red[wave][3 * K] = sl; red[wave][3 * K + 1] = sd;
    red[wave][3 * K + 2] = 0.f; red[wave][3 * K + 3] = 0.f;
  }
  __syncthreads();
  for (int o = tid; o < P; o += kHeadThreads)
    hpart[static_cast<int64_t>(blockIdx.x) * P + o] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
}

// =====================================================================================
// backward of one Linear: dW += dz^T x  and  dx = dz W (+ epilogue), one launch.
//   blocks [0, dw_blocks): dW tiles x splits;  the rest: dx tiles
// =====================================================================================
struct FmBwd {
  const float* g_fm;     // (M)      d loss / d fm value, or null (no FM term)
  const float* fm_sum;   // (M, D)   sum_f e
  const float* e;        // (M, K)   field embeddings
  const float* addend;   // (M, K)   gradient of another consumer of the embeddings (CIN / attention), or null
  int dim;
};

// d weight tile (N x K slab of one batch split; summed by slab_reduce_kernel / the optimizer's prepare launch)
__device__ __forceinline__ void dw_tile_store(const f32x16& acc, const TilePos& pos, int m0, int n0, int N, int K,
                                              float* __restrict__ sl) {
  const int n = n0 + pos.col();
  if (pos.khalf == 0 && n < K) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + pos.row(reg);
      if (m < N) sl[static_cast<int64_t>(m) * K + n] = acc[reg];
    }
  }
}

// d input tile: EPI 0 plain store, 1 the lower layer's BatchNorm mask, 2 + FM backward / another consumer's gradient
template <int EPI>
__device__ __forceinline__ void dx_tile_epilogue(const f32x16& acc, const TilePos& pos, int m0, int n0, int M, int K,
                                                 float* __restrict__ g_x, const BnBwd& bn, const FmBwd& fmb) {
  if (pos.khalf == 1) return;
  if (EPI == 1) {
    bn_mask_tile(bn, acc, pos, m0, n0, M, K);
    return;
  }
  const int n = n0 + pos.col();
  if (n >= K) return;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + pos.row(reg);
    if (m < M) {
      const int64_t off = static_cast<int64_t>(m) * K + n;
      float v = acc[reg];
      if (EPI == 2) {   // d e = d flat + g_fm * (S - e) (fm.py:18-23 backward) + what another layer sent back
        if (fmb.g_fm) v += fmb.g_fm[m] * (fmb.fm_sum[static_cast<int64_t>(m) * fmb.dim + n % fmb.dim] - fmb.e[off]);
        if (fmb.addend) v += fmb.addend[off];
      }
      g_x[off] = v;
    }
  }
}

// X3: both products on the bf16 matrix pipe with the bf16 x 3 split (gemm_core.h::mainloop_x3; FAST operands only) —
// selected by dfm_tower_set_mode(1), never implicitly.
template <bool FAST, int EPI, bool X3 = false>   // EPI 0: plain store, 1: BatchNorm mask of the lower layer, 2: + FM backward
__global__ __launch_bounds__(kThreads) void linear_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ w,
    float* __restrict__ g_x, int M, int N, int K, int dw_tiles_n, int dw_tiles, int splits,
    int dw_blocks, int k_per_split, float* __restrict__ slabs, int dx_tiles_n, BnBwd bn, FmBwd fmb) {
  __shared__ typename std::conditional<X3, SmemX3, Smem>::type sm;
  const TilePos pos;
  f32x16 acc = {};
  const int wg = blockIdx.x;
  const int bid = wg < dw_blocks ? xcd_logical_index(wg, dw_blocks)
                                 : dw_blocks + xcd_logical_index_from(wg - dw_blocks, static_cast<int>(gridDim.x) - dw_blocks, dw_blocks);
  if (bid < dw_blocks) {
    // ---- dW (N x K) = sum over the batch: A = dz (k-strided), B = x (k-strided) ----
    const int tile = bid % dw_tiles, sp = bid / dw_tiles;
    const int m0 = (tile / dw_tiles_n) * BM, n0 = (tile % dw_tiles_n) * BN;
    const int kb = sp * k_per_split;
    const int ke = kb + k_per_split < M ? kb + k_per_split : M;
    if constexpr (X3) mainloop_x3<false, false>(dz, N, x, K, N, K, m0, n0, kb, ke, sm, pos, acc);
    else mainloop<false, false, FAST, FAST>(dz, N, x, K, N, K, m0, n0, kb, ke, sm, pos, acc);
    dw_tile_store(acc, pos, m0, n0, N, K, slabs + static_cast<int64_t>(sp) * N * K);
    return;
  }
  // ---- dx (M x K) = dz W: A = dz (k-contiguous), B = W (k-strided) ----
  const int t = bid - dw_blocks;
  const int m0 = (t / dx_tiles_n) * BM, n0 = (t % dx_tiles_n) * BN;
  if constexpr (X3) mainloop_x3<true, false>(dz, N, w, K, M, K, m0, n0, 0, N, sm, pos, acc);
  else mainloop<true, false, FAST, FAST>(dz, N, w, K, M, K, m0, n0, 0, N, sm, pos, acc);
  dx_tile_epilogue<EPI>(acc, pos, m0, n0, M, K, g_x, bn, fmb);
}

// g_w += sum_s slab_s (fixed order) for every Linear of the tower in ONE launch: the batch-split
// partial products of all dfm_linear_backward calls of a step are finished together, so no GEMM
// workgroup waits on an arrival counter or walks the slabs serially.
constexpr int kMaxSlabRefs = 16;
struct SlabRefs {
  const float* slabs[kMaxSlabRefs];
  float* g[kMaxSlabRefs];
  int elems4[kMaxSlabRefs];        // elements / 4 (out*in is a multiple of 4 on this path)
  int splits[kMaxSlabRefs];
  int first_block[kMaxSlabRefs + 1];
  int count;
};
__global__ __launch_bounds__(256) void slab_reduce_kernel(SlabRefs refs) {
  int e = 0;
  while (e + 1 < refs.count && static_cast<int>(blockIdx.x) >= refs.first_block[e + 1]) ++e;
  const int i4 = (blockIdx.x - refs.first_block[e]) * 256 + threadIdx.x;
  if (i4 >= refs.elems4[e]) return;
  const float* sl = refs.slabs[e] + static_cast<int64_t>(i4) * 4;
  const int64_t stride = static_cast<int64_t>(refs.elems4[e]) * 4;
  const int splits = refs.splits[e];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int q = 0;
  for (; q + 8 <= splits; q += 8) {      // 8 loads in flight, added in order
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = ld4(sl + (q + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
  }
  for (; q < splits; ++q) {
    const float4 t = ld4(sl + q * stride);
    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
  }
  float* g = refs.g[e] + static_cast<int64_t>(i4) * 4;
  const float4 old = ld4(g);
  st4(g, make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w));
}

// =====================================================================================
// host side
// =====================================================================================
namespace {
inline int tiles(int n, int t) { return (n + t - 1) / t; }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// per-tile column statistics of a (m, n) activation: [ceil(m/32)][2][n] floats
inline size_t tile_partial_bytes(int64_t m, int n) { return align256(sizeof(float) * 2 * static_cast<size_t>((m + 31) / 32) * n); }
inline size_t head_partial_bytes(int64_t m, int k) {
  return align256(sizeof(float) * static_cast<size_t>((m + kHeadRows - 1) / kHeadRows) * (3 * static_cast<size_t>(k) + 4));
}

// Batch splits of the d-weight product.  The launch also carries the d-input tiles, and a CU holds
// two of these 8-wave workgroups, so the grid runs in rounds of 2 x 256 workgroups: a round that is
// a quarter full costs as much as a full one (measured: 1160 workgroups = 2.3 rounds took 50 us,
// the MFMA work in it 18 us).  Pick the split so that d-weight + d-input workgroups fill whole rounds.
int dw_splits(int n_out, int k_in, int64_t m) {
  constexpr int64_t kRound = 512;
  const int64_t t = static_cast<int64_t>(tiles(n_out, BM)) * tiles(k_in, BN);
  const int64_t dx = ((m + BM - 1) / BM) * tiles(k_in, BN);
  const int64_t max_s = m / (4 * BK) > 0 ? m / (4 * BK) : 1;   // at least 4 k-slices per split
  int64_t room = (dx / kRound + 1) * kRound - dx;
  int64_t s = room / t;
  if (s < 4 && max_s >= 4) s = (room + kRound) / t;            // too coarse: spill into one more round
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : static_cast<int>(s);
}

bool fill_bn(const dfm_bn_bwd* h, BnBwd* d) {
  if (!h->z || !h->mean_rstd || !h->gamma || !h->beta || !h->dy || !h->g_gamma || !h->g_beta || !h->workspace)
    return false;
  if (!(h->p_drop >= 0.f && h->p_drop < 1.f) || (h->p_drop > 0.f && !h->seed)) return false;
  d->z = h->z; d->mean_rstd = h->mean_rstd; d->gamma = h->gamma; d->beta = h->beta;
  d->dy = h->dy; d->g_gamma = h->g_gamma; d->g_beta = h->g_beta;
  d->seed = h->seed;
  d->partial = static_cast<float*>(h->workspace);
  d->thresh = dropout_thresh(h->p_drop);
  d->inv_keep = 1.f / (1.f - h->p_drop);
  d->salt = h->salt;
  return true;
}
}  // namespace

extern "C" size_t dfm_bn_bwd_workspace_bytes(int64_t batch, int features) {
  const size_t a = tile_partial_bytes(batch, features), b = head_partial_bytes(batch, features);
  return a > b ? a : b;
}

extern "C" size_t dfm_linear_bn_workspace_bytes(int64_t batch, int features) {
  return tile_partial_bytes(batch, features);
}

extern "C" int dfm_linear_bn_forward(const float* d_x, int64_t ldx, const float* d_w, const float* d_bias,
                                     int64_t batch, int out_features, int in_features, float* d_z,
                                     void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w && d_z && d_workspace, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && out_features > 0 && in_features > 0, "bad shape");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  float* partial = static_cast<float*>(d_workspace);
  const int tn = tiles(N, BN);
  const dim3 grid(static_cast<unsigned>(tn) * tiles(M, BM));
  const bool fast = operand_fast(d_x, ldx, true, M, K) && operand_fast(d_w, K, true, N, K);
  if (fast)
    hipLaunchKernelGGL(linear_bn_fwd_kernel<true>, grid, dim3(kThreads), 0, as_stream(stream), d_x, ldx, d_w, d_bias,
                       d_z, M, N, K, tn, partial);
  else
    hipLaunchKernelGGL(linear_bn_fwd_kernel<false>, grid, dim3(kThreads), 0, as_stream(stream), d_x, ldx, d_w, d_bias,
                       d_z, M, N, K, tn, partial);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

namespace {
// planes arguments of the apply entry points: both roles or none; batch % 32 == 0 and features % 8 == 0
bool planes_args_ok(const void* pf, const void* ps, int64_t batch, int features) {
  if (!pf && !ps) return true;
  return pf && ps && aligned16(pf) && aligned16(ps) && batch % 32 == 0 && features % 8 == 0;
}
}  // namespace

static int bn_relu_dropout_apply_impl(const float* d_z, int64_t batch, int features, const void* d_workspace,
                                      const float* d_gamma, const float* d_beta, float* d_mean_rstd,
                                      float* d_running_mean, float* d_running_var, int64_t* d_num_batches,
                                      float momentum, float eps, float p_drop, const int64_t* d_seed, int salt,
                                      float* d_out, void* d_planes_f, void* d_planes_s, dfm_stream_t stream) {
  DFM_REQUIRE(d_z && d_workspace && d_gamma && d_beta && d_mean_rstd && (d_out || d_planes_f), "null argument");
  DFM_REQUIRE(planes_args_ok(d_planes_f, d_planes_s, batch, features),
              "planes: both roles, 16-byte aligned, batch %% 32 == 0, features %% 8 == 0");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && features > 0 && features % 4 == 0,
              "features must be a positive multiple of 4");
  DFM_REQUIRE(aligned16(d_z) && aligned16(d_workspace) && aligned16(d_mean_rstd) && aligned16(d_gamma) &&
                  aligned16(d_beta) && (!d_out || aligned16(d_out)), "pointers must be 16-byte aligned");
  DFM_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || d_seed), "bad dropout arguments");
  const int M = static_cast<int>(batch);
  const dim3 grid(tiles(features, kApCols), tiles(M, kApRows));
  if (d_planes_f) {
    hipLaunchKernelGGL(bn_relu_dropout_apply_kernel<true>, grid, dim3(kApThreads), 0, as_stream(stream), d_z, M,
                       features, static_cast<const float*>(d_workspace), d_gamma, d_beta, d_mean_rstd, d_running_mean,
                       d_running_var, d_num_batches, momentum, eps, dropout_thresh(p_drop), 1.f / (1.f - p_drop),
                       d_seed, salt, d_out, make_planes(d_planes_f, M, features), make_planes(d_planes_s, features, M));
  } else {
    hipLaunchKernelGGL(bn_relu_dropout_apply_kernel<false>, grid, dim3(kApThreads), 0, as_stream(stream), d_z, M,
                       features, static_cast<const float*>(d_workspace), d_gamma, d_beta, d_mean_rstd, d_running_mean,
                       d_running_var, d_num_batches, momentum, eps, dropout_thresh(p_drop), 1.f / (1.f - p_drop),
                       d_seed, salt, d_out, Planes{}, Planes{});
  }
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_bn_relu_dropout_apply(const float* d_z, int64_t batch, int features, const void* d_workspace,
                                         const float* d_gamma, const float* d_beta, float* d_mean_rstd,
                                         float* d_running_mean, float* d_running_var, int64_t* d_num_batches,
                                         float momentum, float eps, float p_drop, const int64_t* d_seed, int salt,
                                         float* d_out, dfm_stream_t stream) {
  DFM_REQUIRE(d_out, "null argument");
  return bn_relu_dropout_apply_impl(d_z, batch, features, d_workspace, d_gamma, d_beta, d_mean_rstd, d_running_mean,
                                    d_running_var, d_num_batches, momentum, eps, p_drop, d_seed, salt, d_out, nullptr,
                                    nullptr, stream);
}

extern "C" int dfm_bn_relu_dropout_apply_planes(const float* d_z, int64_t batch, int features, const void* d_workspace,
                                                const float* d_gamma, const float* d_beta, float* d_mean_rstd,
                                                float* d_running_mean, float* d_running_var, int64_t* d_num_batches,
                                                float momentum, float eps, float p_drop, const int64_t* d_seed,
                                                int salt, float* d_out, void* d_planes_f, void* d_planes_s,
                                                dfm_stream_t stream) {
  DFM_REQUIRE(d_planes_f && d_planes_s, "null argument");
  return bn_relu_dropout_apply_impl(d_z, batch, features, d_workspace, d_gamma, d_beta, d_mean_rstd, d_running_mean,
                                    d_running_var, d_num_batches, momentum, eps, p_drop, d_seed, salt, d_out,
                                    d_planes_f, d_planes_s, stream);
}

static int bn_backward_apply_impl(const dfm_bn_bwd* bn, int64_t batch, int features, const dfm_head_tail* head,
                                  float* d_dz, void* d_planes_f, void* d_planes_s, dfm_stream_t stream) {
  DFM_REQUIRE(bn && (d_dz || d_planes_f), "null argument");
  DFM_REQUIRE(planes_args_ok(d_planes_f, d_planes_s, batch, features),
              "planes: both roles, 16-byte aligned, batch %% 32 == 0, features %% 8 == 0");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && features > 0 && features % 4 == 0,
              "features must be a positive multiple of 4");
  BnBwd d = {};
  DFM_REQUIRE(fill_bn(bn, &d), "incomplete dfm_bn_bwd");
  DFM_REQUIRE(aligned16(d.dy) && aligned16(d.z) && aligned16(d.mean_rstd) && aligned16(d.gamma) &&
                  aligned16(d.partial) && aligned16(d.g_gamma) && aligned16(d.g_beta) && (!d_dz || aligned16(d_dz)),
              "pointers must be 16-byte aligned");
  const int M = static_cast<int>(batch);
  HeadTail ht = {};
  int T = (M + 31) / 32, stride = 2 * features, off1 = features;
  if (head) {       // the mask came from dfm_head_bce: its workgroup partials, not the dx epilogue's
    DFM_REQUIRE(head->g_w && head->loss && aligned16(head->g_w), "incomplete dfm_head_tail");
    ht.g_w = head->g_w; ht.g_b = head->g_b; ht.loss = head->loss; ht.g_b2 = head->g_b2; ht.enabled = 1;
    T = (M + kHeadRows - 1) / kHeadRows;
    stride = 3 * features + 4;
  }
  const dim3 grid(tiles(features, kApCols), tiles(M, kApRows));
  if (d_planes_f) {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(kApThreads), 0, as_stream(stream), d.dy, d.z, M, features,
                       d.mean_rstd, d.gamma, d.partial, T, stride, off1, d.g_gamma, d.g_beta, ht, d_dz,
                       make_planes(d_planes_f, M, features), make_planes(d_planes_s, features, M));
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(kApThreads), 0, as_stream(stream), d.dy, d.z, M, features,
                       d.mean_rstd, d.gamma, d.partial, T, stride, off1, d.g_gamma, d.g_beta, ht, d_dz, Planes{},
                       Planes{});
  }
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_bn_backward_apply(const dfm_bn_bwd* bn, int64_t batch, int features, const dfm_head_tail* head,
                                     float* d_dz, dfm_stream_t stream) {
  DFM_REQUIRE(d_dz, "null argument");
  return bn_backward_apply_impl(bn, batch, features, head, d_dz, nullptr, nullptr, stream);
}

extern "C" int dfm_bn_backward_apply_planes(const dfm_bn_bwd* bn, int64_t batch, int features,
                                            const dfm_head_tail* head, float* d_dz, void* d_planes_f,
                                            void* d_planes_s, dfm_stream_t stream) {
  DFM_REQUIRE(d_planes_f && d_planes_s, "null argument");
  return bn_backward_apply_impl(bn, batch, features, head, d_dz, d_planes_f, d_planes_s, stream);
}

namespace {
template <bool FUSE>
int launch_head(const float* d_a, int64_t batch, int features, const float* d_w, const float* d_b,
                const float* d_first_order, const float* d_fm, const float* d_labels, float* d_logits,
                float* d_g_logits, const BnBwd& dbn, const BnFwdTail& ft, hipStream_t st) {
  const int M = static_cast<int>(batch);
  const unsigned blocks = static_cast<unsigned>((batch + kHeadRows - 1) / kHeadRows);
#define DFM_HEAD(CH)                                                                                             \
  hipLaunchKernelGGL((head_bce_kernel<CH, FUSE>), dim3(blocks), dim3(kHeadThreads), 0, st, d_a, M, d_w, d_b,     \
                     d_first_order, d_fm, d_labels, d_logits, d_g_logits, static_cast<float*>(nullptr), dbn, 1,  \
                     dbn.partial, ft)
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
}  // namespace

extern "C" int dfm_head_bce(const float* d_a, int64_t batch, int features, const float* d_w, const float* d_b,
                            const float* d_first_order, const float* d_fm, const float* d_labels, float* d_logits,
                            float* d_g_logits, const dfm_bn_bwd* bn, dfm_stream_t stream) {
  DFM_REQUIRE(d_a && d_w && d_labels && d_logits && d_g_logits && bn, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30), "bad batch");
  DFM_REQUIRE(features > 0 && features % 32 == 0 && features <= 32 * kHeadMaxChunks,
              "head features must be a multiple of 32, at most 256");
  DFM_REQUIRE(aligned16(d_a) && aligned16(d_w), "pointers must be 16-byte aligned");
  BnBwd dbn = {};
  DFM_REQUIRE(fill_bn(bn, &dbn), "incomplete dfm_bn_bwd");
  return launch_head<false>(d_a, batch, features, d_w, d_b, d_first_order, d_fm, d_labels, d_logits, d_g_logits, dbn,
                            BnFwdTail{}, as_stream(stream));
}

extern "C" int dfm_head_bn_bce(const void* d_fwd_workspace, float* d_mean_rstd, float* d_running_mean,
                               float* d_running_var, int64_t* d_num_batches, float momentum, float eps,
                               int64_t batch, int features, const float* d_w, const float* d_b,
                               const float* d_first_order, const float* d_fm, const float* d_labels,
                               float* d_logits, float* d_g_logits, const dfm_bn_bwd* bn, dfm_stream_t stream) {
  DFM_REQUIRE(d_fwd_workspace && d_mean_rstd && d_w && d_labels && d_logits && d_g_logits && bn, "null argument");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30), "bad batch");
  DFM_REQUIRE(features > 0 && features % 32 == 0 && features <= 32 * kHeadMaxChunks,
              "head features must be a multiple of 32, at most 256");
  BnBwd dbn = {};
  DFM_REQUIRE(fill_bn(bn, &dbn), "incomplete dfm_bn_bwd");
  DFM_REQUIRE(dbn.mean_rstd == d_mean_rstd, "bn->mean_rstd must be the buffer the statistics are written to");
  DFM_REQUIRE(aligned16(d_fwd_workspace) && aligned16(d_mean_rstd) && aligned16(d_w) && aligned16(dbn.z) &&
                  aligned16(dbn.gamma) && aligned16(dbn.beta), "pointers must be 16-byte aligned");
  BnFwdTail ft = {static_cast<const float*>(d_fwd_workspace), d_mean_rstd, d_running_mean, d_running_var,
                  d_num_batches, momentum, eps};
  return launch_head<true>(nullptr, batch, features, d_w, d_b, d_first_order, d_fm, d_labels, d_logits, d_g_logits,
                           dbn, ft, as_stream(stream));
}

// =====================================================================================
// A Linear with ONE output added to the logit (xDeepFM's cin_linear, xdeepfm.py:41-47): forward a row dot
// product, backward an outer product and a column sum — a few microseconds of memory traffic that cost six
// generic GEMM / reduce launches (43 us) before.
// =====================================================================================
namespace {
constexpr int kL1Threads = 256;
constexpr int kL1Rows = 64;          // rows per workgroup (= per slab) of the backward
}  // namespace

// out[b] = x[b, :] . w (+ bias): one wave per row
__global__ __launch_bounds__(kL1Threads) void linear1_fwd_kernel(const float* __restrict__ x, int64_t M, int K,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out) {
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kL1Threads / kWave) + wave;
  if (row >= M) return;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += kWave * 4) {
    const float4 xv = ld4(x + row * K + k), wv = ld4(w + k);
    acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
  if (lane == 0) out[row] = acc + (bias ? bias[0] : 0.f);
}

// g_x[b, :] = g[b] * w;  slab[blockIdx][k] = sum over the workgroup's rows of g[b] * x[b, k] (fixed order),
// added into the weight gradient by dfm_linear_backward_finish (a dfm_slab_ref with splits = gridDim.x)
__global__ __launch_bounds__(kL1Threads) void linear1_bwd_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ x, int64_t M, int K,
                                                                 const float* __restrict__ w,
                                                                 float* __restrict__ g_x, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float red[kL1Threads * 4];
  const int c4n = K / 4;                                   // <= kL1Threads (host)
  const int rln = kL1Threads / c4n;                        // row lanes
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const bool active = rl < rln;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kL1Rows;
  const float4 wv = ld4(w + 4 * cl);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    for (int r = rl; r < kL1Rows; r += rln) {
      const int64_t row = r0 + r;
      if (row >= M) break;
      const float gv = g[row];
      const float4 xv = ld4(x + row * K + 4 * cl);
      st4(g_x + row * K + 4 * cl, make_float4(gv * wv.x, gv * wv.y, gv * wv.z, gv * wv.w));
      acc.x = fmaf(gv, xv.x, acc.x); acc.y = fmaf(gv, xv.y, acc.y);
      acc.z = fmaf(gv, xv.z, acc.z); acc.w = fmaf(gv, xv.w, acc.w);
    }
    st4(red + (rl * c4n + cl) * 4, acc);
  }
  __syncthreads();
  if (rl == 0) {
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < rln; ++i) {
      const float4 v = ld4(red + (i * c4n + cl) * 4);
      tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
    }
    st4(slabs + static_cast<int64_t>(blockIdx.x) * K + 4 * cl, tot);
  }
}

extern "C" int dfm_linear1_supported(int features) {
  return (features > 0 && features % 4 == 0 && features / 4 <= kL1Threads) ? 1 : 0;
}
extern "C" int dfm_linear1_forward(const float* d_x, int64_t batch, int features, const float* d_w, const float* d_b,
                                   float* d_out, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w && d_out, "null argument");
  DFM_REQUIRE(dfm_linear1_supported(features), "features must be a multiple of 4, at most %d", 4 * kL1Threads);
  DFM_REQUIRE(batch >= 0 && batch < (1 << 30), "bad batch");
  DFM_REQUIRE(aligned16(d_x) && aligned16(d_w), "pointers must be 16-byte aligned");
  if (batch == 0) return DFM_OK;
  const int rows = kL1Threads / kWave;
  hipLaunchKernelGGL(linear1_fwd_kernel, dim3(static_cast<unsigned>((batch + rows - 1) / rows)), dim3(kL1Threads), 0,
                     as_stream(stream), d_x, batch, features, d_w, d_b, d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
extern "C" int dfm_linear1_backward_splits(int64_t batch) {
  return static_cast<int>((batch + kL1Rows - 1) / kL1Rows);
}
extern "C" int dfm_linear1_backward(const float* d_g, const float* d_x, int64_t batch, int features,
                                    const float* d_w, float* d_g_x, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_g && d_x && d_w && d_g_x && d_workspace, "null argument");
  DFM_REQUIRE(dfm_linear1_supported(features), "features must be a multiple of 4, at most %d", 4 * kL1Threads);
  DFM_REQUIRE(batch > 0 && batch < (1 << 30), "bad batch");
  DFM_REQUIRE(aligned16(d_x) && aligned16(d_w) && aligned16(d_g_x) && aligned16(d_workspace),
              "pointers must be 16-byte aligned");
  hipLaunchKernelGGL(linear1_bwd_kernel, dim3(static_cast<unsigned>(dfm_linear1_backward_splits(batch))),
                     dim3(kL1Threads), 0, as_stream(stream), d_g, d_x, batch, features, d_w, d_g_x,
                     static_cast<float*>(d_workspace));
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

namespace {
// the batch split actually launched for (out, in, batch): slices per split rounded to whole k-slices
void dw_split_plan(int n_out, int k_in, int m, int* splits, int* k_per_split) {
  const int s0 = dw_splits(n_out, k_in, m);
  *k_per_split = ((m + s0 - 1) / s0 + BK - 1) / BK * BK;
  *splits = (m + *k_per_split - 1) / *k_per_split;
}
}  // namespace

// Arithmetic of the tower's GEMMs: 0 = exact fp32 matrix pipe, 1 = fp32 forward, bf16 x 3 backward
// (gemm_core.h::mainloop_x3, inside dfm_linear_backward), 2 = bf16 x 6 forward and backward (gemm_x6.h: fp32-faithful
// products on the bf16 pipe; the fused steps then call the *_x6 / *_planes entry points below — the library only
// records the choice).  An explicit API call, recorded by bench.py (config.tower_mode) — nothing outside the
// caller's code changes what a run computes.
static int g_tower_mode = 0;
extern "C" int dfm_tower_set_mode(int mode) {
  DFM_REQUIRE(mode >= 0 && mode <= 2, "tower mode %d outside [0, 2]", mode);
  g_tower_mode = mode;
  return DFM_OK;
}
extern "C" int dfm_tower_get_mode(void) { return g_tower_mode; }

extern "C" int dfm_linear_backward_splits(int64_t batch, int out_features, int in_features) {
  int splits, kps;
  dw_split_plan(out_features, in_features, static_cast<int>(batch), &splits, &kps);
  return splits;
}

extern "C" size_t dfm_linear_backward_workspace_bytes(int64_t batch, int out_features, int in_features) {
  const int s = dw_splits(out_features, in_features, batch);
  return align256(sizeof(float) * static_cast<size_t>(s) * out_features * in_features);
}

extern "C" int dfm_linear_backward(const float* d_dz, int64_t batch, int out_features, const float* d_x,
                                   int in_features, const float* d_w, float* d_g_x,
                                   const dfm_bn_bwd* bn_below, const dfm_fm_bwd* fm, int parts, void* d_workspace,
                                   dfm_stream_t stream) {
  DFM_REQUIRE(d_dz && d_x && d_w && d_workspace, "null argument");
  DFM_REQUIRE(parts >= 1 && parts <= 3, "parts: 1 = d weight, 2 = d input, 3 = both");
  DFM_REQUIRE(batch > 0 && batch < (1 << 30) && out_features > 0 && in_features > 0, "bad shape");
  DFM_REQUIRE(!(bn_below && fm), "bn_below and fm are exclusive");
  DFM_REQUIRE(!(parts & 2) || bn_below || d_g_x, "d_g_x is required without bn_below");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  BnBwd dbn = {};
  if (bn_below && (parts & 2)) DFM_REQUIRE(fill_bn(bn_below, &dbn), "incomplete dfm_bn_bwd");
  FmBwd dfm_ = {};
  if (fm && (parts & 2)) {
    DFM_REQUIRE(fm->g_fm || fm->addend, "dfm_fm_bwd: neither an FM gradient nor an addend");
    DFM_REQUIRE(!fm->g_fm || (fm->fm_sum && fm->e && fm->dim > 0 && K % fm->dim == 0), "incomplete dfm_fm_bwd");
    dfm_.g_fm = fm->g_fm; dfm_.fm_sum = fm->fm_sum; dfm_.e = fm->e; dfm_.addend = fm->addend;
    dfm_.dim = fm->dim > 0 ? fm->dim : 1;
  }
  int splits, k_per_split;
  dw_split_plan(N, K, M, &splits, &k_per_split);
  const int dw_tn = tiles(K, BN), dw_t = tiles(N, BM) * dw_tn;
  const int dx_tn = tiles(K, BN), dx_t = tiles(M, BM) * dx_tn;
  float* slabs = static_cast<float*>(d_workspace);
  const bool fast = operand_fast(d_dz, N, false, N, M) && operand_fast(d_x, K, false, K, M) &&
                    operand_fast(d_dz, N, true, M, N) && operand_fast(d_w, K, false, K, N);
  const int dw_blocks = (parts & 1) ? dw_t * splits : 0;
  const dim3 grid(static_cast<unsigned>(dw_blocks + ((parts & 2) ? dx_t : 0)));
#define DFM_LBWD(F, E)                                                                                            \
  hipLaunchKernelGGL((linear_bwd_kernel<F, E>), grid, dim3(kThreads), 0, as_stream(stream), d_dz, d_x, d_w,        \
                     d_g_x, M, N, K, dw_tn, dw_t, splits, dw_blocks, k_per_split, slabs, dx_tn, dbn, dfm_)
  const int epi = bn_below ? 1 : (fm ? 2 : 0);
  // bf16 x 3 backward (dfm_tower_set_mode(1)): 16-byte-regular operands, an out_features extent the 8-wide k groups
  // of the d-input product divide, and batch splits that start on even rows (the strided pieces pair k, k + 1)
  if (g_tower_mode == 1 && fast && N % 8 == 0 && M % 2 == 0 && k_per_split % 2 == 0) {
#define DFM_LBWD3(E)                                                                                              \
  hipLaunchKernelGGL((linear_bwd_kernel<true, E, true>), grid, dim3(kThreads), 0, as_stream(stream), d_dz, d_x,    \
                     d_w, d_g_x, M, N, K, dw_tn, dw_t, splits, dw_blocks, k_per_split, slabs, dx_tn, dbn, dfm_)
    if (epi == 0) DFM_LBWD3(0); else if (epi == 1) DFM_LBWD3(1); else DFM_LBWD3(2);
#undef DFM_LBWD3
  } else if (fast) {
    if (epi == 0) DFM_LBWD(true, 0); else if (epi == 1) DFM_LBWD(true, 1); else DFM_LBWD(true, 2);
  } else {
    if (epi == 0) DFM_LBWD(false, 0); else if (epi == 1) DFM_LBWD(false, 1); else DFM_LBWD(false, 2);
  }
#undef DFM_LBWD
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// =====================================================================================
// bf16 x 6 tower (gemm_x6.h): plane producers, forward and backward GEMM launches
// =====================================================================================
namespace {
constexpr int kMaxSplitJobs = 8;
struct SplitJobs {
  const float* src[kMaxSplitJobs];
  int rows[kMaxSplitJobs], cols[kMaxSplitJobs], tiles_c[kMaxSplitJobs];
  Planes f[kMaxSplitJobs], s[kMaxSplitJobs];
  int first_block[kMaxSplitJobs + 1];
  int count;
};
}  // namespace

// fp32 matrix [rows][cols] -> planes in role F (contraction = columns) and / or role S (contraction = rows);
// a workgroup owns a 32 x 64 tile.  One launch for every weight of the tower.
__global__ __launch_bounds__(kApThreads) void split_planes_kernel(SplitJobs jobs) {
  __shared__ __attribute__((aligned(16))) float tile[kTileRows * kTileStride];
  int e = 0;
  while (e + 1 < jobs.count && static_cast<int>(blockIdx.x) >= jobs.first_block[e + 1]) ++e;
  const int lb = blockIdx.x - jobs.first_block[e];
  const int R = jobs.rows[e], C = jobs.cols[e];
  const int r0 = (lb / jobs.tiles_c[e]) * kTileRows, c0 = (lb % jobs.tiles_c[e]) * kTileCols;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const float* src = jobs.src[e];
#pragma unroll
  for (int i = 0; i < kTileRows / kApRowLanes; ++i) {
    const int m = r0 + rl + i * kApRowLanes, c = c0 + cl * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < R && c < C) v = ld4(src + static_cast<int64_t>(m) * C + c);
    st4(&tile[(rl + i * kApRowLanes) * kTileStride + cl * 4], v);
  }
  __syncthreads();
  emit_planes_from_tile(tile, r0, c0, R, C, jobs.f[e], jobs.s[e]);
}

template <int AK>   // A = x: OP_PLANES (role F planes of a lower layer's output) or OP_F32_KC (fp32, split here)
__global__ __launch_bounds__(kThreads) void linear_bn_fwd_x6_kernel(
    const float* __restrict__ x, int64_t ldx, Planes xp, Planes wp, const float* __restrict__ bias,
    float* __restrict__ z, int M, int N, int K, int tiles_n, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x6_lds[];
  __bf16* smem = reinterpret_cast<__bf16*>(x6_lds);
  const TilePos pos;
  const int lt = xcd_logical_index(blockIdx.x, gridDim.x);
  const int m0 = (lt / tiles_n) * BM, n0 = (lt % tiles_n) * BN;
  f32x16 acc = {};
  const X6Operand<OP_PLANES> ob(wp, n0);
  if constexpr (AK == OP_PLANES) {
    const X6Operand<OP_PLANES> oa(xp, m0);
    mainloop_x6(oa, ob, 0, K, smem, pos, acc);
  } else {
    const X6Operand<OP_F32_KC> oa(x, ldx, m0, M);
    mainloop_x6(oa, ob, 0, K, smem, pos, acc);
  }
  if (pos.khalf == 1) return;
  fwd_tile_epilogue(acc, pos, m0, n0, M, N, bias, z, partial);
}

// backward of one Linear on planes: blocks [0, dw_blocks) d weight tiles x batch splits (A = d z role S, B = x role S
// or fp32 x split here), the rest d input tiles (A = d z role F, B = W role S).
template <int EPI, int XK>
__global__ __launch_bounds__(kThreads) void linear_bwd_x6_kernel(
    Planes dzf, Planes dzs, const float* __restrict__ x, Planes xs, Planes ws, float* __restrict__ g_x, int M, int N,
    int K, int dw_tiles_n, int dw_tiles, int dw_blocks, int k_per_split, float* __restrict__ slabs, int dx_tiles_n,
    BnBwd bn, FmBwd fmb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x6_lds[];
  __bf16* smem = reinterpret_cast<__bf16*>(x6_lds);
  const TilePos pos;
  f32x16 acc = {};
  const int wg = blockIdx.x;
  const int bid = wg < dw_blocks ? xcd_logical_index(wg, dw_blocks)
                                 : dw_blocks + xcd_logical_index_from(wg - dw_blocks, static_cast<int>(gridDim.x) - dw_blocks, dw_blocks);
  if (bid < dw_blocks) {
    const int tile = bid % dw_tiles, sp = bid / dw_tiles;
    const int m0 = (tile / dw_tiles_n) * BM, n0 = (tile % dw_tiles_n) * BN;
    const int kb = sp * k_per_split;
    const int ke = kb + k_per_split < M ? kb + k_per_split : M;
    const X6Operand<OP_PLANES> oa(dzs, m0);
    if constexpr (XK == OP_PLANES) {
      const X6Operand<OP_PLANES> ob(xs, n0);
      mainloop_x6(oa, ob, kb, ke, smem, pos, acc);
    } else {
      const X6Operand<OP_F32_STRIDED> ob(x, K, n0, K);
      mainloop_x6(oa, ob, kb, ke, smem, pos, acc);
    }
    dw_tile_store(acc, pos, m0, n0, N, K, slabs + static_cast<int64_t>(sp) * N * K);
    return;
  }
  const int t = bid - dw_blocks;
  const int m0 = (t / dx_tiles_n) * BM, n0 = (t % dx_tiles_n) * BN;
  const X6Operand<OP_PLANES> oa(dzf, m0);
  const X6Operand<OP_PLANES> ob(ws, n0);
  mainloop_x6(oa, ob, 0, N, smem, pos, acc);
  dx_tile_epilogue<EPI>(acc, pos, m0, n0, M, K, g_x, bn, fmb);
}

namespace {
// batch split of the d-weight product on planes: whole 64-deep slices, about as many per workgroup as a d-input
// workgroup runs (out_features / 64), so that both kinds of workgroup of the launch take the same time
void dw_split_plan_x6(int n_out, int m, int* splits, int* k_per_split) {
  const int target = tiles(n_out, X6_BK) + 1;
  const int slices = tiles(m, X6_BK);
  int per = slices < target ? slices : target;
  *splits = (slices + per - 1) / per;
  per = (slices + *splits - 1) / *splits;
  *k_per_split = per * X6_BK;
  *splits = (slices + per - 1) / per;
}
template <typename K>
int x6_lds_attr(K kernel) {
  DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kX6SmemBytes));
  return DFM_OK;
}
}  // namespace

extern "C" int dfm_tower_x6_supported(int64_t batch, int out_features, int in_features) {
  return batch > 0 && batch < (1 << 30) && batch % 64 == 0 && out_features > 0 && out_features % 8 == 0 &&
         in_features > 0 && in_features % 8 == 0;
}

extern "C" size_t dfm_planes_bytes(int64_t rows, int64_t contraction) {
  if (rows <= 0 || contraction <= 0) return 0;
  return align256(static_cast<size_t>(3) * planes_plane_elems(rows, contraction) * sizeof(__bf16));
}

extern "C" int dfm_split_planes(const dfm_split_job* jobs, int count, dfm_stream_t stream) {
  DFM_REQUIRE(jobs && count >= 1 && count <= kMaxSplitJobs, "1 to %d jobs", kMaxSplitJobs);
  SplitJobs sj = {};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const dfm_split_job& j = jobs[i];
    DFM_REQUIRE(j.src && aligned16(j.src) && j.rows > 0 && j.cols > 0 && j.rows < (1 << 30) && j.cols < (1 << 30) &&
                    j.cols % 4 == 0, "job %d: source must be 16-byte aligned with cols %% 4 == 0", i);
    DFM_REQUIRE(j.planes_f || j.planes_s, "job %d: no destination", i);
    DFM_REQUIRE(!j.planes_f || (aligned16(j.planes_f) && j.cols % 8 == 0), "job %d: role F needs cols %% 8 == 0", i);
    DFM_REQUIRE(!j.planes_s || (aligned16(j.planes_s) && j.rows % 8 == 0), "job %d: role S needs rows %% 8 == 0", i);
    sj.src[i] = j.src;
    sj.rows[i] = static_cast<int>(j.rows); sj.cols[i] = static_cast<int>(j.cols);
    sj.tiles_c[i] = tiles(sj.cols[i], kTileCols);
    sj.f[i] = j.planes_f ? make_planes(j.planes_f, j.rows, j.cols) : Planes{};
    sj.s[i] = j.planes_s ? make_planes(j.planes_s, j.cols, j.rows) : Planes{};
    sj.first_block[i] = blocks;
    blocks += tiles(sj.rows[i], kTileRows) * sj.tiles_c[i];
  }
  sj.first_block[count] = blocks;
  sj.count = count;
  hipLaunchKernelGGL(split_planes_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kApThreads), 0, as_stream(stream), sj);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_linear_bn_forward_x6(const float* d_x, int64_t ldx, const void* d_x_planes_f,
                                        const void* d_w_planes_f, const float* d_bias, int64_t batch,
                                        int out_features, int in_features, float* d_z, void* d_workspace,
                                        dfm_stream_t stream) {
  DFM_REQUIRE((d_x != nullptr) != (d_x_planes_f != nullptr), "exactly one of d_x / d_x_planes_f");
  DFM_REQUIRE(d_w_planes_f && d_z && d_workspace, "null argument");
  DFM_REQUIRE(dfm_tower_x6_supported(batch, out_features, in_features),
              "bf16 x 6 tower: batch %% 64 == 0, features %% 8 == 0");
  DFM_REQUIRE(!d_x || (aligned16(d_x) && ldx % 4 == 0 && ldx >= in_features), "d_x: 16-byte aligned rows");
  DFM_REQUIRE(aligned16(d_w_planes_f) && (!d_x_planes_f || aligned16(d_x_planes_f)), "planes must be 16-byte aligned");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  const int tn = tiles(N, BN);
  const dim3 grid(static_cast<unsigned>(tiles(M, BM) * tn));
  const Planes wp = make_planes(const_cast<void*>(d_w_planes_f), N, K);
  float* partial = static_cast<float*>(d_workspace);
  if (d_x) {
    static const int attr = x6_lds_attr(linear_bn_fwd_x6_kernel<OP_F32_KC>);
    if (attr != DFM_OK) return attr;
    hipLaunchKernelGGL((linear_bn_fwd_x6_kernel<OP_F32_KC>), grid, dim3(kThreads), kX6SmemBytes, as_stream(stream), d_x,
                       ldx, Planes{}, wp, d_bias, d_z, M, N, K, tn, partial);
  } else {
    static const int attr = x6_lds_attr(linear_bn_fwd_x6_kernel<OP_PLANES>);
    if (attr != DFM_OK) return attr;
    hipLaunchKernelGGL((linear_bn_fwd_x6_kernel<OP_PLANES>), grid, dim3(kThreads), kX6SmemBytes, as_stream(stream),
                       nullptr, 0, make_planes(const_cast<void*>(d_x_planes_f), M, K), wp, d_bias, d_z, M, N, K, tn,
                       partial);
  }
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_linear_backward_x6_splits(int64_t batch, int out_features, int in_features) {
  (void)in_features;
  int splits, kps;
  dw_split_plan_x6(out_features, static_cast<int>(batch), &splits, &kps);
  return splits;
}

extern "C" size_t dfm_linear_backward_x6_workspace_bytes(int64_t batch, int out_features, int in_features) {
  const int s = dfm_linear_backward_x6_splits(batch, out_features, in_features);
  return align256(sizeof(float) * static_cast<size_t>(s) * out_features * in_features);
}

extern "C" int dfm_linear_backward_x6(const void* d_dz_planes_f, const void* d_dz_planes_s, int64_t batch,
                                      int out_features, const float* d_x, const void* d_x_planes_s, int in_features,
                                      const void* d_w_planes_s, float* d_g_x, const dfm_bn_bwd* bn_below,
                                      const dfm_fm_bwd* fm, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_dz_planes_f && d_dz_planes_s && d_w_planes_s && d_workspace, "null argument");
  DFM_REQUIRE((d_x != nullptr) != (d_x_planes_s != nullptr), "exactly one of d_x / d_x_planes_s");
  DFM_REQUIRE(dfm_tower_x6_supported(batch, out_features, in_features),
              "bf16 x 6 tower: batch %% 64 == 0, features %% 8 == 0");
  DFM_REQUIRE(!d_x || aligned16(d_x), "d_x must be 16-byte aligned");
  DFM_REQUIRE(!(bn_below && fm), "bn_below and fm are exclusive");
  DFM_REQUIRE(bn_below || d_g_x, "d_g_x is required without bn_below");
  const int M = static_cast<int>(batch), N = out_features, K = in_features;
  BnBwd dbn = {};
  if (bn_below) DFM_REQUIRE(fill_bn(bn_below, &dbn), "incomplete dfm_bn_bwd");
  FmBwd dfm_ = {};
  if (fm) {
    DFM_REQUIRE(fm->g_fm || fm->addend, "dfm_fm_bwd: neither an FM gradient nor an addend");
    DFM_REQUIRE(!fm->g_fm || (fm->fm_sum && fm->e && fm->dim > 0 && K % fm->dim == 0), "incomplete dfm_fm_bwd");
    dfm_.g_fm = fm->g_fm; dfm_.fm_sum = fm->fm_sum; dfm_.e = fm->e; dfm_.addend = fm->addend;
    dfm_.dim = fm->dim > 0 ? fm->dim : 1;
  }
  int splits, k_per_split;
  dw_split_plan_x6(N, M, &splits, &k_per_split);
  const int dw_tn = tiles(K, BN), dw_t = tiles(N, BM) * dw_tn;
  const int dx_tn = tiles(K, BN), dx_t = tiles(M, BM) * dx_tn;
  const int dw_blocks = dw_t * splits;
  const dim3 grid(static_cast<unsigned>(dw_blocks + dx_t));
  const Planes dzf = make_planes(const_cast<void*>(d_dz_planes_f), M, N);
  const Planes dzs = make_planes(const_cast<void*>(d_dz_planes_s), N, M);
  const Planes xs = d_x_planes_s ? make_planes(const_cast<void*>(d_x_planes_s), K, M) : Planes{};
  const Planes ws = make_planes(const_cast<void*>(d_w_planes_s), K, N);
  float* slabs = static_cast<float*>(d_workspace);
#define DFM_LBWD6(E, XK)                                                                                          \
  do {                                                                                                            \
    static const int attr = x6_lds_attr(linear_bwd_x6_kernel<E, XK>);                                             \
    if (attr != DFM_OK) return attr;                                                                              \
    hipLaunchKernelGGL((linear_bwd_x6_kernel<E, XK>), grid, dim3(kThreads), kX6SmemBytes, as_stream(stream), dzf,   \
                       dzs, d_x, xs, ws, d_g_x, M, N, K, dw_tn, dw_t, dw_blocks, k_per_split, slabs, dx_tn, dbn,    \
                       dfm_);                                                                                     \
  } while (0)
  const int epi = bn_below ? 1 : (fm ? 2 : 0);
  if (d_x) {
    if (epi == 0) DFM_LBWD6(0, OP_F32_STRIDED); else if (epi == 1) DFM_LBWD6(1, OP_F32_STRIDED); else DFM_LBWD6(2, OP_F32_STRIDED);
  } else {
    if (epi == 0) DFM_LBWD6(0, OP_PLANES); else if (epi == 1) DFM_LBWD6(1, OP_PLANES); else DFM_LBWD6(2, OP_PLANES);
  }
#undef DFM_LBWD6
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_linear_backward_finish(const dfm_slab_ref* refs, int count, dfm_stream_t stream) {
  DFM_REQUIRE(refs && count > 0 && count <= kMaxSlabRefs, "1..%d slab references", kMaxSlabRefs);
  SlabRefs r = {};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const dfm_slab_ref& h = refs[i];
    DFM_REQUIRE(h.workspace && h.g_w && h.batch > 0 && h.batch < (1 << 30) && h.out_features > 0 && h.in_features > 0,
                "incomplete dfm_slab_ref");
    const int64_t elems = static_cast<int64_t>(h.out_features) * h.in_features;
    DFM_REQUIRE(elems % 4 == 0 && aligned16(h.g_w) && aligned16(h.workspace), "d weight must be float4-addressable");
    int splits, kps;
    dw_split_plan(h.out_features, h.in_features, static_cast<int>(h.batch), &splits, &kps);
    if (h.splits > 0) splits = h.splits;
    r.slabs[i] = static_cast<const float*>(h.workspace);
    r.g[i] = h.g_w;
    r.elems4[i] = static_cast<int>(elems / 4);
    r.splits[i] = splits;
    r.first_block[i] = blocks;
    blocks += (r.elems4[i] + 255) / 256;
  }
  r.first_block[count] = blocks;
  r.count = count;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), r);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
