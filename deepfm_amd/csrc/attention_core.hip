// Field self-attention, GEMM-structured path (reference deepfm/models/layers/attention.py:91-120).
//
// The projections of an _AttentionBlock are plain GEMMs over the B*F rows of x — Q|K|V = x Wqkv^T
// + b (attention.py:95-97) and out = O Wo^T + bo (:115) — and run on dfm_gemm_f32 (exact fp32
// MFMA).  What is left is small and per sample:
//   * attn_core_fwd / attn_core_bwd: softmax(Q_h K_h^T / sqrt(hd)) V_h for one (sample, head) per
//     wave (:100-112).  F = 39 "tokens": lane i owns query row i; K_h / V_h rows are scalar
//     operands (SGPRs); the (B, heads, F, F) score tensor of the reference never leaves LDS.
//     The backward recomputes P from Q, K.
//   * layernorm_fwd / layernorm_bwd: LayerNorm(out + x) over D (:117-118), one row per lane
//     group, parameter gradients as fixed-order partial sums.
#include "common.h"
#include "partial_reduce.h"

using namespace dfm;

namespace {
constexpr int kMaxF = 64;     // one lane per query row
constexpr int kWavesPerBlock = 4;
}  // namespace

// qkv (B*F, 3A): row = [q (A) | k (A) | v (A)];  o (B*F, A).  One wave per (b, h); lane i owns
// query row i.  The K_h / V_h / Q_h / dO_h rows a lane multiplies against are the SAME for every
// lane of the wave (row j of the head), so they are read with wave-uniform addresses straight from
// global memory — scalar loads into SGPRs through the scalar cache, used as scalar operands of the
// FMAs — instead of LDS broadcasts: an LDS broadcast still moves 64 x 16 B through the LDS pipe, and
// with 6 sweeps of 39 rows per (sample, head) that pipe (shared by the CU's four SIMDs) was the
// bound.  LDS now only holds the score rows (F x F per wave; lane i owns row i — a runtime-indexed
// register array would be demoted to scratch).  (Prefetching row j+1 into a second SGPR set while
// row j is multiplied measured slower: 0.912 vs 0.869 ms forward + backward — SGPR spills and moves.)
template <int HD>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[HD]) {
#pragma unroll
  for (int e = 0; e < HD; ++e) r[e] = p[e];
}
template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const float (&b)[HD]) {
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < HD; ++e) acc = fmaf(a[e], b[e], acc);
  return acc;
}

template <int HD>
__global__ __launch_bounds__(kWavesPerBlock * 64) void attn_core_fwd(const float* __restrict__ qkv, int64_t B,
                                                                       int F, int A, int heads,
                                                                       float* __restrict__ o) {
  extern __shared__ float lds[];
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;   // (b, h): wave-uniform
  if (unit >= B * heads) return;
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  float* Ps = lds + static_cast<size_t>(wave) * F * F;
  const float* base = qkv + b * F * 3 * A + h * HD;          // wave-uniform
  const int64_t rs = 3 * static_cast<int64_t>(A);            // row stride of qkv
  // wave-private LDS: same-wave LDS ops are ordered, no barrier needed
  const bool live = lane < F;
  const int row = live ? lane : 0;
  float* prow = Ps + row * F;
  float q[HD];
  load_row<HD>(base + row * rs, q);
  const float inv_scale = 1.f / sqrtf(static_cast<float>(HD));
  const int last = F - 1;
  constexpr int kGroup = HD <= 16 ? 3 : 2;                   // rows per scalar-load group (SGPR budget)
  // scores of this lane's row; running max
  float mx = -INFINITY;
  // rows in groups of kGroup: the group's scalar loads are issued together (distinct SGPR sets), so a
  // cache-missing row costs one L2 round trip per group, not per row; past-the-end rows of the last
  // group are clamped loads whose results are skipped (uniform branch)
  const float* kp = base + A;                                // K row j0; running pointer (no multiply per row)
  const float* const k_last = base + A + last * rs;
  for (int j0 = 0; j0 < F; j0 += kGroup, kp += kGroup * rs) {
    float kr[kGroup][HD];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) load_row<HD>(j0 + u < F ? kp + u * rs : k_last, kr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
    // (no branch on j0 + u < F: a branch lets the compiler sink the scalar load into it, one round
    // trip per row again.  A past-the-end slot re-does row F - 1: same value, same address.)
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      const int jc = j0 + u < F ? j0 + u : last;
      const float acc = dot_row<HD>(q, kr[u]) * inv_scale;
      if (live) prow[jc] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  float sum = 0.f;
  for (int j = 0; j < F; ++j) {
    const float ev = expf(prow[j] - mx);
    if (live) prow[j] = ev;
    sum += ev;
  }
  const float inv = 1.f / sum;
  float out[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) out[e] = 0.f;
  const float* vp = base + 2 * A;                            // V row j0
  const float* const v_last = base + 2 * A + last * rs;
  for (int j0 = 0; j0 < F; j0 += kGroup, vp += kGroup * rs) {
    float vr[kGroup][HD];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) load_row<HD>(j0 + u < F ? vp + u * rs : v_last, vr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      const bool ok = j0 + u < F;
      const float p = ok ? prow[ok ? j0 + u : last] * inv : 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) out[e] = fmaf(p, vr[u][e], out[e]);
    }
  }
  if (live) {
    float* dst = o + (b * F + lane) * A + h * HD;
#pragma unroll
    for (int e = 0; e < HD; ++e) dst[e] = out[e];
  }
}

// d_qkv (B*F, 3A) from d_o (B*F, A); recomputes P.  LDS per wave: ONE F x F matrix that holds P for
// the dV sweep and is then overwritten by dS for the dK sweep (dP is recomputed instead of stored).
template <int HD>
__global__ __launch_bounds__(kWavesPerBlock * 64) void attn_core_bwd(const float* __restrict__ qkv,
                                                                       const float* __restrict__ d_o, int64_t B,
                                                                       int F, int A, int heads,
                                                                       float* __restrict__ d_qkv) {
  extern __shared__ float lds[];
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (unit >= B * heads) return;
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  float* Ms = lds + static_cast<size_t>(wave) * F * F;       // P, later dS  (row i, col j at i*F + j)
  const float* base = qkv + b * F * 3 * A + h * HD;          // wave-uniform
  const float* gbase = d_o + b * F * A + h * HD;
  const int64_t rs = 3 * static_cast<int64_t>(A);
  const bool live = lane < F;
  const int row = live ? lane : 0;
  const float inv_scale = 1.f / sqrtf(static_cast<float>(HD));
  float q[HD], g[HD];
  load_row<HD>(base + row * rs, q);
  load_row<HD>(gbase + static_cast<int64_t>(row) * A, g);
  float* mrow = Ms + row * F;       // dead lanes shadow row 0 and never write
  // ---- P row of this lane ---------------------------------------------------------------------
  const int last = F - 1;
  constexpr int kGroup = HD <= 16 ? 3 : 2;                   // rows per scalar-load group (see attn_core_fwd)
  constexpr int kGroup2 = HD <= 16 ? 2 : 1;                  // sweeps that load two rows per j
  float mx = -INFINITY;
  const float* kp = base + A;                                // K row j0; running pointer (no multiply per row)
  const float* const k_last = base + A + last * rs;
  for (int j0 = 0; j0 < F; j0 += kGroup, kp += kGroup * rs) {
    float kr[kGroup][HD];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) load_row<HD>(j0 + u < F ? kp + u * rs : k_last, kr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      const int jc = j0 + u < F ? j0 + u : last;           // past the end: row F - 1 again (same value)
      const float acc = dot_row<HD>(q, kr[u]) * inv_scale;
      if (live) mrow[jc] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  float sum = 0.f;
  for (int j = 0; j < F; ++j) {
    const float ev = expf(mrow[j] - mx);
    if (live) mrow[j] = ev;
    sum += ev;
  }
  const float inv = 1.f / sum;
  float dot = 0.f;                  // sum_j dP_ij P_ij
  const float* vp = base + 2 * A;                            // V row j0
  const float* const v_last = base + 2 * A + last * rs;
  for (int j0 = 0; j0 < F; j0 += kGroup, vp += kGroup * rs) {
    float vr[kGroup][HD];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) load_row<HD>(j0 + u < F ? vp + u * rs : v_last, vr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      const bool ok = j0 + u < F;
      const float p = mrow[ok ? j0 + u : 0] * inv;
      const float dp = dot_row<HD>(g, vr[u]);
      if (live && ok) mrow[ok ? j0 + u : 0] = p;
      dot = fmaf(dp, ok ? p : 0.f, dot);
    }
  }
  // ---- dV[j] = sum_i P[i][j] dO[i]   (lane j owns key/value row j) ----------------------------------
  float* dbase = d_qkv + b * F * 3 * A + h * HD;
  {
    float dv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dv[e] = 0.f;
    const float* gp = gbase;                                 // dO row i0
    const float* const g_last = gbase + static_cast<int64_t>(last) * A;
    for (int i0 = 0; i0 < F; i0 += kGroup, gp += kGroup * A) {
      float gr[kGroup][HD];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) load_row<HD>(i0 + u < F ? gp + u * A : g_last, gr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const bool ok = i0 + u < F;
        const float p = ok ? Ms[(ok ? i0 + u : last) * F + row] : 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) dv[e] = fmaf(p, gr[u][e], dv[e]);
      }
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < HD; ++e) dbase[row * rs + 2 * A + e] = dv[e];
    }
  }
  // ---- dS row (overwrites P), dQ -----------------------------------------------------------------
  {
    float dq[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dq[e] = 0.f;
    const float* rp = base + A;                              // K row j0 (V row: + A)
    const float* const r_last = base + A + last * rs;
    for (int j0 = 0; j0 < F; j0 += kGroup2, rp += kGroup2 * rs) {
      float vr[kGroup2][HD], kr[kGroup2][HD];
#pragma unroll
      for (int u = 0; u < kGroup2; ++u) {
        const float* pr = j0 + u < F ? rp + u * rs : r_last;
        load_row<HD>(pr + A, vr[u]);
        load_row<HD>(pr, kr[u]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kGroup2; ++u) {
        const bool ok = j0 + u < F;
        const float dp = dot_row<HD>(g, vr[u]);
        const float ds = ok ? mrow[ok ? j0 + u : 0] * (dp - dot) * inv_scale : 0.f;
        if (live && ok) mrow[ok ? j0 + u : 0] = ds;
#pragma unroll
        for (int e = 0; e < HD; ++e) dq[e] = fmaf(ds, kr[u][e], dq[e]);
      }
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < HD; ++e) dbase[row * rs + e] = dq[e];
    }
  }
  // ---- dK[j] = sum_i dS[i][j] q[i] --------------------------------------------------------------
  {
    float dk[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dk[e] = 0.f;
    const float* qp = base;                                  // Q row i0
    const float* const q_last = base + last * rs;
    for (int i0 = 0; i0 < F; i0 += kGroup, qp += kGroup * rs) {
      float qr[kGroup][HD];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) load_row<HD>(i0 + u < F ? qp + u * rs : q_last, qr[u]);
    __builtin_amdgcn_sched_barrier(0);         // all of the group's loads are issued before any is waited for
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const bool ok = i0 + u < F;
        const float ds = ok ? Ms[(ok ? i0 + u : last) * F + row] : 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) dk[e] = fmaf(ds, qr[u][e], dk[e]);
      }
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < HD; ++e) dbase[row * rs + A + e] = dk[e];
    }
  }
}

// ---- LayerNorm over the last dimension, rows = B*F ---------------------------------------------
// out = LN(y + res) * gamma + beta;  stats[row] = (mean, rstd).
// LPR = pow2 >= D lanes per row (D <= 64): a wave covers 64/LPR consecutive rows with fully
// coalesced loads; the row reductions are xor-shuffles inside the LPR-lane group.
__global__ __launch_bounds__(256) void layernorm_fwd(const float* __restrict__ y, const float* __restrict__ res,
                                                     int64_t rows, int D, int lpr, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     float* __restrict__ out, float* __restrict__ stats) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t r = t / lpr;
  const int d = static_cast<int>(t % lpr);
  const bool live = r < rows && d < D;
  const float v = live ? y[r * D + d] + res[r * D + d] : 0.f;
  float mu = v;
  for (int m = 1; m < lpr; m <<= 1) mu += __shfl_xor(mu, m, kWave);
  mu /= D;
  const float c = live ? v - mu : 0.f;
  float var = c * c;
  for (int m = 1; m < lpr; m <<= 1) var += __shfl_xor(var, m, kWave);
  const float rstd = rsqrtf(var / D + eps);
  if (live) {
    out[r * D + d] = c * rstd * gamma[d] + beta[d];
    if (d == 0) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }
  }
}

// g_s = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)).  A workgroup owns kLnRows
// consecutive rows; lane (sub-row, d) also accumulates d gamma[d] = sum g*xhat and d beta[d] = sum g
// over its rows, combined across sub-rows in a fixed order into partial[block][2][D].
constexpr int kLnRows = 128;
__global__ __launch_bounds__(256) void layernorm_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                     const float* __restrict__ res,
                                                     const float* __restrict__ stats, int64_t rows, int D, int lpr,
                                                     const float* __restrict__ gamma, float* __restrict__ g_s,
                                                     float* __restrict__ partial) {
  __shared__ float red[2][256];
  const int rpp = 256 / lpr;                       // rows per pass
  const int sr = threadIdx.x / lpr, d = threadIdx.x % lpr;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kLnRows;
  const float ga = d < D ? gamma[d] : 0.f;
  float sg = 0.f, sb = 0.f;
  for (int pass = 0; pass < kLnRows / rpp; ++pass) {
    const int64_t r = r0 + pass * rpp + sr;
    const bool live = r < rows && d < D;
    float gv = 0.f, xh = 0.f, rstd = 0.f;
    if (live) {
      rstd = stats[2 * r + 1];
      xh = (y[r * D + d] + res[r * D + d] - stats[2 * r]) * rstd;
      gv = g[r * D + d];
    }
    const float gg = gv * ga;
    float m1 = gg, m2 = gg * xh;
    for (int m = 1; m < lpr; m <<= 1) { m1 += __shfl_xor(m1, m, kWave); m2 += __shfl_xor(m2, m, kWave); }
    m1 /= D; m2 /= D;
    if (live) g_s[r * D + d] = rstd * (gg - m1 - xh * m2);
    sg = fmaf(gv, xh, sg);
    sb += gv;
  }
  red[0][threadIdx.x] = sg;
  red[1][threadIdx.x] = sb;
  __syncthreads();
  if (sr == 0 && d < D) {
    float tg = 0.f, tb = 0.f;
    for (int i = 0; i < rpp; ++i) { tg += red[0][i * lpr + d]; tb += red[1][i * lpr + d]; }
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * D + d] = tg;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * D + d] = tb;
  }
}

// ---- 16-byte variants (D a multiple of 4): a lane owns 4 consecutive d, D/4 (rounded up to a power of
// two) lanes per row.  Same arithmetic per row; the row sums add the lane's four values first.
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

__global__ __launch_bounds__(256) void layernorm_fwd4(const float* __restrict__ y, const float* __restrict__ res,
                                                      int64_t rows, int D, int l4, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      float* __restrict__ out, float* __restrict__ stats,
                                                      int64_t group_rows, int64_t group_stride) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t r = t / l4;
  const int d = static_cast<int>(t % l4) * 4;
  const bool live = r < rows && d < D;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float4 a = ld4(y + r * D + d), b = ld4(res + r * D + d);
    v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  float mu = sum4(v);
  for (int m = 1; m < l4; m <<= 1) mu += __shfl_xor(mu, m, kWave);
  mu /= D;
  const float4 c = live ? make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu) : make_float4(0.f, 0.f, 0.f, 0.f);
  float var = sum4(make_float4(c.x * c.x, c.y * c.y, c.z * c.z, c.w * c.w));
  for (int m = 1; m < l4; m <<= 1) var += __shfl_xor(var, m, kWave);
  const float rstd = rsqrtf(var / D + eps);
  if (live) {
    const float4 ga = ld4(gamma + d), be = ld4(beta + d);
    // group_rows > 0: row r = (sample r / group_rows, field r % group_rows) lands in a buffer whose samples
    // are group_stride floats apart (the attention half of the tower's concatenated input)
    const int64_t o = group_rows > 0 ? (r / group_rows) * group_stride + (r % group_rows) * D : r * D;
    st4(out + o + d, make_float4(c.x * rstd * ga.x + be.x, c.y * rstd * ga.y + be.y, c.z * rstd * ga.z + be.z,
                                 c.w * rstd * ga.w + be.w));
    if (d == 0) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd4(const float* __restrict__ g, const float* __restrict__ y,
                                                      const float* __restrict__ res,
                                                      const float* __restrict__ stats, int64_t rows, int D, int l4,
                                                      const float* __restrict__ gamma, float* __restrict__ g_s,
                                                      float* __restrict__ partial, int64_t group_rows,
                                                      int64_t group_stride) {
  __shared__ float4 red[2][256];
  const int rpp = 256 / l4;                        // rows per pass
  const int sr = threadIdx.x / l4, q = threadIdx.x % l4, d = q * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kLnRows;
  const bool dlive = d < D;
  const float4 ga = dlive ? ld4(gamma + d) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
  for (int pass = 0; pass < kLnRows / rpp; ++pass) {
    const int64_t r = r0 + pass * rpp + sr;
    const bool live = r < rows && dlive;
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), xh = gv;
    float rstd = 0.f;
    if (live) {
      rstd = stats[2 * r + 1];
      const float mu = stats[2 * r];
      const float4 a = ld4(y + r * D + d), b = ld4(res + r * D + d);
      xh = make_float4((a.x + b.x - mu) * rstd, (a.y + b.y - mu) * rstd, (a.z + b.z - mu) * rstd,
                       (a.w + b.w - mu) * rstd);
      gv = ld4(g + (group_rows > 0 ? (r / group_rows) * group_stride + (r % group_rows) * D : r * D) + d);
    }
    const float4 gg = make_float4(gv.x * ga.x, gv.y * ga.y, gv.z * ga.z, gv.w * ga.w);
    float m1 = sum4(gg), m2 = sum4(make_float4(gg.x * xh.x, gg.y * xh.y, gg.z * xh.z, gg.w * xh.w));
    for (int m = 1; m < l4; m <<= 1) { m1 += __shfl_xor(m1, m, kWave); m2 += __shfl_xor(m2, m, kWave); }
    m1 /= D; m2 /= D;
    if (live)
      st4(g_s + r * D + d, make_float4(rstd * (gg.x - m1 - xh.x * m2), rstd * (gg.y - m1 - xh.y * m2),
                                       rstd * (gg.z - m1 - xh.z * m2), rstd * (gg.w - m1 - xh.w * m2)));
    sg.x = fmaf(gv.x, xh.x, sg.x); sg.y = fmaf(gv.y, xh.y, sg.y); sg.z = fmaf(gv.z, xh.z, sg.z); sg.w = fmaf(gv.w, xh.w, sg.w);
    sb.x += gv.x; sb.y += gv.y; sb.z += gv.z; sb.w += gv.w;
  }
  red[0][threadIdx.x] = sg;
  red[1][threadIdx.x] = sb;
  __syncthreads();
  if (sr == 0 && dlive) {
    float4 tg = make_float4(0.f, 0.f, 0.f, 0.f), tb = tg;
    for (int i = 0; i < rpp; ++i) {
      const float4 u = red[0][i * l4 + q], w = red[1][i * l4 + q];
      tg.x += u.x; tg.y += u.y; tg.z += u.z; tg.w += u.w;
      tb.x += w.x; tb.y += w.y; tb.z += w.z; tb.w += w.w;
    }
    st4(partial + (static_cast<int64_t>(blockIdx.x) * 2 + 0) * D + d, tg);
    st4(partial + (static_cast<int64_t>(blockIdx.x) * 2 + 1) * D + d, tb);
  }
}

// d gamma / d beta += column sums of the partial planes.  One workgroup per column d: thread t adds
// partial rows t, t + 256, ... and the 256 sums are combined by a fixed binary tree (the old form,
// two workgroups walking 1 248 rows with 16 threads per column, took 22 us of dependent loads).
__global__ __launch_bounds__(256) void layernorm_bwd_finalize(const float* __restrict__ partial, int blocks, int D,
                                                              float* __restrict__ d_gamma,
                                                              float* __restrict__ d_beta) {
  __shared__ float red[2][256];
  partials::layernorm_finalize_body(blockIdx.x, partial, blocks, D, d_gamma, d_beta, red);
}

namespace {
int check_core(int F, int A, int heads) {
  DFM_REQUIRE(F > 0 && F <= kMaxF, "attention core supports up to %d fields (got %d)", kMaxF, F);
  DFM_REQUIRE(heads > 0 && A % heads == 0, "attention_dim must be divisible by num_heads");
  const int hd = A / heads;
  DFM_REQUIRE(hd == 4 || hd == 8 || hd == 16 || hd == 32, "attention core supports head_dim 4/8/16/32 (got %d)", hd);
  return DFM_OK;
}
}  // namespace

namespace dfm {
bool attn_mfma_supported(int F, int A, int heads);
int attn_mfma_forward(const float* qkv, int64_t B, int F, int A, int heads, float* o, hipStream_t st);
int attn_mfma_backward(const float* qkv, const float* d_o, int64_t B, int F, int A, int heads, float* d_qkv,
                       hipStream_t st);
bool attn_qkv_mfma_supported(int F, int D, int A, int heads);
int attn_qkv_mfma_forward(const float* x, const float* w, const float* bias, int64_t B, int F, int D, int A, int heads,
                          float* o, hipStream_t st);
int attn_qkv_mfma_backward(const float* x, const float* w, const float* bias, const float* d_o, int64_t B, int F, int D,
                           int A, int heads, float* d_qkv, hipStream_t st);
bool attn_block_mfma_supported(int F, int D, int A, int heads);
int attn_block_mfma_forward(const float* x, const float* w, const float* bias, const float* wo, const float* bo,
                            const float* gamma, const float* beta, float eps, int64_t B, int F, int D, int A, float* o,
                            float* y, float* out, float* stats, int64_t out_group_stride, float* x_copy,
                            int64_t x_copy_stride, hipStream_t st);
int attn_block_mfma_backward(const float* x, const float* w, const float* bias, const float* wo, const float* g_y,
                             bool residual, int64_t B, int F, int D, int A, float* d_qkv, float* d_x,
                             const AttnGradTail& tail, hipStream_t st);
}  // namespace dfm

// Attention core with the Q | K | V projection inside (attention_mfma.hip): x (B*F, D), stacked weight
// (3A, D) = [W_q; W_k; W_v], stacked bias (3A).
extern "C" int dfm_attention_qkv_core_supported(int num_fields, int embed_dim, int attention_dim, int num_heads) {
  return attn_qkv_mfma_supported(num_fields, embed_dim, attention_dim, num_heads) ? 1 : 0;
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int dfm_attention_qkv_core_forward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv, int64_t batch,
                                              int num_fields, int embed_dim, int attention_dim, int num_heads,
                                              float* d_o, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w_qkv && d_b_qkv && d_o, "null argument");
  DFM_REQUIRE(attn_qkv_mfma_supported(num_fields, embed_dim, attention_dim, num_heads),
              "unsupported shape (dfm_attention_qkv_core_supported)");
  DFM_REQUIRE(al16(d_x) && al16(d_w_qkv) && al16(d_b_qkv) && al16(d_o), "16-byte aligned buffers only");
  if (batch == 0) return DFM_OK;
  return attn_qkv_mfma_forward(d_x, d_w_qkv, d_b_qkv, batch, num_fields, embed_dim, attention_dim, num_heads, d_o,
                               as_stream(stream));
}

// The whole block forward in one launch (attention_mfma.hip): + W_out, bias, residual LayerNorm.
extern "C" int dfm_attention_block_supported(int num_fields, int embed_dim, int attention_dim, int num_heads) {
  return attn_block_mfma_supported(num_fields, embed_dim, attention_dim, num_heads) ? 1 : 0;
}

extern "C" int dfm_attention_block_forward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv,
                                           const float* d_w_out, const float* d_b_out, const float* d_gamma,
                                           const float* d_beta, float eps, int64_t batch, int num_fields,
                                           int embed_dim, int attention_dim, int num_heads, float* d_o, float* d_y,
                                           float* d_out, float* d_stats, int64_t out_group_stride,
                                           float* d_x_copy, int64_t x_copy_group_stride, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w_qkv && d_b_qkv && d_w_out && d_b_out && d_o && d_y && d_out, "null argument");
  DFM_REQUIRE((d_gamma == nullptr) == (d_beta == nullptr) && (d_gamma == nullptr || d_stats != nullptr),
              "residual LayerNorm needs gamma, beta and the statistics buffer together");
  DFM_REQUIRE(attn_block_mfma_supported(num_fields, embed_dim, attention_dim, num_heads),
              "unsupported shape (dfm_attention_block_supported)");
  DFM_REQUIRE(al16(d_x) && al16(d_w_qkv) && al16(d_b_qkv) && al16(d_w_out) && al16(d_o), "16-byte aligned buffers only");
  DFM_REQUIRE(out_group_stride == 0 || out_group_stride >= static_cast<int64_t>(num_fields) * embed_dim, "bad output grouping");
  DFM_REQUIRE(!d_x_copy || x_copy_group_stride >= static_cast<int64_t>(num_fields) * embed_dim, "bad grouping of the input copy");
  DFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch out of range");
  if (batch == 0) return DFM_OK;
  return attn_block_mfma_forward(d_x, d_w_qkv, d_b_qkv, d_w_out, d_b_out, d_gamma, d_beta, eps, batch, num_fields,
                                 embed_dim, attention_dim, d_o, d_y, d_out, d_stats, out_group_stride, d_x_copy,
                                 x_copy_group_stride, as_stream(stream));
}

extern "C" int dfm_attention_block_backward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv,
                                            const float* d_w_out, const float* d_g_y, int residual, int64_t batch,
                                            int num_fields, int embed_dim, int attention_dim, int num_heads,
                                            float* d_g_qkv, float* d_g_x, const float* d_g_flat, int64_t ld_flat,
                                            const float* d_g_fm, const float* d_fm_sum, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w_qkv && d_b_qkv && d_w_out && d_g_y && d_g_qkv && d_g_x, "null argument");
  DFM_REQUIRE(!d_g_flat || ld_flat >= static_cast<int64_t>(num_fields) * embed_dim, "bad row stride of d_g_flat");
  DFM_REQUIRE((d_g_fm == nullptr) == (d_fm_sum == nullptr), "the FM term needs d_g_fm and d_fm_sum together");
  DFM_REQUIRE(attn_block_mfma_supported(num_fields, embed_dim, attention_dim, num_heads),
              "unsupported shape (dfm_attention_block_supported)");
  DFM_REQUIRE(al16(d_x) && al16(d_w_qkv) && al16(d_b_qkv) && al16(d_g_y) && al16(d_g_qkv), "16-byte aligned buffers only");
  DFM_REQUIRE(d_g_x != d_g_y && d_g_x != d_x, "d_g_x must be its own buffer");
  DFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch out of range");
  if (batch == 0) return DFM_OK;
  return attn_block_mfma_backward(d_x, d_w_qkv, d_b_qkv, d_w_out, d_g_y, residual != 0, batch, num_fields, embed_dim,
                                  attention_dim, d_g_qkv, d_g_x, AttnGradTail{d_g_flat, ld_flat, d_g_fm, d_fm_sum},
                                  as_stream(stream));
}

extern "C" int dfm_attention_qkv_core_backward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv,
                                               const float* d_g_o, int64_t batch, int num_fields, int embed_dim,
                                               int attention_dim, int num_heads, float* d_g_qkv, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_w_qkv && d_b_qkv && d_g_o && d_g_qkv, "null argument");
  DFM_REQUIRE(attn_qkv_mfma_supported(num_fields, embed_dim, attention_dim, num_heads),
              "unsupported shape (dfm_attention_qkv_core_supported)");
  DFM_REQUIRE(al16(d_x) && al16(d_w_qkv) && al16(d_b_qkv) && al16(d_g_o) && al16(d_g_qkv), "16-byte aligned buffers only");
  if (batch == 0) return DFM_OK;
  return attn_qkv_mfma_backward(d_x, d_w_qkv, d_b_qkv, d_g_o, batch, num_fields, embed_dim, attention_dim, num_heads,
                                d_g_qkv, as_stream(stream));
}

extern "C" int dfm_attention_core_supported(int num_fields, int attention_dim, int num_heads) {
  if (num_fields <= 0 || num_fields > kMaxF || num_heads <= 0 || attention_dim % num_heads) return 0;
  const int hd = attention_dim / num_heads;
  return hd == 4 || hd == 8 || hd == 16 || hd == 32;
}

extern "C" int dfm_attention_core_forward(const float* d_qkv, int64_t batch, int num_fields, int attention_dim,
                                          int num_heads, float* d_o, dfm_stream_t stream) {
  DFM_REQUIRE(d_qkv && d_o, "null argument");
  if (int rc = check_core(num_fields, attention_dim, num_heads)) return rc;
  if (batch == 0) return DFM_OK;
  // head_dim 16, <= 48 fields (the Criteo shape): matrix-core kernels (attention_mfma.hip)
  if (attn_mfma_supported(num_fields, attention_dim, num_heads) && (reinterpret_cast<uintptr_t>(d_qkv) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_o) & 15) == 0)
    return attn_mfma_forward(d_qkv, batch, num_fields, attention_dim, num_heads, d_o, as_stream(stream));
  const int hd = attention_dim / num_heads;
  const int64_t units = batch * num_heads;
  const dim3 grid(static_cast<unsigned>((units + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  const size_t lds = sizeof(float) * kWavesPerBlock * (num_fields * num_fields);
  DFM_REQUIRE(lds <= 64 * 1024, "attention core needs %zu bytes of LDS", lds);
  hipStream_t st = as_stream(stream);
#define DFM_CORE(HD) hipLaunchKernelGGL(attn_core_fwd<HD>, grid, block, lds, st, d_qkv, batch, num_fields, attention_dim, num_heads, d_o)
  if (hd == 4) DFM_CORE(4); else if (hd == 8) DFM_CORE(8); else if (hd == 16) DFM_CORE(16); else DFM_CORE(32);
#undef DFM_CORE
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_attention_core_backward(const float* d_qkv, const float* d_g_o, int64_t batch, int num_fields,
                                           int attention_dim, int num_heads, float* d_g_qkv,
                                           dfm_stream_t stream) {
  DFM_REQUIRE(d_qkv && d_g_o && d_g_qkv, "null argument");
  if (int rc = check_core(num_fields, attention_dim, num_heads)) return rc;
  if (batch == 0) return DFM_OK;
  if (attn_mfma_supported(num_fields, attention_dim, num_heads) && (reinterpret_cast<uintptr_t>(d_qkv) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(d_g_o) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_g_qkv) & 15) == 0)
    return attn_mfma_backward(d_qkv, d_g_o, batch, num_fields, attention_dim, num_heads, d_g_qkv, as_stream(stream));
  const int hd = attention_dim / num_heads;
  const int64_t units = batch * num_heads;
  const dim3 grid(static_cast<unsigned>((units + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  const size_t lds = sizeof(float) * kWavesPerBlock * (num_fields * num_fields);
  DFM_REQUIRE(lds <= 160 * 1024, "attention core backward needs %zu bytes of LDS", lds);
  hipStream_t st = as_stream(stream);
#define DFM_CORE(HD)                                                                                        \
  do {                                                                                                      \
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_core_bwd<HD>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));    \
    hipLaunchKernelGGL(attn_core_bwd<HD>, grid, block, lds, st, d_qkv, d_g_o, batch, num_fields,            \
                       attention_dim, num_heads, d_g_qkv);                                                   \
  } while (0)
  if (hd == 4) DFM_CORE(4); else if (hd == 8) DFM_CORE(8); else if (hd == 16) DFM_CORE(16); else DFM_CORE(32);
#undef DFM_CORE
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

namespace {
int ln_lanes(int dim) { int l = 1; while (l < dim) l <<= 1; return l; }
// 16-byte kernels: D a multiple of 4 and every row pointer 16-byte aligned
bool ln_vec4(int dim, const void* a, const void* b, const void* c, const void* d, const void* e) {
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                         reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d) |
                         reinterpret_cast<uintptr_t>(e);
  return dim % 4 == 0 && (bits & 15) == 0;
}
}
extern "C" size_t dfm_layernorm_workspace_bytes(int64_t rows, int dim) {
  return sizeof(float) * 2 * static_cast<size_t>((rows + kLnRows - 1) / kLnRows) * dim;
}

extern "C" int dfm_layernorm_forward(const float* d_y, const float* d_res, int64_t rows, int dim,
                                     const float* d_gamma, const float* d_beta, float eps, float* d_out,
                                     float* d_stats, int64_t out_group_rows, int64_t out_group_stride,
                                     dfm_stream_t stream) {
  DFM_REQUIRE(d_y && d_res && d_gamma && d_beta && d_out && d_stats, "null argument");
  DFM_REQUIRE(rows >= 0 && dim > 0 && dim <= 64, "LayerNorm kernel supports 1 <= dim <= 64");
  DFM_REQUIRE(out_group_rows >= 0 && (out_group_rows == 0 || (out_group_stride >= out_group_rows * dim &&
                                                              out_group_stride % 4 == 0)), "bad output grouping");
  if (rows == 0) return DFM_OK;
  DFM_REQUIRE(out_group_rows == 0 || ln_vec4(dim, d_y, d_res, d_out, d_gamma, d_beta),
              "a grouped output needs dim % 4 == 0 and 16-byte aligned buffers");
  if (ln_vec4(dim, d_y, d_res, d_out, d_gamma, d_beta)) {
    const int l4 = ln_lanes(dim / 4);
    const int64_t threads = rows * l4;
    hipLaunchKernelGGL(layernorm_fwd4, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                       as_stream(stream), d_y, d_res, rows, dim, l4, d_gamma, d_beta, eps, d_out, d_stats, out_group_rows,
                       out_group_stride);
    DFM_LAUNCH_CHECK();
    return DFM_OK;
  }
  const int lpr = ln_lanes(dim);
  const int64_t threads = rows * lpr;
  hipLaunchKernelGGL(layernorm_fwd, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_y, d_res, rows, dim, lpr, d_gamma, d_beta, eps, d_out, d_stats);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_layernorm_backward(const float* d_g_out, const float* d_y, const float* d_res,
                                      const float* d_stats, int64_t rows, int dim, const float* d_gamma,
                                      float* d_g_sum, float* d_g_gamma, float* d_g_beta, void* d_workspace,
                                      int64_t g_group_rows, int64_t g_group_stride, dfm_stream_t stream) {
  DFM_REQUIRE(d_g_out && d_y && d_res && d_stats && d_gamma && d_g_sum && d_workspace, "null argument");
  DFM_REQUIRE((d_g_gamma != nullptr) == (d_g_beta != nullptr), "d_g_gamma and d_g_beta: both or (deferred finish) none");
  DFM_REQUIRE(rows >= 0 && dim > 0 && dim <= 64, "LayerNorm kernel supports 1 <= dim <= 64");
  DFM_REQUIRE(g_group_rows >= 0 && (g_group_rows == 0 || (g_group_stride >= g_group_rows * dim && g_group_stride % 4 == 0)),
              "bad gradient grouping");
  if (rows == 0) return DFM_OK;
  DFM_REQUIRE(g_group_rows == 0 || (ln_vec4(dim, d_y, d_res, d_g_out, d_gamma, d_g_sum) &&
                                    (reinterpret_cast<uintptr_t>(d_workspace) & 15) == 0),
              "a grouped gradient needs dim % 4 == 0 and 16-byte aligned buffers");
  hipStream_t st = as_stream(stream);
  const int blocks = static_cast<int>((rows + kLnRows - 1) / kLnRows);
  float* partial = static_cast<float*>(d_workspace);
  if (ln_vec4(dim, d_y, d_res, d_g_out, d_gamma, d_g_sum) && (reinterpret_cast<uintptr_t>(partial) & 15) == 0)
    hipLaunchKernelGGL(layernorm_bwd4, dim3(blocks), dim3(256), 0, st, d_g_out, d_y, d_res, d_stats, rows, dim,
                       ln_lanes(dim / 4), d_gamma, d_g_sum, partial, g_group_rows, g_group_stride);
  else
    hipLaunchKernelGGL(layernorm_bwd, dim3(blocks), dim3(256), 0, st, d_g_out, d_y, d_res, d_stats, rows, dim,
                       ln_lanes(dim), d_gamma, d_g_sum, partial);
  DFM_LAUNCH_CHECK();
  if (!d_g_gamma) return DFM_OK;        // deferred: dfm_partials_finish (kind 1, blocks = dfm_layernorm_partial_blocks(rows))
  hipLaunchKernelGGL(layernorm_bwd_finalize, dim3(dim), dim3(256), 0, st, partial, blocks, dim,
                     d_g_gamma, d_g_beta);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_layernorm_partial_blocks(int64_t rows) {
  return rows > 0 ? static_cast<int>((rows + kLnRows - 1) / kLnRows) : 0;
}
