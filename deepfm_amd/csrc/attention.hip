// Field self-attention block, fused, exact fp32
// (reference deepfm/models/layers/attention.py:67-120, one _AttentionBlock per launch):
//   Q,K,V = x W^T + b  (F,A)            attention.py:95-97
//   P = softmax(Q_h K_h^T / sqrt(hd))   per head, (F,F)   attention.py:100-106
//   O = concat_h(P_h V_h)  (F,A)        attention.py:109-112
//   out = O Wo^T + bo ;  out = LayerNorm(out + x) when use_residual   attention.py:115-118
//
// "Sequence" here is the F feature fields (39 at the Criteo shape): one sample's whole block
// fits in LDS, so a workgroup owns a sample end to end and the (B,heads,F,F) score tensor
// the reference materialises never exists.  Workgroups are persistent over a strided set
// of samples so the weights are staged into LDS once and — in the backward — parameter
// gradients accumulate in LDS in a fixed order and leave as one partial per workgroup.
//
// LDS images are padded by one float per row (row strides A+1 / D+1): every inner loop then
// reads either a broadcast or consecutive banks.
#include "common.h"

using namespace dfm;

namespace {
constexpr int kFwdThreads = 256;
constexpr int kBwdThreads = 512;
constexpr float kLnEps = 1e-5f;

struct AttnParams {
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *ln_w, *ln_b;  // ln_* null without residual
};
struct AttnDims {
  int F, D, A, heads, hd, QS, WS, WOS;
};

__host__ __device__ inline AttnDims make_dims(int F, int D, int A, int heads) {
  AttnDims d;
  d.F = F; d.D = D; d.A = A; d.heads = heads; d.hd = A / heads;
  d.QS = A + 1; d.WS = D + 1; d.WOS = A + 1;
  return d;
}

// weights in LDS: Wq,Wk,Wv (A rows x D, stride WS), Wo (D rows x A, stride WOS), biases
__host__ __device__ inline int weight_floats(const AttnDims& m) {
  return 3 * m.A * m.WS + m.D * m.WOS + 3 * m.A + m.D + 2 * m.D;
}

struct WeightLds {
  float *wq, *wk, *wv, *wo, *bq, *bk, *bv, *bo, *lnw, *lnb;
};

__device__ inline WeightLds carve_weights(float* base, const AttnDims& m) {
  WeightLds w;
  w.wq = base; w.wk = w.wq + m.A * m.WS; w.wv = w.wk + m.A * m.WS; w.wo = w.wv + m.A * m.WS;
  w.bq = w.wo + m.D * m.WOS; w.bk = w.bq + m.A; w.bv = w.bk + m.A; w.bo = w.bv + m.A;
  w.lnw = w.bo + m.D; w.lnb = w.lnw + m.D;
  return w;
}

template <int T>
__device__ inline void stage_weights(const AttnParams& p, const AttnDims& m, const WeightLds& w) {
  const int tid = threadIdx.x;
  for (int i = tid; i < m.A * m.D; i += T) {
    const int a = i / m.D, d = i % m.D;
    w.wq[a * m.WS + d] = p.wq[i];
    w.wk[a * m.WS + d] = p.wk[i];
    w.wv[a * m.WS + d] = p.wv[i];
  }
  for (int i = tid; i < m.D * m.A; i += T) w.wo[(i / m.A) * m.WOS + i % m.A] = p.wo[i];
  for (int i = tid; i < m.A; i += T) { w.bq[i] = p.bq[i]; w.bk[i] = p.bk[i]; w.bv[i] = p.bv[i]; }
  for (int i = tid; i < m.D; i += T) {
    w.bo[i] = p.bo[i];
    w.lnw[i] = p.ln_w ? p.ln_w[i] : 1.f;
    w.lnb[i] = p.ln_b ? p.ln_b[i] : 0.f;
  }
}

// q,k,v (F rows, stride QS) from xs (F x D)
template <int T>
__device__ inline void project_qkv(const AttnDims& m, const WeightLds& w, const float* xs, float* q,
                                   float* k, float* v) {
  for (int o = threadIdx.x; o < 3 * m.F * m.A; o += T) {
    const int which = o / (m.F * m.A), r = o % (m.F * m.A);
    const int f = r / m.A, a = r % m.A;
    const float* W = which == 0 ? w.wq : which == 1 ? w.wk : w.wv;
    float acc = (which == 0 ? w.bq : which == 1 ? w.bk : w.bv)[a];
    const float* xr = xs + f * m.D;
    const float* wr = W + a * m.WS;
    for (int d = 0; d < m.D; ++d) acc = fmaf(xr[d], wr[d], acc);
    (which == 0 ? q : which == 1 ? k : v)[f * m.QS + a] = acc;
  }
}

// P_h (F x F) = softmax(q_h k_h^T / scale), one head
template <int T>
__device__ inline void head_probs(const AttnDims& m, int h, const float* q, const float* k, float* P) {
  const float inv_scale = 1.f / sqrtf(static_cast<float>(m.hd));
  for (int o = threadIdx.x; o < m.F * m.F; o += T) {
    const int i = o / m.F, j = o % m.F;
    const float* qi = q + i * m.QS + h * m.hd;
    const float* kj = k + j * m.QS + h * m.hd;
    float acc = 0.f;
    for (int e = 0; e < m.hd; ++e) acc = fmaf(qi[e], kj[e], acc);
    P[o] = acc * inv_scale;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m.F; i += T) {
    float* row = P + i * m.F;
    float mx = row[0];
    for (int j = 1; j < m.F; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < m.F; ++j) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int j = 0; j < m.F; ++j) row[j] *= inv;
  }
  __syncthreads();
}
}  // namespace

// ---------------------------------------------------------------------------------------
// forward: dynamic LDS = weights + xs(F*D) + q,k,v,o (4*F*QS) + P(F*F) + y(F*D)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFwdThreads) void attn_fwd_kernel(const float* __restrict__ x,
                                                               AttnParams p, int64_t B, int F, int D,
                                                               int A, int heads, int residual,
                                                               float* __restrict__ out) {
  extern __shared__ float lds[];
  const AttnDims m = make_dims(F, D, A, heads);
  const WeightLds w = carve_weights(lds, m);
  float* xs = lds + weight_floats(m);
  float* q = xs + F * D;
  float* k = q + F * m.QS;
  float* v = k + F * m.QS;
  float* o = v + F * m.QS;
  float* P = o + F * m.QS;
  float* y = P + F * F;
  constexpr int T = kFwdThreads;
  const int tid = threadIdx.x;
  stage_weights<T>(p, m, w);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < F * D; i += T) xs[i] = x[b * F * D + i];
    __syncthreads();
    project_qkv<T>(m, w, xs, q, k, v);
    __syncthreads();
    for (int h = 0; h < heads; ++h) {
      head_probs<T>(m, h, q, k, P);
      for (int oidx = tid; oidx < F * m.hd; oidx += T) {
        const int i = oidx / m.hd, e = oidx % m.hd;
        float acc = 0.f;
        for (int j = 0; j < F; ++j) acc = fmaf(P[i * F + j], v[j * m.QS + h * m.hd + e], acc);
        o[i * m.QS + h * m.hd + e] = acc;
      }
      __syncthreads();
    }
    for (int oidx = tid; oidx < F * D; oidx += T) {
      const int i = oidx / D, d = oidx % D;
      float acc = w.bo[d];
      const float* orow = o + i * m.QS;
      const float* wr = w.wo + d * m.WOS;
      for (int a = 0; a < A; ++a) acc = fmaf(orow[a], wr[a], acc);
      y[oidx] = residual ? acc + xs[oidx] : acc;
    }
    __syncthreads();
    if (residual) {
      for (int i = tid; i < F; i += T) {
        float* row = y + i * D;
        float mu = 0.f;
        for (int d = 0; d < D; ++d) mu += row[d];
        mu /= D;
        float var = 0.f;
        for (int d = 0; d < D; ++d) { const float c = row[d] - mu; var = fmaf(c, c, var); }
        const float rstd = rsqrtf(var / D + kLnEps);
        for (int d = 0; d < D; ++d) row[d] = (row[d] - mu) * rstd * w.lnw[d] + w.lnb[d];
      }
      __syncthreads();
    }
    for (int i = tid; i < F * D; i += T) out[b * F * D + i] = y[i];
  }
}

// ---------------------------------------------------------------------------------------
// backward (recomputes the forward per sample).  Parameter-gradient accumulators live in
// LDS across the workgroup's samples; layout of one partial (floats):
//   dWq, dWk, dWv (A*D each) | dWo (D*A) | dbq, dbk, dbv (A each) | dbo (D) | dln_w, dln_b (D each)
// ---------------------------------------------------------------------------------------
namespace {
__host__ __device__ inline int grad_floats(int D, int A) { return 4 * A * D + 3 * A + 3 * D; }
}

__global__ __launch_bounds__(kBwdThreads) void attn_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ g_out, AttnParams p, int64_t B, int F,
    int D, int A, int heads, int residual, float* __restrict__ g_x, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const AttnDims m = make_dims(F, D, A, heads);
  const WeightLds w = carve_weights(lds, m);
  float* acc = lds + weight_floats(m);       // parameter-gradient accumulators
  float* xs = acc + grad_floats(D, A);
  float* q = xs + F * D;
  float* k = q + F * m.QS;
  float* v = k + F * m.QS;
  float* o = v + F * m.QS;                   // O, later dO
  float* P = o + F * m.QS;
  float* dS = P + F * F;
  float* tq = dS + F * F;                    // dQ_h, dK_h temporaries (F*hd each)
  float* tk = tq + F * m.hd;
  float* gl = tk + F * m.hd;                 // gradient entering the output projection (F*D)
  float* xh = gl + F * D;                    // LayerNorm xhat (F*D)
  float* gx = xh + F * D;                    // gradient w.r.t. x (F*D)
  float* rs = gx + F * D;                    // rstd per row (F)
  constexpr int T = kBwdThreads;
  const int tid = threadIdx.x;
  float* dWq = acc; float* dWk = dWq + A * D; float* dWv = dWk + A * D; float* dWo = dWv + A * D;
  float* dbq = dWo + D * A; float* dbk = dbq + A; float* dbv = dbk + A; float* dbo = dbv + A;
  float* dlw = dbo + D; float* dlb = dlw + D;
  stage_weights<T>(p, m, w);
  for (int i = tid; i < grad_floats(D, A); i += T) acc[i] = 0.f;
  const float inv_scale = 1.f / sqrtf(static_cast<float>(m.hd));

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < F * D; i += T) { xs[i] = x[b * F * D + i]; gl[i] = g_out[b * F * D + i]; }
    __syncthreads();
    // ---- recompute forward: q,k,v, O ------------------------------------------------------
    project_qkv<T>(m, w, xs, q, k, v);
    __syncthreads();
    for (int h = 0; h < heads; ++h) {
      head_probs<T>(m, h, q, k, P);
      for (int oidx = tid; oidx < F * m.hd; oidx += T) {
        const int i = oidx / m.hd, e = oidx % m.hd;
        float s = 0.f;
        for (int j = 0; j < F; ++j) s = fmaf(P[i * F + j], v[j * m.QS + h * m.hd + e], s);
        o[i * m.QS + h * m.hd + e] = s;
      }
      __syncthreads();
    }
    // ---- LayerNorm backward (residual) ---------------------------------------------------
    if (residual) {
      for (int oidx = tid; oidx < F * D; oidx += T) {   // y = O Wo^T + bo + x  -> xh (temporarily y)
        const int i = oidx / D, d = oidx % D;
        float s = w.bo[d];
        const float* orow = o + i * m.QS;
        const float* wr = w.wo + d * m.WOS;
        for (int a = 0; a < A; ++a) s = fmaf(orow[a], wr[a], s);
        xh[oidx] = s + xs[oidx];
      }
      __syncthreads();
      for (int i = tid; i < F; i += T) {
        float* row = xh + i * D;
        float mu = 0.f;
        for (int d = 0; d < D; ++d) mu += row[d];
        mu /= D;
        float var = 0.f;
        for (int d = 0; d < D; ++d) { const float c = row[d] - mu; var = fmaf(c, c, var); }
        const float rstd = rsqrtf(var / D + kLnEps);
        rs[i] = rstd;
        for (int d = 0; d < D; ++d) row[d] = (row[d] - mu) * rstd;
      }
      __syncthreads();
      for (int d = tid; d < D; d += T) {                 // d gamma / d beta, rows in order
        float sw = 0.f, sb = 0.f;
        for (int i = 0; i < F; ++i) { const float g = gl[i * D + d]; sw = fmaf(g, xh[i * D + d], sw); sb += g; }
        dlw[d] += sw;
        dlb[d] += sb;
      }
      __syncthreads();
      for (int i = tid; i < F; i += T) {
        float m1 = 0.f, m2 = 0.f;
        for (int d = 0; d < D; ++d) {
          const float gg = gl[i * D + d] * w.lnw[d];
          m1 += gg;
          m2 = fmaf(gg, xh[i * D + d], m2);
        }
        m1 /= D; m2 /= D;
        for (int d = 0; d < D; ++d) {
          const float gg = gl[i * D + d] * w.lnw[d];
          const float gy = rs[i] * (gg - m1 - xh[i * D + d] * m2);
          gl[i * D + d] = gy;       // gradient into the output projection
          gx[i * D + d] = gy;       // and through the residual branch
        }
      }
    } else {
      for (int i = tid; i < F * D; i += T) gx[i] = 0.f;
    }
    __syncthreads();
    // ---- output projection backward --------------------------------------------------------
    for (int oidx = tid; oidx < D * A; oidx += T) {      // dWo[d][a] += sum_i gl[i][d] O[i][a]
      const int d = oidx / A, a = oidx % A;
      float s = 0.f;
      for (int i = 0; i < F; ++i) s = fmaf(gl[i * D + d], o[i * m.QS + a], s);
      dWo[oidx] += s;
    }
    for (int d = tid; d < D; d += T) {
      float s = 0.f;
      for (int i = 0; i < F; ++i) s += gl[i * D + d];
      dbo[d] += s;
    }
    __syncthreads();
    // dO overwrites O: dWo above was the last reader of O (barrier in between), and dO reads
    // only gl and Wo, so each thread may overwrite its own element directly
    for (int oidx = tid; oidx < F * A; oidx += T) {
      const int i = oidx / A, a = oidx % A;
      float s = 0.f;
      for (int d = 0; d < D; ++d) s = fmaf(gl[i * D + d], w.wo[d * m.WOS + a], s);
      o[i * m.QS + a] = s;
    }
    __syncthreads();
    // ---- attention backward, head by head ------------------------------------------------
    for (int h = 0; h < heads; ++h) {
      head_probs<T>(m, h, q, k, P);
      // dP[i][j] = sum_e dO[i][he] v[j][he]
      for (int oidx = tid; oidx < F * F; oidx += T) {
        const int i = oidx / F, j = oidx % F;
        float s = 0.f;
        for (int e = 0; e < m.hd; ++e) s = fmaf(o[i * m.QS + h * m.hd + e], v[j * m.QS + h * m.hd + e], s);
        dS[oidx] = s;
      }
      __syncthreads();
      // dV_h[j][e] = sum_i P[i][j] dO[i][he]  -> overwrites v's head slice (v_h no longer needed)
      for (int oidx = tid; oidx < F * m.hd; oidx += T) {
        const int j = oidx / m.hd, e = oidx % m.hd;
        float s = 0.f;
        for (int i = 0; i < F; ++i) s = fmaf(P[i * F + j], o[i * m.QS + h * m.hd + e], s);
        v[j * m.QS + h * m.hd + e] = s;
      }
      // dS = P * (dP - rowsum(dP * P)) / scale
      for (int i = tid; i < F; i += T) {
        float dot = 0.f;
        for (int j = 0; j < F; ++j) dot = fmaf(dS[i * F + j], P[i * F + j], dot);
        for (int j = 0; j < F; ++j) dS[i * F + j] = P[i * F + j] * (dS[i * F + j] - dot) * inv_scale;
      }
      __syncthreads();
      for (int oidx = tid; oidx < F * m.hd; oidx += T) {
        const int i = oidx / m.hd, e = oidx % m.hd;
        float sq = 0.f, sk = 0.f;
        for (int j = 0; j < F; ++j) {
          sq = fmaf(dS[i * F + j], k[j * m.QS + h * m.hd + e], sq);   // dQ[i] = sum_j dS[i][j] k[j]
          sk = fmaf(dS[j * F + i], q[j * m.QS + h * m.hd + e], sk);   // dK[i] = sum_j dS[j][i] q[j]
        }
        tq[oidx] = sq;
        tk[oidx] = sk;
      }
      __syncthreads();
      for (int oidx = tid; oidx < F * m.hd; oidx += T) {
        const int i = oidx / m.hd, e = oidx % m.hd;
        q[i * m.QS + h * m.hd + e] = tq[oidx];
        k[i * m.QS + h * m.hd + e] = tk[oidx];
      }
      __syncthreads();
    }
    // q,k,v now hold dQ,dK,dV
    // ---- projection backward ---------------------------------------------------------------
    for (int oidx = tid; oidx < 3 * A * D; oidx += T) {   // dW[a][d] += sum_i dT[i][a] x[i][d]
      const int which = oidx / (A * D), r = oidx % (A * D);
      const int a = r / D, d = r % D;
      const float* dT = which == 0 ? q : which == 1 ? k : v;
      float s = 0.f;
      for (int i = 0; i < F; ++i) s = fmaf(dT[i * m.QS + a], xs[i * D + d], s);
      (which == 0 ? dWq : which == 1 ? dWk : dWv)[r] += s;
    }
    for (int oidx = tid; oidx < 3 * A; oidx += T) {
      const int which = oidx / A, a = oidx % A;
      const float* dT = which == 0 ? q : which == 1 ? k : v;
      float s = 0.f;
      for (int i = 0; i < F; ++i) s += dT[i * m.QS + a];
      (which == 0 ? dbq : which == 1 ? dbk : dbv)[a] += s;
    }
    for (int oidx = tid; oidx < F * D; oidx += T) {       // d x += dQ Wq + dK Wk + dV Wv
      const int i = oidx / D, d = oidx % D;
      float s = gx[oidx];
      for (int a = 0; a < A; ++a) {
        s = fmaf(q[i * m.QS + a], w.wq[a * m.WS + d], s);
        s = fmaf(k[i * m.QS + a], w.wk[a * m.WS + d], s);
        s = fmaf(v[i * m.QS + a], w.wv[a * m.WS + d], s);
      }
      g_x[b * F * D + oidx] = s;
    }
  }
  __syncthreads();
  float* mine = partial + static_cast<int64_t>(blockIdx.x) * grad_floats(D, A);
  for (int i = tid; i < grad_floats(D, A); i += T) mine[i] = acc[i];
}

// grads[i] += sum_blocks partial[block][i]  (fixed order)
__global__ __launch_bounds__(256) void attn_reduce_partials(const float* __restrict__ partial, int n,
                                                            int blocks, AttnParams g, int D, int A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[static_cast<int64_t>(b) * n + i];
  const int AD = A * D;
  float* dst;
  int off = i;
  if (off < AD) dst = const_cast<float*>(g.wq);
  else if ((off -= AD) < AD) dst = const_cast<float*>(g.wk);
  else if ((off -= AD) < AD) dst = const_cast<float*>(g.wv);
  else if ((off -= AD) < AD) dst = const_cast<float*>(g.wo);
  else if ((off -= AD) < A) dst = const_cast<float*>(g.bq);
  else if ((off -= A) < A) dst = const_cast<float*>(g.bk);
  else if ((off -= A) < A) dst = const_cast<float*>(g.bv);
  else if ((off -= A) < D) dst = const_cast<float*>(g.bo);
  else if ((off -= D) < D) dst = const_cast<float*>(g.ln_w);
  else { off -= D; dst = const_cast<float*>(g.ln_b); }
  if (dst) dst[off] += s;
}

namespace {
int fwd_lds_floats(const AttnDims& m) { return weight_floats(m) + 2 * m.F * m.D + 4 * m.F * m.QS + m.F * m.F; }
int bwd_lds_floats(const AttnDims& m) {
  return weight_floats(m) + grad_floats(m.D, m.A) + 4 * m.F * m.D + 4 * m.F * m.QS + 2 * m.F * m.F +
         2 * m.F * m.hd + m.F;
}
int grid_blocks(int64_t B) { return static_cast<int>(B < 512 ? B : 512); }

int check_shape(int F, int D, int A, int heads) {
  DFM_REQUIRE(F > 0 && D > 0 && A > 0 && heads > 0, "bad attention shape");
  DFM_REQUIRE(A % heads == 0, "attention_dim (%d) must be divisible by num_heads (%d)", A, heads);
  return DFM_OK;
}
AttnParams make_params(const float* const* p, int residual) {
  AttnParams a;
  a.wq = p[0]; a.bq = p[1]; a.wk = p[2]; a.bk = p[3]; a.wv = p[4]; a.bv = p[5]; a.wo = p[6]; a.bo = p[7];
  a.ln_w = residual ? p[8] : nullptr;
  a.ln_b = residual ? p[9] : nullptr;
  return a;
}
}  // namespace

extern "C" size_t dfm_attention_backward_workspace_bytes(int64_t batch, int embed_dim, int attention_dim) {
  return sizeof(float) * static_cast<size_t>(grid_blocks(batch)) * grad_floats(embed_dim, attention_dim);
}

extern "C" int dfm_attention_forward(const float* d_x, int64_t batch, int num_fields, int embed_dim,
                                     int attention_dim, int num_heads, int use_residual,
                                     const float* const* params, float* d_out, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && params && d_out, "null argument");
  if (int rc = check_shape(num_fields, embed_dim, attention_dim, num_heads)) return rc;
  for (int i = 0; i < (use_residual ? 10 : 8); ++i) DFM_REQUIRE(params[i], "attention parameter %d is null", i);
  if (batch == 0) return DFM_OK;
  const AttnDims m = make_dims(num_fields, embed_dim, attention_dim, num_heads);
  const size_t lds = sizeof(float) * fwd_lds_floats(m);
  DFM_REQUIRE(lds <= 160 * 1024, "attention block needs %zu bytes of LDS (> 160 KiB)", lds);
  DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid_blocks(batch)), dim3(kFwdThreads), lds, as_stream(stream),
                     d_x, make_params(params, use_residual), batch, num_fields, embed_dim, attention_dim,
                     num_heads, use_residual, d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_attention_backward(const float* d_x, const float* d_g_out, int64_t batch,
                                      int num_fields, int embed_dim, int attention_dim, int num_heads,
                                      int use_residual, const float* const* params, float* d_g_x,
                                      float* const* g_params, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_g_out && params && d_g_x && g_params && d_workspace, "null argument");
  if (int rc = check_shape(num_fields, embed_dim, attention_dim, num_heads)) return rc;
  const int np = use_residual ? 10 : 8;
  for (int i = 0; i < np; ++i) DFM_REQUIRE(params[i] && g_params[i], "attention parameter %d is null", i);
  if (batch == 0) return DFM_OK;
  const AttnDims m = make_dims(num_fields, embed_dim, attention_dim, num_heads);
  const size_t lds = sizeof(float) * bwd_lds_floats(m);
  DFM_REQUIRE(lds <= 160 * 1024, "attention backward needs %zu bytes of LDS (> 160 KiB)", lds);
  DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  const int blocks = grid_blocks(batch);
  float* partial = static_cast<float*>(d_workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(blocks), dim3(kBwdThreads), lds, st, d_x, d_g_out,
                     make_params(params, use_residual), batch, num_fields, embed_dim, attention_dim,
                     num_heads, use_residual, d_g_x, partial);
  DFM_LAUNCH_CHECK();
  AttnParams g;
  const float* gp[10] = {};
  for (int i = 0; i < np; ++i) gp[i] = g_params[i];
  g = make_params(gp, use_residual);
  const int n = grad_floats(embed_dim, attention_dim);
  hipLaunchKernelGGL(attn_reduce_partials, dim3((n + 255) / 256), dim3(256), 0, st, partial, n, blocks, g,
                     embed_dim, attention_dim);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
