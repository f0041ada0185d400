// Field self-attention core on the matrix cores: softmax(Q_h K_h^T / sqrt(hd)) V_h and its backward for
// head_dim 16 and up to 48 fields (the Criteo shape: F = 39, 4 heads of 16; reference
// deepfm/models/layers/attention.py:100-112), one wave per (sample, head), exact fp32
// (v_mfma_f32_16x16x4_f32: a k-ordered fp32 fma chain).
//
// Why: the VALU kernels of attention_core.hip keep 39 of 64 lanes busy with scalar-operand FMAs
// (19 of the 64 FLOP/clk a SIMD has) behind scalar-load latency: 81 us forward, 217 us backward for
// 1.6 + 4 GFLOP.  The 39 x 16 . 16 x 39 and 39 x 39 . 39 x 16 products fit 16 x 16 x 4 MFMA tiles
// (F padded to 3 tiles of 16).
//
// Layouts (lane l: c = l & 15, g = l >> 4):
//   MFMA 16x16x4:  A[i = c][k' = g],  B[k' = g][j = c],  D reg r = D[i = 4g + r][j = c].
//   * "row fragment" of a (tokens x 16) matrix X, tile t: the float4 X[16t + c][4g .. 4g+3]; element kc
//     feeds MFMA kc, so the contraction index of MFMA kc at position g is k = 4g + kc — the same
//     permutation on both operands, which a sum over k does not see.  One 16-byte load per tile.
//   * a D tile of Z (rows R, cols C) is, register r at a time, the B operand of a product that
//     contracts over Z's ROW index: X[m][C] = sum_R A[m][R] Z[R][C] with A[i = c][k' = g] = A[m = c][R = 4g + r].
//     So scores are computed TRANSPOSED, ST[key][query] = K Q^T: softmax'ed in place they are the B
//     operand of O^T[d][query] = sum_key V^T[d][key] PT[key][query] — no lane movement, no LDS.
//   * the backward needs P and dS with queries as the contraction index too (dV = P^T dO, dK = dS^T Q):
//     those two are transposed through a wave-private LDS image (row stride 52 floats: conflict-free).
//   Reductions over keys run over a tile's registers and tiles (in-lane) and over the four 16-lane rows:
//   v_permlane16_swap + v_permlane32_swap (VALU), not ds_bpermute.
#include "common.h"

using namespace dfm;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kHd = 16;
constexpr int kUnitsPerBlock = 4;         // waves per workgroup, one (sample, head) each
constexpr int kTS = 52;                   // row stride (floats) of the transpose image: 4 * 52 % 32 == 16

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }

// combine over the four 16-lane rows of the wave (lanes c, c+16, c+32, c+48), result in every lane
__device__ __forceinline__ float rows4_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(fbits(x), fbits(x), false, false);     // rows {0,1} and {2,3}
  const float y = bitsf(a[0]) + bitsf(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(fbits(y), fbits(y), false, false);     // halves
  return bitsf(b[0]) + bitsf(b[1]);
}
__device__ __forceinline__ float rows4_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(fbits(x), fbits(x), false, false);
  const float y = fmaxf(bitsf(a[0]), bitsf(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(fbits(y), fbits(y), false, false);
  return fmaxf(bitsf(b[0]), bitsf(b[1]));
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
// acc += sum over the 16 contraction values held as two row fragments
__device__ __forceinline__ f32x4 mfma_frag(const float4& a, const float4& b, f32x4 acc) {
  acc = mfma4(a.x, b.x, acc);
  acc = mfma4(a.y, b.y, acc);
  acc = mfma4(a.z, b.z, acc);
  acc = mfma4(a.w, b.w, acc);
  return acc;
}

// row fragment of rows [16t, 16t+16) of the matrix at `p` (row stride ld); rows >= F read as `fill` * row F-1
__device__ __forceinline__ float4 row_frag(const float* p, int64_t ld, int t, int c, int g, int F, float scale) {
  const int row = 16 * t + c;
  const int rc = row < F ? row : F - 1;
  const float m = row < F ? scale : 0.f;
  const float4 v = ld4(p + static_cast<int64_t>(rc) * ld + 4 * g);
  return make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
}
// A operand "token on the contraction index": X[16t + 4g + r][c]; rows >= F clamped (times `zero_pad` ? 0 : 1)
__device__ __forceinline__ float col_elem(const float* p, int64_t ld, int t, int r, int c, int g, int F, bool zero_pad) {
  const int row = 16 * t + 4 * g + r;
  const int rc = row < F ? row : F - 1;
  const float v = p[static_cast<int64_t>(rc) * ld + c];
  return (zero_pad && row >= F) ? 0.f : v;
}

// scores (transposed) -> probabilities, in place: pt[tk][tq] reg r = P[query 16tq + c][key 16tk + 4g + r]
template <int NT>
__device__ __forceinline__ void scores_softmax(const float4 (&kf)[NT], const float4 (&qf)[NT], int F, int g,
                                               f32x4 (&pt)[NT][NT]) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      pt[tk][tq] = mfma_frag(kf[tk], qf[tq], acc);
    }
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    float mx = -INFINITY;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * tk + 4 * g + r >= F) pt[tk][tq][r] = -INFINITY;        // keys past the end
        mx = fmaxf(mx, pt[tk][tq][r]);
      }
    mx = rows4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(pt[tk][tq][r] - mx);
        pt[tk][tq][r] = e;
        sum += e;
      }
    const float inv = 1.f / rows4_sum(sum);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[tk][tq][r] *= inv;
  }
}

// Transposes through the wave-private image: lane (c, g) of tile (tk, tq) holds keys 16tk+4g..+3 of query
// 16tq + c -> one 16-byte store at [query][key]; read back one query row per register.
template <int NT>
__device__ __forceinline__ void put_image(float* img, const f32x4 (&m)[NT][NT], int c, int g) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int tq = 0; tq < NT; ++tq)
      st4(img + (16 * tq + c) * kTS + 16 * tk + 4 * g, make_float4(m[tk][tq][0], m[tk][tq][1], m[tk][tq][2], m[tk][tq][3]));
}
// X^T[d][key] = sum_query Y[query][d] M[query][key] for M in the image (Y rows past the end count as 0);
// X[key][:] goes to out[key * ldo + col0 + d].
// frag[tk] = the row fragment X[key 16tk + c][4g ..] that was stored (keys past the end: 0).
template <int NT>
__device__ __forceinline__ void contract_queries(const float* img, const float (&ya)[NT][4], float* __restrict__ out,
                                                 int64_t ldo, int col0, int F, int c, int g, float4 (&frag)[NT]) {
#pragma unroll
  for (int tk = 0; tk < NT; ++tk) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tq = 0; tq < NT; ++tq)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma4(ya[tq][r], img[(16 * tq + 4 * g + r) * kTS + 16 * tk + c], acc);
    const int key = 16 * tk + c;
    frag[tk] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (key < F) st4(out + static_cast<int64_t>(key) * ldo + col0 + 4 * g, frag[tk]);
  }
}
template <int NT>
__device__ __forceinline__ void contract_queries(const float* img, const float (&ya)[NT][4], float* __restrict__ out,
                                                 int64_t ldo, int col0, int F, int c, int g) {
  float4 frag[NT];
  contract_queries<NT>(img, ya, out, ldo, col0, F, c, g, frag);
}

template <int NT>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_mfma_fwd(const float* __restrict__ qkv, int64_t B, int F,
                                                                      int A, int heads, float* __restrict__ o) {
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kUnitsPerBlock + wave;
  if (unit >= B * heads) return;                       // wave-uniform; the kernel has no workgroup barrier
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  const int c = lane & 15, g = lane >> 4;
  const int64_t rs = 3 * static_cast<int64_t>(A);
  const float* base = qkv + b * F * rs + h * kHd;
  const float inv_scale = 0.25f;                       // 1 / sqrt(16)   (attention.py:100-103)
  float4 qf[NT], kf[NT];
  float va[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    qf[t] = row_frag(base, rs, t, c, g, F, inv_scale);
    kf[t] = row_frag(base + A, rs, t, c, g, F, 1.f);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) va[t][r] = col_elem(base + 2 * A, rs, t, r, c, g, F, false);   // P is 0 there
  f32x4 pt[NT][NT];
  scores_softmax<NT>(kf, qf, F, g, pt);
  // O^T[d = 4g + r][query = 16tq + c] = sum_key V[key][d] P[query][key]
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot = mfma4(va[tk][r], pt[tk][tq][r], ot);
    const int query = 16 * tq + c;
    if (query < F) st4(o + (b * F + query) * A + h * kHd + 4 * g, make_float4(ot[0], ot[1], ot[2], ot[3]));
  }
}

// d_qkv (B*F, 3A) from d_o (B*F, A); recomputes the probabilities.
template <int NT>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_mfma_bwd(const float* __restrict__ qkv,
                                                                      const float* __restrict__ d_o, int64_t B, int F,
                                                                      int A, int heads, float* __restrict__ d_qkv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kUnitsPerBlock + wave;
  if (unit >= B * heads) return;                       // wave-uniform; LDS images are wave-private, no barrier
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  const int c = lane & 15, g = lane >> 4;
  const int64_t rs = 3 * static_cast<int64_t>(A);
  const float* base = qkv + b * F * rs + h * kHd;
  const float* gbase = d_o + b * F * A + h * kHd;
  float* dbase = d_qkv + b * F * rs + h * kHd;
  float* img = lds + static_cast<size_t>(wave) * (16 * NT) * kTS;      // [query][key] image, wave-private
  const float inv_scale = 0.25f;
  f32x4 pt[NT][NT], ds[NT][NT];
  {
    float4 qf[NT], kf[NT], vf[NT], gf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      qf[t] = row_frag(base, rs, t, c, g, F, inv_scale);
      kf[t] = row_frag(base + A, rs, t, c, g, F, 1.f);
      vf[t] = row_frag(base + 2 * A, rs, t, c, g, F, 1.f);
      gf[t] = row_frag(gbase, A, t, c, g, F, 1.f);                     // rows >= F are zero: no gradient from padding
    }
    scores_softmax<NT>(kf, qf, F, g, pt);
    // dP^T[key][query] = sum_d V[key][d] dO[query][d]
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int tq = 0; tq < NT; ++tq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        ds[tk][tq] = mfma_frag(vf[tk], gf[tq], acc);
      }
  }
  // dS^T = P^T o (dP^T - sum_key dP o P) / sqrt(hd)
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    float dot = 0.f;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = fmaf(ds[tk][tq][r], pt[tk][tq][r], dot);
    dot = rows4_sum(dot);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[tk][tq][r] = pt[tk][tq][r] * (ds[tk][tq][r] - dot) * inv_scale;
  }
  // ---- dQ^T[d][query] = sum_key K[key][d] dS[query][key]   (contraction over keys: dS^T as it stands) ----
  {
    float ka[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) ka[t][r] = col_elem(base + A, rs, t, r, c, g, F, false);    // dS^T is 0 there
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(ka[tk][r], ds[tk][tq][r], acc);
      const int query = 16 * tq + c;
      if (query < F) st4(dbase + static_cast<int64_t>(query) * rs + 4 * g, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
  }
  // ---- the two products that contract over QUERIES go through the transposed image ----
  float ya[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ya[t][r] = col_elem(gbase, A, t, r, c, g, F, true);
  put_image<NT>(img, pt, c, g);
  contract_queries<NT>(img, ya, dbase, rs, 2 * A, F, c, g);         // dV^T = dO^T P   (dO past the end counts as 0)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ya[t][r] = col_elem(base, rs, t, r, c, g, F, true);
  put_image<NT>(img, ds, c, g);                                     // same wave, LDS in order: the reads above are done
  contract_queries<NT>(img, ya, dbase, rs, A, F, c, g);             // dK^T = Q^T dS   (unscaled Q; dS carries 1/sqrt(hd))
}


// ==================================================================================================
// The same two kernels with the Q | K | V projection INSIDE (attention.py:95-97: Linear(D, A) each):
// x (B*F, D) and the stacked weight (3A, D) + bias (3A) come in, the (B*F, 3A) projection is never
// written or read (123 MB each way at the Criteo shape; the core kernels above are HBM-bound on it).
// Per (sample, head) the head's 16 output features of Q, K, V are 16 x D . D x 16-token products:
//   transposed  (A = W_h, B = x^T): D[d = 4g + r][token = c]  = the ROW FRAGMENT of the projection,
//   direct      (A = x,  B = W_h^T): D[token = 4g + r][d = c] = its "token on the contraction index" form,
// both from the same operand registers (x row fragments, W row fragments), bias as the accumulator's
// start value.  KD = D / 16 sixteen-wide chunks of the input features.
template <int NT, int KD>
struct Proj {
  float4 xf[NT][KD];          // x[token 16t + c][16kd + 4g ..]   (tokens past the end: 0)
  int c, g;
  __device__ __forceinline__ void load_x(const float* __restrict__ x, int64_t row0, int F, int D, int c_, int g_) {
    c = c_; g = g_;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) xf[t][kd] = row_frag(x + row0 * D + 16 * kd, D, t, c, g, F, 1.f);
  }
  // rows [r0, r0 + 16) of the stacked weight: W[r0 + c][16kd + 4g ..]
  __device__ __forceinline__ void load_w(const float* __restrict__ w, int r0, int D, float4 (&wf)[KD]) const {
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) wf[kd] = ld4(w + static_cast<int64_t>(r0 + c) * D + 16 * kd + 4 * g);
  }
  // row fragments of (x W^T + b)[:, r0 : r0 + 16], times `scale`
  __device__ __forceinline__ void transposed(const float4 (&wf)[KD], const float* __restrict__ bias, int r0, float scale,
                                             float4 (&out)[NT]) const {
    const float4 b4 = bias ? ld4(bias + r0 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);      // (no bias: uniform)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 acc = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) acc = mfma_frag(wf[kd], xf[t][kd], acc);
      out[t] = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
    }
  }
  // out[t][r] = (x W^T + b)[token 16t + 4g + r][r0 + c]; tokens past the end: the bias alone, or 0 (zero_pad)
  __device__ __forceinline__ void direct(const float4 (&wf)[KD], const float* __restrict__ bias, int r0, int F,
                                         bool zero_pad, float (&out)[NT][4]) const {
    const float bc = bias ? bias[r0 + c] : 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 acc = {bc, bc, bc, bc};
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) acc = mfma_frag(xf[t][kd], wf[kd], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = (zero_pad && 16 * t + 4 * g + r >= F) ? 0.f : acc[r];
    }
  }
};

template <int NT, int KD>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_qkv_mfma_fwd(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int64_t B, int F, int D,
    int A, int heads, float* __restrict__ o) {
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kUnitsPerBlock + wave;
  if (unit >= B * heads) return;
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  const int c = lane & 15, g = lane >> 4;
  Proj<NT, KD> pj;
  pj.load_x(x, b * F, F, D, c, g);
  float4 wq[KD], wk[KD], wv[KD];
  pj.load_w(w, h * kHd, D, wq);
  pj.load_w(w, A + h * kHd, D, wk);
  pj.load_w(w, 2 * A + h * kHd, D, wv);
  float4 qf[NT], kf[NT];
  float va[NT][4];
  pj.transposed(wq, bias, h * kHd, 0.25f, qf);                  // 1 / sqrt(16)
  pj.transposed(wk, bias, A + h * kHd, 1.f, kf);
  pj.direct(wv, bias, 2 * A + h * kHd, F, false, va);           // P is 0 past the end
  f32x4 pt[NT][NT];
  scores_softmax<NT>(kf, qf, F, g, pt);
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot = mfma4(va[tk][r], pt[tk][tq][r], ot);
    const int query = 16 * tq + c;
    if (query < F) st4(o + (b * F + query) * A + h * kHd + 4 * g, make_float4(ot[0], ot[1], ot[2], ot[3]));
  }
}

// The whole forward of an _AttentionBlock (attention.py:91-120) in ONE launch, for num_heads == 4 (the four
// waves of a workgroup are the four heads of one sample): projection + softmax(QK^T)V as above, then each
// wave multiplies its head's output rows by its 16 columns of W_out (the accumulator of P V is already the
// row fragment that product wants), the four partial (F, D) results meet in LDS, and the workgroup finishes
// y = sum + b_out, and — with the residual — LayerNorm(y + x): mean / rstd per row by a 32- or 64-lane
// reduction.  o (head outputs, for dW_out), y (for the LayerNorm backward) and the statistics are written
// for the backward pass; the separate output GEMM and LayerNorm launches and their re-reads of o and y go.
template <int NT, int KD, bool RES>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_block_mfma_fwd(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ wo, const float* __restrict__ bo, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int64_t B, int F, int A, float* __restrict__ o, float* __restrict__ y,
    float* __restrict__ out, float* __restrict__ stats, int64_t out_group_stride, float* __restrict__ x_copy,
    int64_t x_copy_stride) {
  constexpr int D = 16 * KD, YS = D + 4;                  // row stride of the partial images
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [4 heads][16 NT tokens][YS]
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t b = blockIdx.x;                            // heads == kUnitsPerBlock: one sample per workgroup
  const int h = wave;
  const int c = lane & 15, g = lane >> 4;
  // the last phase's operands (row t = (wave + 4 it) * RPP + sub, element e for this lane), requested now: at
  // the end of a workgroup's life nothing hides their latency (see attn_block_mfma_bwd)
  constexpr int LPT = D <= 32 ? 32 : 64, RPP = 64 / LPT;
  constexpr int NIT = (16 * NT + kUnitsPerBlock * RPP - 1) / (kUnitsPerBlock * RPP);
  constexpr bool kEarly = NIT <= 6;
  const int e = lane % LPT, sub = lane / LPT;
  const bool elive = e < D;
  const float bo_e = elive ? bo[e] : 0.f;
  const float ga = (RES && elive) ? gamma[e] : 0.f, be = (RES && elive) ? beta[e] : 0.f;
  float xe[kEarly ? NIT : 1];
  if (kEarly) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int t = (wave + kUnitsPerBlock * it) * RPP + sub;
      xe[it] = ((RES || x_copy) && elive) ? x[(b * F + (t < F ? t : F - 1)) * D + e] : 0.f;
    }
  }
  {
    Proj<NT, KD> pj;
    pj.load_x(x, b * F, F, D, c, g);
    float4 wq[KD], wk[KD], wv[KD];
    pj.load_w(w, h * kHd, D, wq);
    pj.load_w(w, A + h * kHd, D, wk);
    pj.load_w(w, 2 * A + h * kHd, D, wv);
    float4 qf[NT], kf[NT];
    float va[NT][4];
    pj.transposed(wq, bias, h * kHd, 0.25f, qf);
    pj.transposed(wk, bias, A + h * kHd, 1.f, kf);
    pj.direct(wv, bias, 2 * A + h * kHd, F, false, va);
    f32x4 pt[NT][NT];
    scores_softmax<NT>(kf, qf, F, g, pt);
    float4 wof[KD];                                        // W_out[e = 16 et + c][h * 16 + 4 g ..]
#pragma unroll
    for (int et = 0; et < KD; ++et) wof[et] = ld4(wo + static_cast<int64_t>(16 * et + c) * A + h * kHd + 4 * g);
    float* mine = lds + static_cast<size_t>(wave) * (16 * NT) * YS;
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
      f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot = mfma4(va[tk][r], pt[tk][tq][r], ot);
      const int query = 16 * tq + c;
      const float4 of = make_float4(ot[0], ot[1], ot[2], ot[3]);      // O[query c][4 g ..]: a row fragment
      if (query < F) st4(o + (b * F + query) * A + h * kHd + 4 * g, of);
#pragma unroll
      for (int et = 0; et < KD; ++et) {
        f32x4 yp = {0.f, 0.f, 0.f, 0.f};
        yp = mfma_frag(of, wof[et], yp);                   // yp[r] = partial y[token 16 tq + 4 g + r][e = 16 et + c]
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(16 * tq + 4 * g + r) * YS + 16 * et + c] = yp[r];
      }
    }
  }
  __syncthreads();
  // rows of the sample, round-robin over the waves; LPT lanes per row (32 for D <= 32: two rows per pass)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if ((wave + kUnitsPerBlock * it) * RPP >= F) break;
    const int t = (wave + kUnitsPerBlock * it) * RPP + sub;
    const bool live = elive && t < F;
    const int tc = t < F ? t : F - 1;
    float v = bo_e;
#pragma unroll
    for (int hh = 0; hh < kUnitsPerBlock; ++hh) v += elive ? lds[(static_cast<size_t>(hh) * 16 * NT + tc) * YS + e] : 0.f;
    const int64_t row = b * F + tc;
    if (live) y[row * D + e] = v;
    // x_copy: the block's input rows again, sample b at x_copy + b * x_copy_stride (the DNN of AttentionDeepFM
    // takes cat([attention(e), e]): the second half is written here instead of by a copy kernel)
    const float xv = kEarly ? xe[it] : (((RES || x_copy) && elive) ? x[row * D + e] : 0.f);
    if (x_copy && live) x_copy[b * x_copy_stride + static_cast<int64_t>(tc) * D + e] = xv;
    if (!RES) {
      if (live) out[out_group_stride ? b * out_group_stride + static_cast<int64_t>(tc) * D + e : row * D + e] = v;
      continue;
    }
    const float s = elive ? v + xv : 0.f;
    float mu = group_sum<16>(s);
    mu += __shfl_xor(mu, 16, kWave);
    if (LPT == 64) mu += __shfl_xor(mu, 32, kWave);
    mu /= D;
    const float cz = elive ? s - mu : 0.f;
    float var = group_sum<16>(cz * cz);
    var += __shfl_xor(var, 16, kWave);
    if (LPT == 64) var += __shfl_xor(var, 32, kWave);
    const float rstd = rsqrtf(var / D + eps);
    if (live) {
      out[out_group_stride ? b * out_group_stride + static_cast<int64_t>(tc) * D + e : row * D + e] = cz * rstd * ga + be;
      if (e == 0) { stats[2 * row] = mu; stats[2 * row + 1] = rstd; }
    }
  }
}

// d_qkv (B*F, 3A) = gradient of the projection's output, from d_o; Q, K, V recomputed from x.
template <int NT, int KD>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_qkv_mfma_bwd(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ d_o, int64_t B, int F, int D, int A, int heads, float* __restrict__ d_qkv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kUnitsPerBlock + wave;
  if (unit >= B * heads) return;
  const int64_t b = unit / heads;
  const int h = static_cast<int>(unit % heads);
  const int c = lane & 15, g = lane >> 4;
  const int64_t rs = 3 * static_cast<int64_t>(A);
  const float* gbase = d_o + b * F * A + h * kHd;
  float* dbase = d_qkv + b * F * rs + h * kHd;
  float* img = lds + static_cast<size_t>(wave) * (16 * NT) * kTS;
  Proj<NT, KD> pj;
  pj.load_x(x, b * F, F, D, c, g);
  float4 wq[KD], wk[KD];
  pj.load_w(w, h * kHd, D, wq);
  pj.load_w(w, A + h * kHd, D, wk);
  f32x4 pt[NT][NT], ds[NT][NT];
  {
    float4 qf[NT], kf[NT], vf[NT], gf[NT], wv[KD];
    pj.load_w(w, 2 * A + h * kHd, D, wv);
    pj.transposed(wq, bias, h * kHd, 0.25f, qf);
    pj.transposed(wk, bias, A + h * kHd, 1.f, kf);
    pj.transposed(wv, bias, 2 * A + h * kHd, 1.f, vf);
#pragma unroll
    for (int t = 0; t < NT; ++t) gf[t] = row_frag(gbase, A, t, c, g, F, 1.f);
    scores_softmax<NT>(kf, qf, F, g, pt);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int tq = 0; tq < NT; ++tq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        ds[tk][tq] = mfma_frag(vf[tk], gf[tq], acc);
      }
  }
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    float dot = 0.f;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = fmaf(ds[tk][tq][r], pt[tk][tq][r], dot);
    dot = rows4_sum(dot);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[tk][tq][r] = pt[tk][tq][r] * (ds[tk][tq][r] - dot) * 0.25f;
  }
  float ya[NT][4];
  pj.direct(wk, bias, A + h * kHd, F, false, ya);               // K, token on the contraction index (dS^T is 0 past the end)
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma4(ya[tk][r], ds[tk][tq][r], acc);
    const int query = 16 * tq + c;
    if (query < F) st4(dbase + static_cast<int64_t>(query) * rs + 4 * g, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ya[t][r] = col_elem(gbase, A, t, r, c, g, F, true);
  put_image<NT>(img, pt, c, g);
  contract_queries<NT>(img, ya, dbase, rs, 2 * A, F, c, g);     // dV^T = dO^T P
  pj.direct(wq, bias, h * kHd, F, true, ya);                    // unscaled Q, 0 past the end
  put_image<NT>(img, ds, c, g);
  contract_queries<NT>(img, ya, dbase, rs, A, F, c, g);         // dK^T = Q^T dS
}

// The backward of an _AttentionBlock from d y (the gradient of the output projection's result; with the residual
// it is also the gradient that reaches x directly) to d x, in ONE launch for num_heads == 4 (workgroup = sample,
// wave = head, as in attn_block_mfma_fwd): the head's d O = d y W_out[:, head] is formed in-kernel in both operand
// forms (no (B*F, A) d_o tensor, no GEMM launch), the core backward runs as in attn_qkv_mfma_bwd, and d Q, d K,
// d V — row fragments in the accumulators — are multiplied by the head's 16 rows of W_q, W_k, W_v on the spot:
// the four heads' partial (F, D) results meet in LDS and d x = sum (+ d y) is written once.  d_qkv is still
// written: the weight gradient d W_qkv = d_qkv^T x is a GEMM over the whole batch.
//
// `tail` (optional, for the block that reads the field embeddings themselves): what else flows back into them in
// AttentionDeepFM (attention_deepfm.py:48-66) — the flat half of the DNN's d input and the FM backward
// g_fm * (S - e) (fm.py:18-23; this block's x IS e) — added before the one store, instead of a separate pass.
template <int NT, int KD, bool RES>
__global__ __launch_bounds__(kUnitsPerBlock * 64) void attn_block_mfma_bwd(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ wo, const float* __restrict__ g_y, int64_t B, int F, int A,
    float* __restrict__ d_qkv, float* __restrict__ d_x, AttnGradTail tail) {
  constexpr int D = 16 * KD, YS = D + 4, IS = kTS > YS ? kTS : YS;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [4 heads][16 NT][IS]: transpose image, then d x part
  const int lane = lane_id(), wave = wave_id_uniform();
  const int64_t b = blockIdx.x;
  const int h = wave;
  const int c = lane & 15, g = lane >> 4;
  const int64_t rs = 3 * static_cast<int64_t>(A);
  float* dbase = d_qkv + b * F * rs + h * kHd;
  float* img = lds + static_cast<size_t>(wave) * (16 * NT) * IS;
  Proj<NT, KD> pj;
  pj.load_x(x, b * F, F, D, c, g);
  float4 wq[KD], wk[KD];
  pj.load_w(w, h * kHd, D, wq);
  pj.load_w(w, A + h * kHd, D, wk);
  // What the last phase adds to the heads' sum, requested NOW: with two workgroups per CU nothing hides a
  // memory latency at the end of a workgroup's life (loaded there, these terms cost 20 us of the launch).
  // The last phase gives row t = (wave + 4 it) * RPP + sub, element e to this lane.
  constexpr int LPT = D <= 32 ? 32 : 64, RPP = 64 / LPT;
  constexpr int NIT = (16 * NT + kUnitsPerBlock * RPP - 1) / (kUnitsPerBlock * RPP);
  constexpr bool kEarly = NIT <= 6;                         // (embed_dim 64: 12 rows per lane — too many registers)
  const int e = lane % LPT, sub = lane / LPT;
  const bool elive = e < D;
  float tl_gy[kEarly ? NIT : 1], tl_gf[kEarly ? NIT : 1], tl_x[kEarly ? NIT : 1];
  const float tl_gfm = (tail.g_fm && elive) ? tail.g_fm[b] : 0.f;
  const float tl_s = (tail.g_fm && elive) ? tail.fm_sum[b * D + e] : 0.f;
  if (kEarly) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int t = (wave + kUnitsPerBlock * it) * RPP + sub;
      const int64_t row = b * F + (t < F ? t : F - 1);
      tl_gy[it] = (RES && elive) ? g_y[row * D + e] : 0.f;
      tl_gf[it] = (tail.g_flat && elive) ? tail.g_flat[b * tail.ld_flat + (row - b * F) * D + e] : 0.f;
      tl_x[it] = (tail.g_fm && elive) ? x[row * D + e] : 0.f;
    }
  }
  f32x4 pt[NT][NT], ds[NT][NT];
  float gcol[NT][4];                                        // d O[token 16t + 4g + r][c]
  {
    float4 gf[NT];                                          // d O[token 16t + c][4g ..]
    {
      Proj<NT, KD> pg;
      pg.load_x(g_y, b * F, F, D, c, g);
      float4 wot[KD];                                       // W_out^T[h*16 + c][16kd + 4g ..]
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const float* p = wo + static_cast<int64_t>(16 * kd + 4 * g) * A + h * kHd + c;
        wot[kd] = make_float4(p[0], p[A], p[2 * A], p[3 * static_cast<int64_t>(A)]);
      }
      pg.transposed(wot, nullptr, 0, 1.f, gf);
      pg.direct(wot, nullptr, 0, F, true, gcol);
    }
    float4 qf[NT], kf[NT], vf[NT], wv[KD];
    pj.load_w(w, 2 * A + h * kHd, D, wv);
    pj.transposed(wq, bias, h * kHd, 0.25f, qf);
    pj.transposed(wk, bias, A + h * kHd, 1.f, kf);
    pj.transposed(wv, bias, 2 * A + h * kHd, 1.f, vf);
    scores_softmax<NT>(kf, qf, F, g, pt);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int tq = 0; tq < NT; ++tq) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        ds[tk][tq] = mfma_frag(vf[tk], gf[tq], acc);
      }
  }
#pragma unroll
  for (int tq = 0; tq < NT; ++tq) {
    float dot = 0.f;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = fmaf(ds[tk][tq][r], pt[tk][tq][r], dot);
    dot = rows4_sum(dot);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[tk][tq][r] = pt[tk][tq][r] * (ds[tk][tq][r] - dot) * 0.25f;
  }
  // d x part of this head: yp[t][et][r] = part[token 16t + 4g + r][16et + c], accumulated over Q, K, V
  f32x4 yp[NT][KD];
  // rows [r0, r0 + 16) of the stacked weight as the B operand of (tokens x 16) . (16 x D): W[r0 + 4g + kc][16et + c]
  auto weight_cols = [&](int r0, float4 (&wc)[KD]) {
#pragma unroll
    for (int et = 0; et < KD; ++et) {
      const float* p = w + static_cast<int64_t>(r0 + 4 * g) * D + 16 * et + c;
      wc[et] = make_float4(p[0], p[D], p[2 * D], p[3 * D]);
    }
  };
  float ya[NT][4];
  {
    float4 wc[KD];
    weight_cols(h * kHd, wc);
    pj.direct(wk, bias, A + h * kHd, F, false, ya);             // K, token on the contraction index (dS^T is 0 past the end)
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(ya[tk][r], ds[tk][tq][r], acc);
      const int query = 16 * tq + c;
      const float4 dq = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (query < F) st4(dbase + static_cast<int64_t>(query) * rs + 4 * g, dq);
#pragma unroll
      for (int et = 0; et < KD; ++et) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        yp[tq][et] = mfma_frag(dq, wc[et], z);
      }
    }
  }
  {
    float4 wc[KD], frag[NT];
    weight_cols(2 * A + h * kHd, wc);
    put_image<NT>(img, pt, c, g);
    contract_queries<NT>(img, gcol, dbase, rs, 2 * A, F, c, g, frag);       // dV^T = dO^T P
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int et = 0; et < KD; ++et) yp[t][et] = mfma_frag(frag[t], wc[et], yp[t][et]);
    weight_cols(A + h * kHd, wc);
    pj.direct(wq, bias, h * kHd, F, true, ya);                  // unscaled Q, 0 past the end
    put_image<NT>(img, ds, c, g);
    contract_queries<NT>(img, ya, dbase, rs, A, F, c, g, frag);             // dK^T = Q^T dS
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int et = 0; et < KD; ++et) yp[t][et] = mfma_frag(frag[t], wc[et], yp[t][et]);
  }
  // (the image is this wave's own: its reads above are done before these writes in program order)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int et = 0; et < KD; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(16 * t + 4 * g + r) * IS + 16 * et + c] = yp[t][et][r];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int t = (wave + kUnitsPerBlock * it) * RPP + sub;
    if ((wave + kUnitsPerBlock * it) * RPP >= F) break;
    const bool live = elive && t < F;
    const int tc = t < F ? t : F - 1;
    const int64_t row = b * F + tc;
    float v, gf, xv;
    if (kEarly) {
      v = tl_gy[it]; gf = tl_gf[it]; xv = tl_x[it];
    } else {
      v = (RES && elive) ? g_y[row * D + e] : 0.f;
      gf = (tail.g_flat && elive) ? tail.g_flat[b * tail.ld_flat + static_cast<int64_t>(tc) * D + e] : 0.f;
      xv = (tail.g_fm && elive) ? x[row * D + e] : 0.f;
    }
#pragma unroll
    for (int hh = 0; hh < kUnitsPerBlock; ++hh) v += elive ? lds[(static_cast<size_t>(hh) * 16 * NT + tc) * IS + e] : 0.f;
    v += gf;
    v += tl_gfm * (tl_s - xv);
    if (live) d_x[row * D + e] = v;
  }
}

}  // namespace

namespace dfm {

bool attn_mfma_supported(int F, int A, int heads) {
  return heads > 0 && A % heads == 0 && A / heads == kHd && F >= 1 && F <= 48 && A % 4 == 0;
}

int attn_mfma_forward(const float* qkv, int64_t B, int F, int A, int heads, float* o, hipStream_t st) {
  const int64_t units = B * heads;
  const dim3 grid(static_cast<unsigned>((units + kUnitsPerBlock - 1) / kUnitsPerBlock)), block(kUnitsPerBlock * 64);
  const int nt = (F + 15) / 16;
  if (nt == 1) hipLaunchKernelGGL(attn_mfma_fwd<1>, grid, block, 0, st, qkv, B, F, A, heads, o);
  else if (nt == 2) hipLaunchKernelGGL(attn_mfma_fwd<2>, grid, block, 0, st, qkv, B, F, A, heads, o);
  else hipLaunchKernelGGL(attn_mfma_fwd<3>, grid, block, 0, st, qkv, B, F, A, heads, o);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int attn_mfma_backward(const float* qkv, const float* d_o, int64_t B, int F, int A, int heads, float* d_qkv,
                       hipStream_t st) {
  const int64_t units = B * heads;
  const dim3 grid(static_cast<unsigned>((units + kUnitsPerBlock - 1) / kUnitsPerBlock)), block(kUnitsPerBlock * 64);
  const int nt = (F + 15) / 16;
  const size_t lds = sizeof(float) * kUnitsPerBlock * (16 * nt) * kTS;
  if (nt == 1) hipLaunchKernelGGL(attn_mfma_bwd<1>, grid, block, lds, st, qkv, d_o, B, F, A, heads, d_qkv);
  else if (nt == 2) hipLaunchKernelGGL(attn_mfma_bwd<2>, grid, block, lds, st, qkv, d_o, B, F, A, heads, d_qkv);
  else hipLaunchKernelGGL(attn_mfma_bwd<3>, grid, block, lds, st, qkv, d_o, B, F, A, heads, d_qkv);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

bool attn_qkv_mfma_supported(int F, int D, int A, int heads) {
  return attn_mfma_supported(F, A, heads) && (D == 16 || D == 32 || D == 48 || D == 64);
}

template <int KD>
static int launch_qkv_fwd(const float* x, const float* w, const float* bias, int64_t B, int F, int D, int A, int heads,
                          float* o, hipStream_t st) {
  const int64_t units = B * heads;
  const dim3 grid(static_cast<unsigned>((units + kUnitsPerBlock - 1) / kUnitsPerBlock)), block(kUnitsPerBlock * 64);
  const int nt = (F + 15) / 16;
  if (nt == 1) hipLaunchKernelGGL((attn_qkv_mfma_fwd<1, KD>), grid, block, 0, st, x, w, bias, B, F, D, A, heads, o);
  else if (nt == 2) hipLaunchKernelGGL((attn_qkv_mfma_fwd<2, KD>), grid, block, 0, st, x, w, bias, B, F, D, A, heads, o);
  else hipLaunchKernelGGL((attn_qkv_mfma_fwd<3, KD>), grid, block, 0, st, x, w, bias, B, F, D, A, heads, o);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
template <int KD>
static int launch_qkv_bwd(const float* x, const float* w, const float* bias, const float* d_o, int64_t B, int F, int D,
                          int A, int heads, float* d_qkv, hipStream_t st) {
  const int64_t units = B * heads;
  const dim3 grid(static_cast<unsigned>((units + kUnitsPerBlock - 1) / kUnitsPerBlock)), block(kUnitsPerBlock * 64);
  const int nt = (F + 15) / 16;
  const size_t lds = sizeof(float) * kUnitsPerBlock * (16 * nt) * kTS;
  if (nt == 1) hipLaunchKernelGGL((attn_qkv_mfma_bwd<1, KD>), grid, block, lds, st, x, w, bias, d_o, B, F, D, A, heads, d_qkv);
  else if (nt == 2) hipLaunchKernelGGL((attn_qkv_mfma_bwd<2, KD>), grid, block, lds, st, x, w, bias, d_o, B, F, D, A, heads, d_qkv);
  else hipLaunchKernelGGL((attn_qkv_mfma_bwd<3, KD>), grid, block, lds, st, x, w, bias, d_o, B, F, D, A, heads, d_qkv);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int attn_qkv_mfma_forward(const float* x, const float* w, const float* bias, int64_t B, int F, int D, int A, int heads,
                          float* o, hipStream_t st) {
  switch (D / 16) {
    case 1: return launch_qkv_fwd<1>(x, w, bias, B, F, D, A, heads, o, st);
    case 2: return launch_qkv_fwd<2>(x, w, bias, B, F, D, A, heads, o, st);
    case 3: return launch_qkv_fwd<3>(x, w, bias, B, F, D, A, heads, o, st);
    default: return launch_qkv_fwd<4>(x, w, bias, B, F, D, A, heads, o, st);
  }
}
int attn_qkv_mfma_backward(const float* x, const float* w, const float* bias, const float* d_o, int64_t B, int F, int D,
                           int A, int heads, float* d_qkv, hipStream_t st) {
  switch (D / 16) {
    case 1: return launch_qkv_bwd<1>(x, w, bias, d_o, B, F, D, A, heads, d_qkv, st);
    case 2: return launch_qkv_bwd<2>(x, w, bias, d_o, B, F, D, A, heads, d_qkv, st);
    case 3: return launch_qkv_bwd<3>(x, w, bias, d_o, B, F, D, A, heads, d_qkv, st);
    default: return launch_qkv_bwd<4>(x, w, bias, d_o, B, F, D, A, heads, d_qkv, st);
  }
}

bool attn_block_mfma_supported(int F, int D, int A, int heads) {
  return attn_qkv_mfma_supported(F, D, A, heads) && heads == kUnitsPerBlock;
}

template <int KD, bool RES>
static int launch_block_fwd(const float* x, const float* w, const float* bias, const float* wo, const float* bo,
                            const float* gamma, const float* beta, float eps, int64_t B, int F, int A, float* o,
                            float* y, float* out, float* stats, int64_t out_group_stride, float* x_copy,
                            int64_t x_copy_stride, hipStream_t st) {
  const int nt = (F + 15) / 16;
  const size_t lds = sizeof(float) * kUnitsPerBlock * (16 * nt) * (16 * KD + 4);
  const dim3 grid(static_cast<unsigned>(B)), block(kUnitsPerBlock * 64);
#define DFM_BLK(NT_)                                                                                              \
  hipLaunchKernelGGL((attn_block_mfma_fwd<NT_, KD, RES>), grid, block, lds, st, x, w, bias, wo, bo, gamma, beta, eps, \
                     B, F, A, o, y, out, stats, out_group_stride, x_copy, x_copy_stride)
  if (nt == 1) DFM_BLK(1); else if (nt == 2) DFM_BLK(2); else DFM_BLK(3);
#undef DFM_BLK
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int attn_block_mfma_forward(const float* x, const float* w, const float* bias, const float* wo, const float* bo,
                            const float* gamma, const float* beta, float eps, int64_t B, int F, int D, int A, float* o,
                            float* y, float* out, float* stats, int64_t out_group_stride, float* x_copy,
                            int64_t x_copy_stride, hipStream_t st) {
  const bool res = gamma != nullptr;
#define DFM_KD(K_)                                                                                                  \
  if (D == 16 * K_)                                                                                                 \
    return res ? launch_block_fwd<K_, true>(x, w, bias, wo, bo, gamma, beta, eps, B, F, A, o, y, out, stats, out_group_stride, x_copy, x_copy_stride, st) \
               : launch_block_fwd<K_, false>(x, w, bias, wo, bo, gamma, beta, eps, B, F, A, o, y, out, stats, out_group_stride, x_copy, x_copy_stride, st)
  DFM_KD(1); DFM_KD(2); DFM_KD(3); DFM_KD(4);
#undef DFM_KD
  return fail(DFM_ERR_UNSUPPORTED, "attention block kernel: embed_dim %d", D);
}

template <int KD, bool RES>
static int launch_block_bwd(const float* x, const float* w, const float* bias, const float* wo, const float* g_y,
                            int64_t B, int F, int A, float* d_qkv, float* d_x, const AttnGradTail& tail,
                            hipStream_t st) {
  const int nt = (F + 15) / 16;
  constexpr int IS = kTS > 16 * KD + 4 ? kTS : 16 * KD + 4;
  const size_t lds = sizeof(float) * kUnitsPerBlock * (16 * nt) * IS;
  const dim3 grid(static_cast<unsigned>(B)), block(kUnitsPerBlock * 64);
#define DFM_BLK(NT_)                                                                                              \
  hipLaunchKernelGGL((attn_block_mfma_bwd<NT_, KD, RES>), grid, block, lds, st, x, w, bias, wo, g_y, B, F, A, d_qkv, d_x, tail)
  if (nt == 1) DFM_BLK(1); else if (nt == 2) DFM_BLK(2); else DFM_BLK(3);
#undef DFM_BLK
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int attn_block_mfma_backward(const float* x, const float* w, const float* bias, const float* wo, const float* g_y,
                             bool residual, int64_t B, int F, int D, int A, float* d_qkv, float* d_x,
                             const AttnGradTail& tail, hipStream_t st) {
#define DFM_KD(K_)                                                                                     \
  if (D == 16 * K_)                                                                                    \
    return residual ? launch_block_bwd<K_, true>(x, w, bias, wo, g_y, B, F, A, d_qkv, d_x, tail, st)   \
                    : launch_block_bwd<K_, false>(x, w, bias, wo, g_y, B, F, A, d_qkv, d_x, tail, st)
  DFM_KD(1); DFM_KD(2); DFM_KD(3); DFM_KD(4);
#undef DFM_KD
  return fail(DFM_ERR_UNSUPPORTED, "attention block kernel: embed_dim %d", D);
}

}  // namespace dfm
