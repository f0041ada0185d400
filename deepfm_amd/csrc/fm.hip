// FMInteraction forward / backward (reference deepfm/models/layers/fm.py:18-23).
//   out[b]      = 0.5 * sum_d [ (sum_f e[b,f,d])^2 - sum_f e[b,f,d]^2 ]
//   d e[b,f,d]  = g[b] * (S[b,d] - e[b,f,d]),  S = sum_f e
// HBM-bound: (F*D*4 + 4) bytes per sample forward.  A sample is owned by D/VEC
// consecutive lanes (VEC floats each) that sweep the fields; a wave covers
// 64/(D/VEC) samples, so every wave-instruction touches whole 16-byte pieces.
#include "common.h"

using namespace dfm;

template <int VEC>
struct Pack;
template <>
struct Pack<4> {
  using T = float4;
};
template <>
struct Pack<1> {
  using T = float;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) st4(p, make_float4(v[0], v[1], v[2], v[3]));
  else *p = v[0];
}

// lanes-per-sample LPS = ceil_pow2(D / VEC) (<= 64); lanes with chunk >= D/VEC idle.
template <int VEC>
__global__ __launch_bounds__(256) void fm_fwd_kernel(const float* __restrict__ e, int64_t B, int F,
                                                     int D, int lps, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int chunk = static_cast<int>(t % lps);
  const int64_t b = t / lps;
  const bool live = b < B && chunk * VEC < D;
  float S[VEC] = {}, SQ[VEC] = {};
  if (live) {
    const float* p = e + b * F * D + chunk * VEC;
#pragma unroll 4
    for (int f = 0; f < F; ++f) {
      float v[VEC];
      load_vec<VEC>(p + f * D, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { S[i] += v[i]; SQ[i] = fmaf(v[i], v[i], SQ[i]); }
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc += S[i] * S[i] - SQ[i];
  for (int m = 1; m < lps; m <<= 1) acc += __shfl_xor(acc, m, kWave);
  if (live && chunk == 0) out[b] = 0.5f * acc;
}

template <int VEC>
__global__ __launch_bounds__(256) void fm_bwd_kernel(const float* __restrict__ e,
                                                     const float* __restrict__ g, int64_t B, int F,
                                                     int D, int lps, float* __restrict__ ge) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int chunk = static_cast<int>(t % lps);
  const int64_t b = t / lps;
  if (!(b < B && chunk * VEC < D)) return;
  const float* p = e + b * F * D + chunk * VEC;
  float* o = ge + b * F * D + chunk * VEC;
  float S[VEC] = {};
#pragma unroll 4
  for (int f = 0; f < F; ++f) {
    float v[VEC];
    load_vec<VEC>(p + f * D, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) S[i] += v[i];
  }
  const float gb = g[b];
#pragma unroll 4
  for (int f = 0; f < F; ++f) {
    float v[VEC], r[VEC];
    load_vec<VEC>(p + f * D, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = gb * (S[i] - v[i]);
    store_vec<VEC>(o + f * D, r);
  }
}

// d e = d flat + d (other consumer) + g_fm * (S - e): everything that flows back into field_embeddings of
// AttentionDeepFM (attention_deepfm.py:48-66: FM on the raw embeddings, the attention stack, and the
// flat copy that enters the DNN next to the attended one), one pass, 16 bytes per lane.
__global__ __launch_bounds__(256) void embedding_grad_combine_kernel(
    const float* __restrict__ g_flat, int64_t ld_flat, const float* __restrict__ g_extra,
    const float* __restrict__ g_fm, const float* __restrict__ fm_sum, const float* __restrict__ e, int64_t rows,
    int width, int dim, float* __restrict__ out) {
  const int w4 = width / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= rows * w4) return;
  const int64_t r = t / w4;
  const int c = static_cast<int>(t % w4) * 4;
  float4 v = ld4(g_flat + r * ld_flat + c);
  if (g_extra) {
    const float4 x = ld4(g_extra + r * width + c);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  if (g_fm) {
    const float g = g_fm[r];
    const float4 s4 = ld4(fm_sum + r * dim + c % dim), e4 = ld4(e + r * width + c);
    v.x += g * (s4.x - e4.x); v.y += g * (s4.y - e4.y); v.z += g * (s4.z - e4.z); v.w += g * (s4.w - e4.w);
  }
  st4(out + r * width + c, v);
}

// (rows x width) block copy between row-strided buffers, 16 bytes per lane: slices of the DNN's input /
// d input (attention_deepfm.py:57-61 concatenates two (B, F*D) halves) without torch's generic strided copy.
__global__ __launch_bounds__(256) void copy_2d_kernel(const float* __restrict__ src, int64_t ld_src,
                                                      float* __restrict__ dst, int64_t ld_dst, int64_t rows, int width) {
  const int w4 = width / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= rows * w4) return;
  const int64_t r = t / w4;
  const int c = static_cast<int>(t % w4) * 4;
  st4(dst + r * ld_dst + c, ld4(src + r * ld_src + c));
}

static int pow2_at_least(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

extern "C" {

int dfm_fm_forward(const float* d_field_emb, int64_t batch, int num_fields, int dim, float* d_out,
                   dfm_stream_t stream) {
  DFM_REQUIRE(d_field_emb && d_out, "null argument");
  DFM_REQUIRE(num_fields > 0 && dim > 0 && batch >= 0, "bad shape");
  if (batch == 0) return DFM_OK;
  const int vec = (dim % 4 == 0 && dim / 4 <= 64) ? 4 : 1;
  DFM_REQUIRE(dim / vec <= 64, "dim %d too large for the FM kernel", dim);
  const int lps = pow2_at_least(dim / vec);
  const int64_t threads = batch * lps;
  const dim3 grid(static_cast<unsigned>((threads + 255) / 256)), block(256);
  if (vec == 4)
    hipLaunchKernelGGL(fm_fwd_kernel<4>, grid, block, 0, as_stream(stream), d_field_emb, batch, num_fields, dim, lps, d_out);
  else
    hipLaunchKernelGGL(fm_fwd_kernel<1>, grid, block, 0, as_stream(stream), d_field_emb, batch, num_fields, dim, lps, d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int dfm_fm_backward(const float* d_field_emb, const float* d_g_out, int64_t batch, int num_fields,
                    int dim, float* d_g_field, dfm_stream_t stream) {
  DFM_REQUIRE(d_field_emb && d_g_out && d_g_field, "null argument");
  DFM_REQUIRE(num_fields > 0 && dim > 0 && batch >= 0, "bad shape");
  if (batch == 0) return DFM_OK;
  const int vec = (dim % 4 == 0 && dim / 4 <= 64) ? 4 : 1;
  DFM_REQUIRE(dim / vec <= 64, "dim %d too large for the FM kernel", dim);
  const int lps = pow2_at_least(dim / vec);
  const int64_t threads = batch * lps;
  const dim3 grid(static_cast<unsigned>((threads + 255) / 256)), block(256);
  if (vec == 4)
    hipLaunchKernelGGL(fm_bwd_kernel<4>, grid, block, 0, as_stream(stream), d_field_emb, d_g_out, batch, num_fields, dim, lps, d_g_field);
  else
    hipLaunchKernelGGL(fm_bwd_kernel<1>, grid, block, 0, as_stream(stream), d_field_emb, d_g_out, batch, num_fields, dim, lps, d_g_field);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int dfm_embedding_grad_combine(const float* d_g_flat, int64_t ld_flat, const float* d_g_extra, const float* d_g_fm,
                               const float* d_fm_sum, const float* d_field_emb, int64_t batch, int num_fields, int dim,
                               float* d_g_field, dfm_stream_t stream) {
  DFM_REQUIRE(d_g_flat && d_g_field, "null argument");
  DFM_REQUIRE(num_fields > 0 && dim > 0 && dim % 4 == 0 && batch >= 0, "bad shape (dim must be a multiple of 4)");
  DFM_REQUIRE(!d_g_fm || (d_fm_sum && d_field_emb), "the FM term needs fm_sum and field_embeddings");
  const int width = num_fields * dim;
  DFM_REQUIRE(ld_flat >= width && ld_flat % 4 == 0, "bad leading dimension");
  if (batch == 0) return DFM_OK;
  const int64_t threads = batch * (width / 4);
  hipLaunchKernelGGL(embedding_grad_combine_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                     as_stream(stream), d_g_flat, ld_flat, d_g_extra, d_g_fm, d_fm_sum, d_field_emb, batch, width, dim,
                     d_g_field);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int dfm_copy_2d(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t rows, int width,
                dfm_stream_t stream) {
  DFM_REQUIRE(d_src && d_dst, "null argument");
  DFM_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0 && ld_src >= width && ld_dst >= width && ld_src % 4 == 0 &&
              ld_dst % 4 == 0, "bad shape (width and leading dimensions must be multiples of 4)");
  DFM_REQUIRE((reinterpret_cast<uintptr_t>(d_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0,
              "16-byte aligned buffers only");
  if (rows == 0) return DFM_OK;
  const int64_t threads = rows * (width / 4);
  hipLaunchKernelGGL(copy_2d_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, as_stream(stream),
                     d_src, ld_src, d_dst, ld_dst, rows, width);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

}  // extern "C"
