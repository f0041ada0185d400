// Row-wise Adam over the rows a batch touched (train-step tail of the reference:
// L2 term base.py:78-83 / trainer.py:224-225, clip_grad_norm_ trainer.py:232-235,
// Adam trainer.py:67-70,237 — restricted to touched rows; DESIGN.md states the delta).
//
// Input: L = ranks x chunks lists per SPARSE field, each a sorted set of distinct ids
// with one gradient row per id (rowplan.hip).  A row that appears in several lists is
// OWNED by the first list holding it; the owner adds the other lists' rows in list
// order, so every data-parallel replica computes bit-identical sums from the same
// all-gathered lists, with no atomics.
//   pass A (merge):  g = grad_scale * sum_lists(row) + 2*l2*w ; store g; accumulate |g|^2
//   pass B (apply):  g *= clip ; m,v,w <- Adam
#include "common.h"

using namespace dfm;

namespace {
constexpr int CH = DFM_ROWPLAN_CHUNK;
struct TableArgs {
  dfm_table t[DFM_MAX_FIELDS];
};

__device__ __forceinline__ int find_row(const int32_t* __restrict__ rows, int n, int32_t row) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = rows[mid];
    if (v < row) lo = mid + 1; else hi = mid;
  }
  return (lo < n && rows[lo] == row) ? lo : -1;
}
}  // namespace

__global__ __launch_bounds__(256) void rowadam_merge_kernel(
    TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, float* __restrict__ row_g2, float* __restrict__ row_g1,
    int32_t* __restrict__ owner_flag, float grad_scale, float l2, float* __restrict__ partial) {
  const int lpr = D / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int q = static_cast<int>(t % lpr);
  const int64_t entry = t / lpr;
  const int64_t list = entry / CH;  // l * S + s
  const int u = static_cast<int>(entry % CH);
  float sq = 0.f;
  if (list < static_cast<int64_t>(L) * S && u < num_uniq[list]) {
    const int l = static_cast<int>(list / S), s = static_cast<int>(list % S);
    const int32_t row = uniq_rows[list * CH + u];
    bool owner = true;
    for (int lp = 0; lp < l && owner; ++lp) {
      const int64_t other = static_cast<int64_t>(lp) * S + s;
      if (find_row(uniq_rows + other * CH, num_uniq[other], row) >= 0) owner = false;
    }
    if (q == 0) owner_flag[list * CH + u] = owner ? 1 : 0;
    if (owner) {
      float4 g = ld4(row_g2 + (list * CH + u) * D + q * 4);
      float g1 = q == 0 ? row_g1[list * CH + u] : 0.f;
      for (int ln = l + 1; ln < L; ++ln) {
        const int64_t other = static_cast<int64_t>(ln) * S + s;
        const int pos = find_row(uniq_rows + other * CH, num_uniq[other], row);
        if (pos >= 0) {
          const float4 o = ld4(row_g2 + (other * CH + pos) * D + q * 4);
          g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
          if (q == 0) g1 += row_g1[other * CH + pos];
        }
      }
      const dfm_table tb = tabs.t[s];
      const float4 w = ld4(tb.w2 + static_cast<int64_t>(row) * tb.stride2 + q * 4);
      const float k = 2.f * l2;
      g.x = fmaf(k, w.x, grad_scale * g.x); g.y = fmaf(k, w.y, grad_scale * g.y);
      g.z = fmaf(k, w.z, grad_scale * g.z); g.w = fmaf(k, w.w, grad_scale * g.w);
      st4(row_g2 + (list * CH + u) * D + q * 4, g);
      sq = g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
      if (q == 0) {
        g1 = fmaf(k, tb.w1[static_cast<int64_t>(row) * tb.stride1], grad_scale * g1);
        row_g1[list * CH + u] = g1;
        sq = fmaf(g1, g1, sq);
      }
    }
  }
  // fixed-order block reduction -> one partial per block
  __shared__ float wsum[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, float b1, float b2,
                                      float step_size, float inv_bc2_sqrt, float eps) {
  m = fmaf(b1, m, (1.f - b1) * g);
  v = fmaf(b2, v, (1.f - b2) * g * g);
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  w -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void rowadam_apply_kernel(
    TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ row_g2,
    const float* __restrict__ row_g1, const int32_t* __restrict__ owner_flag,
    const float* __restrict__ clip_coef, float lr, float b1, float b2, float eps,
    const int32_t* __restrict__ step_ptr) {
  const int lpr = D / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int q = static_cast<int>(t % lpr);
  const int64_t entry = t / lpr;
  const int64_t list = entry / CH;
  const int u = static_cast<int>(entry % CH);
  if (list >= static_cast<int64_t>(L) * S || u >= num_uniq[list]) return;
  if (!owner_flag[list * CH + u]) return;
  const int s = static_cast<int>(list % S);
  const int64_t row = uniq_rows[list * CH + u];
  const float clip = clip_coef ? clip_coef[0] : 1.f;
  const float step = static_cast<float>(step_ptr[0]);
  const float bc1 = 1.f - powf(b1, step);
  const float bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1;
  const float inv_bc2_sqrt = 1.f / sqrtf(bc2);
  const dfm_table tb = tabs.t[s];
  float4 g = ld4(row_g2 + (list * CH + u) * D + q * 4);
  const int64_t o2 = row * tb.stride2 + q * 4, o1 = row * tb.stride1;
  float4 w = ld4(tb.w2 + o2);
  float4 m = ld4(tb.m2 + o2);
  float4 v = ld4(tb.v2 + o2);
  adam1(w.x, m.x, v.x, g.x * clip, b1, b2, step_size, inv_bc2_sqrt, eps);
  adam1(w.y, m.y, v.y, g.y * clip, b1, b2, step_size, inv_bc2_sqrt, eps);
  adam1(w.z, m.z, v.z, g.z * clip, b1, b2, step_size, inv_bc2_sqrt, eps);
  adam1(w.w, m.w, v.w, g.w * clip, b1, b2, step_size, inv_bc2_sqrt, eps);
  st4(tb.w2 + o2, w);
  st4(tb.m2 + o2, m);
  st4(tb.v2 + o2, v);
  if (q == 0) {
    float w1 = tb.w1[o1], m1 = tb.m1[o1], v1 = tb.v1[o1];
    adam1(w1, m1, v1, row_g1[list * CH + u] * clip, b1, b2, step_size, inv_bc2_sqrt, eps);
    tb.w1[o1] = w1; tb.m1[o1] = m1; tb.v1[o1] = v1;
  }
}

static int fill_tables(const dfm_table* tables, int S, int D, TableArgs* out, bool need_state) {
  memset(out, 0, sizeof(*out));
  for (int s = 0; s < S; ++s) {
    DFM_REQUIRE(tables[s].w2 && tables[s].w1, "table %d: null weights", s);
    if (need_state)
      DFM_REQUIRE(tables[s].m2 && tables[s].v2 && tables[s].m1 && tables[s].v1, "table %d: null Adam state", s);
    out->t[s] = tables[s];
    if (out->t[s].stride2 == 0) out->t[s].stride2 = D;
    if (out->t[s].stride1 == 0) out->t[s].stride1 = 1;
    DFM_REQUIRE(out->t[s].stride2 >= D && out->t[s].stride2 % 4 == 0 && out->t[s].stride1 >= 1,
                "table %d: bad row strides", s);
  }
  return DFM_OK;
}

static inline int64_t merge_blocks(int S, int D, int L) {
  const int64_t threads = static_cast<int64_t>(L) * S * CH * (D / 4);
  return (threads + 255) / 256;
}

extern "C" {

int64_t dfm_rowadam_num_partials(int num_sparse, int dim, int num_lists) {
  return merge_blocks(num_sparse, dim, num_lists);  // one |g|^2 partial per merge block
}

int dfm_rowadam_merge(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                      const int32_t* d_uniq_rows, const int32_t* d_num_uniq, float* d_row_g2,
                      float* d_row_g1, int32_t* d_owner_flag, float grad_scale, float l2,
                      float* d_partials, dfm_stream_t stream) {
  DFM_REQUIRE(tables && d_uniq_rows && d_num_uniq && d_row_g2 && d_row_g1 && d_owner_flag && d_partials,
              "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS && num_lists > 0, "bad sizes");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 256, "dim must be a multiple of 4 and <= 256");
  TableArgs ta;
  if (int rc = fill_tables(tables, num_sparse, dim, &ta, false)) return rc;
  const int64_t blocks = merge_blocks(num_sparse, dim, num_lists);
  hipStream_t st = as_stream(stream);
  float* partial = d_partials;
  hipLaunchKernelGGL(rowadam_merge_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, ta,
                     num_sparse, dim, num_lists, d_uniq_rows, d_num_uniq, d_row_g2, d_row_g1,
                     d_owner_flag, grad_scale, l2, partial);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int dfm_rowadam_apply(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                      const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                      const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef,
                      float lr, float beta1, float beta2, float eps, const int32_t* d_step,
                      dfm_stream_t stream) {
  DFM_REQUIRE(tables && d_uniq_rows && d_num_uniq && d_row_g2 && d_row_g1 && d_owner_flag && d_step,
              "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS && num_lists > 0, "bad sizes");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 256, "dim must be a multiple of 4 and <= 256");
  TableArgs ta;
  if (int rc = fill_tables(tables, num_sparse, dim, &ta, true)) return rc;
  const int64_t blocks = merge_blocks(num_sparse, dim, num_lists);
  hipLaunchKernelGGL(rowadam_apply_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     as_stream(stream), ta, num_sparse, dim, num_lists, d_uniq_rows, d_num_uniq,
                     d_row_g2, d_row_g1, d_owner_flag, d_clip_coef, lr, beta1, beta2, eps, d_step);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

}  // extern "C"
