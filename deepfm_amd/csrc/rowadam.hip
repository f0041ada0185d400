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
#include "tail_bodies.h"

using namespace dfm;

using tail::CH;
using tail::TableArgs;

__global__ __launch_bounds__(256) void rowadam_merge_kernel(
    TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, float* __restrict__ row_g2, float* __restrict__ row_g1,
    int32_t* __restrict__ owner_flag, float grad_scale, float l2, float* __restrict__ partial) {
  tail::rowadam_merge_body(blockIdx.x, tabs, S, D, L, uniq_rows, num_uniq, row_g2, row_g1, owner_flag, grad_scale, l2,
                           partial);
}

__global__ __launch_bounds__(256) void rowadam_apply_kernel(
    TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ row_g2,
    const float* __restrict__ row_g1, const int32_t* __restrict__ owner_flag,
    const float* __restrict__ clip_coef, float lr, float b1, float b2, float eps,
    const int32_t* __restrict__ step_ptr) {
  tail::rowadam_apply_body(blockIdx.x, tabs, S, D, L, uniq_rows, num_uniq, row_g2, row_g1, owner_flag, clip_coef, lr,
                           b1, b2, eps, step_ptr);
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
