// Grouped launches for the tail of the training step (reference trainer.py:224-237 + the embedding
// backward of embedding.py:76-126): kernels that do not depend on each other share one dispatch,
// because on an MI355X a dependent launch costs ~4.5 us whatever it does.
//
//   dfm_step_embedding_backward : [DENSE-field Linear gradients | one gradient row per distinct id]
//   dfm_step_prepare            : [row lists: ownership merge + lazy L2 + |g|^2 |
//                                  dense buffer: + batch-split d-weight slabs, + L2, |g|^2]
//   (dfm_grad_norm_finalize     : clip coefficient, step / dropout-seed tick — one workgroup)
//   dfm_step_apply              : [row-wise Adam on the owned rows | Adam on the dense buffer]
//
// (Folding dfm_grad_norm_finalize into dfm_step_apply — every workgroup summing the ~2 500 partials itself
// instead of a one-workgroup launch in between — was measured and lost: apply 20.0 -> 25.2 us for the
// 4.9 us launch it removes.)
//
// The bodies are the ones of the stand-alone kernels (tail_bodies.h): identical arithmetic and
// reduction order, so grouped and stand-alone launches give bit-identical results.
#include "rowplan_body.h"
#include "tail_bodies.h"

using namespace dfm;
using namespace dfm::tail;

namespace {
// dense_prepare_body with the batch-split d-weight products folded in, ONE float4 per thread (the
// slab sums want many threads with few dependent loads each): parameters start on 64-byte
// boundaries and weights have a multiple of 4 elements, so a float4 belongs to at most one
// slab-backed weight; its slabs are added in order (8 loads in flight) before the L2 term.
constexpr int kStepPrepPerThread = 4;
//
// Data parallel: `gathered` holds every rank's dense gradient buffer (world x n, rank-major, from the
// step's single all-gather); the mean over ranks is formed here, in rank order — an all-reduce whose
// summation order is fixed, at the cost of no extra launch.
__device__ __forceinline__ void dense_prepare_slabs_body(int blk, float* __restrict__ g, const float* __restrict__ p,
                                                         int64_t n, int64_t n_l2, float l2, const SlabTable& st,
                                                         const float* __restrict__ gathered, int world,
                                                         int64_t gathered_stride, float scale,
                                                         float* __restrict__ partial) {
  const int64_t i = (static_cast<int64_t>(blk) * kTailThreads + threadIdx.x) * kStepPrepPerThread;
  float sq = 0.f;
  const float k = 2.f * l2;
  if (i + 3 < n) {
    float4 gi = ld4(g + i);
    bool dirty = false;
    if (gathered) {
      gi = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < world; ++r) {
        const float4 t = ld4(gathered + static_cast<int64_t>(r) * gathered_stride + i);
        gi.x += t.x; gi.y += t.y; gi.z += t.z; gi.w += t.w;
      }
      gi.x *= scale; gi.y *= scale; gi.z *= scale; gi.w *= scale;
      dirty = true;
    }
    if (add_slabs(gi, g + i, st)) dirty = true;
    if (i < n_l2) {                          // n_l2 is a multiple of 16 (padded parameters)
      const float4 pi = ld4(p + i);
      gi.x = fmaf(k, pi.x, gi.x); gi.y = fmaf(k, pi.y, gi.y); gi.z = fmaf(k, pi.z, gi.z); gi.w = fmaf(k, pi.w, gi.w);
      dirty = true;
    }
    if (dirty) st4(g + i, gi);
    sq = fmaf(gi.x, gi.x, sq); sq = fmaf(gi.y, gi.y, sq); sq = fmaf(gi.z, gi.z, sq); sq = fmaf(gi.w, gi.w, sq);
  } else {
    for (int64_t e = i; e < n; ++e) {
      float ge = g[e];
      if (gathered) {
        ge = 0.f;
        for (int r = 0; r < world; ++r) ge += gathered[static_cast<int64_t>(r) * gathered_stride + e];
        ge *= scale;
        g[e] = ge;
      }
      if (e < n_l2) { ge = fmaf(k, p[e], ge); g[e] = ge; }
      sq = fmaf(ge, ge, sq);
    }
  }
  block_partial(sq, partial, blk);
}
}  // namespace

__global__ __launch_bounds__(kTailThreads) void step_embedding_backward_kernel(
    int dense_blocks, const int32_t* __restrict__ dense_list, PtrTable in, GradTable gt, int64_t B, int F, int D,
    const float* __restrict__ g_first, const float* __restrict__ g_field, FieldMap fmap, int S, int lists,
    const int32_t* __restrict__ sorted_pos, int32_t* seg_start,
    const int32_t* __restrict__ num_uniq, float* __restrict__ row_g2, float* __restrict__ row_g1,
    DensePartials dp) {
  const int blk = blockIdx.x;
  if (blk < dense_blocks)
    dense_fields_uniform_body(blk, dense_list, in, gt, B, F, D, g_first, g_field, dp);
  else
    rowgrad_body(blk - dense_blocks, fmap, S, F, D, lists, g_first, g_field, sorted_pos, seg_start, num_uniq, row_g2,
                 row_g1);
}

constexpr int kMatchThreads = 1024;
__global__ __launch_bounds__(kMatchThreads) void step_match_kernel(int S, int L, const int32_t* __restrict__ uniq_rows,
                                                                  const int32_t* __restrict__ num_uniq,
                                                                  uint8_t* __restrict__ match) {
  rowadam_match_body<kMatchThreads>(blockIdx.x, S, L, uniq_rows, num_uniq, match);
}

__global__ __launch_bounds__(kTailThreads) void step_prepare_kernel(
    int merge_blocks, TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, float* __restrict__ row_g2, float* __restrict__ row_g1,
    int32_t* __restrict__ owner_flag, float grad_scale, float l2, float* __restrict__ g, const float* __restrict__ p,
    int64_t n, int64_t n_l2, SlabTable slabs, const float* __restrict__ dense_gathered, int world,
    int64_t gathered_stride, float* __restrict__ partial, int64_t dense_partial_offset,
    const uint8_t* __restrict__ match) {
  const int blk = blockIdx.x;
  if (blk < merge_blocks)
    rowadam_merge_body(blk, tabs, S, D, L, uniq_rows, num_uniq, row_g2, row_g1, owner_flag, grad_scale, l2, partial,
                       match);
  else
    dense_prepare_slabs_body(blk - merge_blocks, g, p, n, n_l2, l2, slabs, dense_gathered, world, gathered_stride,
                             grad_scale, partial + dense_partial_offset);
}

__global__ __launch_bounds__(kTailThreads) void step_apply_kernel(
    int row_blocks, TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ row_g2, const float* __restrict__ row_g1,
    const int32_t* __restrict__ owner_flag, const float* __restrict__ clip_coef, float lr, float b1, float b2,
    float eps, const int32_t* __restrict__ step_ptr, float* __restrict__ p, float* __restrict__ m,
    float* __restrict__ v, float* __restrict__ g, int64_t n, int zero_grad) {
  const int blk = blockIdx.x;
  if (blk < row_blocks)
    rowadam_apply_body(blk, tabs, S, D, L, uniq_rows, num_uniq, row_g2, row_g1, owner_flag, clip_coef, lr, b1, b2, eps,
                       step_ptr);
  else
    dense_adam_body(blk - row_blocks, p, m, v, g, n, clip_coef, lr, b1, b2, eps, step_ptr, zero_grad ? g : nullptr);
}

// step_apply of step t + the ROW PLAN of step t + 1 (+ its row touch) in one launch (round 3).  The plan needs only
// the next batch's ids, and on its own it keeps 26 CUs busy for ~13 us in the step's dependent chain; here its
// workgroups are the FIRST blocks of the optimizer's last launch — they start at once and sort beside the
// row-wise Adam, which is bound by HBM latency, not by CUs.  1024-thread workgroups with the plan's dynamic LDS:
// an apply workgroup runs four of step_apply_kernel's 256-thread blocks (their bodies index by the flat thread
// number only).  The next step then starts at its gather.  Ids: (S, n) int64, column s at ids_base + s * ids_stride
// (a batch record); vocab on the device; plan outputs = the OTHER set of plan buffers.
template <typename KeyT, int SHIFT>
__global__ __launch_bounds__(rowplan::SORT_THREADS) void step_apply_plan_kernel(
    int plan_blocks, const int64_t* __restrict__ ids_base, int64_t ids_stride, const int32_t* __restrict__ vocab,
    int64_t plan_n, int chunks, int32_t* __restrict__ p_sorted_pos, int32_t* __restrict__ p_uniq_rows,
    int32_t* __restrict__ p_seg_start, int32_t* __restrict__ p_num_uniq, int32_t* p_error,
    int row_blocks4, TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ row_g2, const float* __restrict__ row_g1,
    const int32_t* __restrict__ owner_flag, const float* __restrict__ clip_coef, float lr, float b1, float b2,
    float eps, const int32_t* __restrict__ step_ptr, float* __restrict__ p, float* __restrict__ m,
    float* __restrict__ v, float* __restrict__ g, int64_t n, int zero_grad) {
  if (static_cast<int>(blockIdx.x) < plan_blocks) {
    const int s = blockIdx.x % S, y = blockIdx.x / S;
    const int64_t* src = ids_base + s * ids_stride;
    if (y < chunks) {
      rowplan::rowplan_chunk_body<KeyT, SHIFT>(src, vocab[s], s, y, S, plan_n, p_sorted_pos, p_uniq_rows, p_seg_start,
                                               p_num_uniq, p_error, 0);
    } else {
      const int q = y - chunks;
      const dfm_table tb = tabs.t[s];
      rowplan::rowplan_touch_body(src, vocab[s], tb.w2, tb.stride2, q / rowplan::kTouchParts, q % rowplan::kTouchParts,
                                  plan_n);
    }
    return;
  }
  const int blk = (static_cast<int>(blockIdx.x) - plan_blocks) * (rowplan::SORT_THREADS / kTailThreads);
  if (blk < row_blocks4)
    rowadam_apply_body(blk, tabs, S, D, L, uniq_rows, num_uniq, row_g2, row_g1, owner_flag, clip_coef, lr, b1, b2, eps,
                       step_ptr);
  else
    dense_adam_body(blk - row_blocks4, p, m, v, g, n, clip_coef, lr, b1, b2, eps, step_ptr, zero_grad ? g : nullptr);
}

namespace {
int fill_tables(const dfm_table* tables, int S, int D, TableArgs* out, bool need_state) {
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
inline int64_t row_blocks(int S, int D, int L) {
  return (static_cast<int64_t>(L) * S * CH * (D / 4) + kTailThreads - 1) / kTailThreads;
}
inline int64_t prep_blocks(int64_t n) {
  const int64_t per = static_cast<int64_t>(kTailThreads) * kStepPrepPerThread;
  return (n + per - 1) / per;
}
}  // namespace

extern "C" int dfm_step_embedding_backward(const int32_t* d_dense_list, int num_dense, const void* const* dense_x,
                                           const dfm_field_grad* dense_grads, const int32_t* field_of_sparse,
                                           int num_sparse, int num_fields, int dim, int64_t batch,
                                           const float* d_g_first, const float* d_g_field,
                                           const int32_t* d_sorted_pos, int32_t* d_seg_start,
                                           const int32_t* d_num_uniq, float* d_row_g2, float* d_row_g1,
                                           float* d_dense_partials, int dense_parts, const float* d_dense_grad_base,
                                           int64_t dense_grad_elems, dfm_stream_t stream) {
  DFM_REQUIRE(d_g_first && d_g_field, "null argument");
  DFM_REQUIRE(num_dense >= 0 && num_sparse >= 0 && num_dense + num_sparse > 0 && num_fields <= DFM_MAX_FIELDS &&
                  num_dense + num_sparse <= num_fields, "bad field counts");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && batch > 0 && batch < (int64_t(1) << 31), "bad shape");
  DFM_REQUIRE((reinterpret_cast<uintptr_t>(d_g_field) & 15) == 0, "d_g_field must be 16-byte aligned");
  PtrTable in;
  GradTable gt;
  memset(&in, 0, sizeof(in));
  memset(&gt, 0, sizeof(gt));
  if (num_dense > 0) {
    // dense_x / dense_grads are indexed by SCHEMA position (like dfm_embedding_forward's inputs);
    // d_dense_list holds the schema positions of the DENSE fields on the device
    DFM_REQUIRE(d_dense_list && dense_x && dense_grads, "null argument");
    for (int f = 0; f < num_fields; ++f) {
      in.p[f] = dense_x[f];
      gt.g[f] = dense_grads[f];
    }
  }
  FieldMap fm;
  memset(&fm, 0, sizeof(fm));
  int lists = 0;
  int64_t rg_blocks = 0;
  if (num_sparse > 0) {
    DFM_REQUIRE(field_of_sparse && d_sorted_pos && d_seg_start && d_num_uniq && d_row_g2 && d_row_g1, "null argument");
    for (int s = 0; s < num_sparse; ++s) {
      DFM_REQUIRE(field_of_sparse[s] >= 0 && field_of_sparse[s] < num_fields, "field_of_sparse[%d] out of range", s);
      fm.f[s] = field_of_sparse[s];
    }
    lists = static_cast<int>((batch + CH - 1) / CH) * num_sparse;
    rg_blocks = (static_cast<int64_t>(lists) * CH * (dim / 4) + kTailThreads - 1) / kTailThreads;
  }
  int dense_blocks = num_dense * (dim / 4 + 1);
  DensePartials dp = {nullptr, nullptr, 0, 0, 0};
  if (d_dense_partials && num_dense > 0) {
    DFM_REQUIRE(dense_parts >= 1 && dense_parts <= 64 && d_dense_grad_base && dense_grad_elems > 0, "bad dense slices");
    for (int i = 0; i < num_fields; ++i) {
      const dfm_field_grad& g = gt.g[i];
      if (!g.w2) continue;
      DFM_REQUIRE(g.w2 >= d_dense_grad_base && g.w2 + dim <= d_dense_grad_base + dense_grad_elems &&
                      g.b2 >= d_dense_grad_base && g.b2 + dim <= d_dense_grad_base + dense_grad_elems &&
                      g.w1 >= d_dense_grad_base && g.w1 < d_dense_grad_base + dense_grad_elems &&
                      g.b1 >= d_dense_grad_base && g.b1 < d_dense_grad_base + dense_grad_elems,
                  "field %d: a DENSE-field gradient buffer lies outside the sliced range", i);
    }
    dp.out = d_dense_partials; dp.base = d_dense_grad_base; dp.elems = dense_grad_elems;
    dp.rows = (batch + dense_parts - 1) / dense_parts;
    dp.per_slice = dense_blocks;
    dense_blocks *= dense_parts;
  }
  hipLaunchKernelGGL(step_embedding_backward_kernel, dim3(static_cast<unsigned>(dense_blocks + rg_blocks)),
                     dim3(kTailThreads), 0, as_stream(stream), dense_blocks, d_dense_list, in, gt, batch, num_fields, dim,
                     d_g_first, d_g_field, fm, num_sparse, lists, d_sorted_pos, d_seg_start, d_num_uniq, d_row_g2,
                     d_row_g1, dp);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" size_t dfm_step_match_bytes(int num_sparse, int num_lists) {
  return num_lists > 2 ? static_cast<size_t>(num_lists) * num_sparse * CH * num_lists : 0;
}

extern "C" int64_t dfm_step_prepare_num_partials(int num_sparse, int dim, int num_lists, int64_t n) {
  return row_blocks(num_sparse, dim, num_lists) + prep_blocks(n);
}

extern "C" int dfm_step_prepare(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                                const int32_t* d_uniq_rows, const int32_t* d_num_uniq, float* d_row_g2,
                                float* d_row_g1, int32_t* d_owner_flag, float grad_scale, float l2, float* d_g,
                                const float* d_p, int64_t n, int64_t n_l2, const dfm_slab_ref* slabs, int num_slabs,
                                const float* d_dense_gathered, int world, int64_t gathered_stride, float* d_partials,
                                int64_t dense_partial_offset, void* d_match, dfm_stream_t stream) {
  DFM_REQUIRE(tables && d_uniq_rows && d_num_uniq && d_row_g2 && d_row_g1 && d_owner_flag && d_partials && d_g && d_p,
              "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS && num_lists > 0, "bad sizes");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 256, "dim must be a multiple of 4 and <= 256");
  DFM_REQUIRE(n > 0 && n_l2 >= 0 && n_l2 <= n && n_l2 % 16 == 0, "bad dense sizes (n_l2 must be a multiple of 16)");
  DFM_REQUIRE((reinterpret_cast<uintptr_t>(d_g) & 63) == 0 && (reinterpret_cast<uintptr_t>(d_p) & 15) == 0,
              "dense buffers must be 64-byte aligned");
  DFM_REQUIRE(!d_dense_gathered || (world >= 1 && (reinterpret_cast<uintptr_t>(d_dense_gathered) & 15) == 0 && n % 4 == 0),
              "gathered dense gradients: world >= 1, 16-byte aligned, n a multiple of 4");
  if (d_dense_gathered && gathered_stride == 0) gathered_stride = n;
  DFM_REQUIRE(!d_dense_gathered || (gathered_stride >= n && gathered_stride % 4 == 0),
              "gathered dense gradients: the stride between ranks must be >= n and a multiple of 4 floats");
  TableArgs ta;
  if (int rc = fill_tables(tables, num_sparse, dim, &ta, false)) return rc;
  SlabTable st = {};
  if (int rc = fill_slab_table(slabs, num_slabs, d_g, n, &st)) return rc;
  const int64_t mb = row_blocks(num_sparse, dim, num_lists), pb = prep_blocks(n);
  if (dense_partial_offset == 0) dense_partial_offset = mb;
  DFM_REQUIRE(dense_partial_offset >= mb, "the dense partials must not overlap the %lld row partials", (long long)mb);
  // with three or more lists (data-parallel ranks x chunks) the list memberships are resolved once,
  // through LDS, instead of by L-1 global binary searches per entry inside the merge
  uint8_t* match = (d_match && num_lists > 2) ? static_cast<uint8_t*>(d_match) : nullptr;
  if (match) {
    DFM_REQUIRE(num_lists <= 255, "at most 255 lists");
    hipLaunchKernelGGL(step_match_kernel, dim3(static_cast<unsigned>(num_sparse * num_lists * num_lists)),
                       dim3(kMatchThreads), 0, as_stream(stream), num_sparse, num_lists, d_uniq_rows, d_num_uniq, match);
    DFM_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(step_prepare_kernel, dim3(static_cast<unsigned>(mb + pb)), dim3(kTailThreads), 0,
                     as_stream(stream), static_cast<int>(mb), ta, num_sparse, dim, num_lists, d_uniq_rows, d_num_uniq,
                     d_row_g2, d_row_g1, d_owner_flag, grad_scale, l2, d_g, d_p, n, n_l2, st, d_dense_gathered, world,
                     gathered_stride, d_partials, dense_partial_offset, match);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_step_apply(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                              const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                              const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef, float lr,
                              float beta1, float beta2, float eps, const int32_t* d_step, float* d_p, float* d_m,
                              float* d_v, float* d_g, int64_t n, int zero_grad, dfm_stream_t stream) {
  DFM_REQUIRE(tables && d_uniq_rows && d_num_uniq && d_row_g2 && d_row_g1 && d_owner_flag && d_step && d_p && d_m &&
                  d_v && d_g, "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS && num_lists > 0 && n > 0, "bad sizes");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 256, "dim must be a multiple of 4 and <= 256");
  TableArgs ta;
  if (int rc = fill_tables(tables, num_sparse, dim, &ta, true)) return rc;
  const int64_t rb = row_blocks(num_sparse, dim, num_lists), ab = (n + kTailThreads - 1) / kTailThreads;
  hipLaunchKernelGGL(step_apply_kernel, dim3(static_cast<unsigned>(rb + ab)), dim3(kTailThreads), 0, as_stream(stream),
                     static_cast<int>(rb), ta, num_sparse, dim, num_lists, d_uniq_rows, d_num_uniq, d_row_g2, d_row_g1,
                     d_owner_flag, d_clip_coef, lr, beta1, beta2, eps, d_step, d_p, d_m, d_v, d_g, n, zero_grad);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// ---- dfm_step_apply + the next step's row plan in one launch -----------------------------------------
namespace {
struct ApplyPlanLaunch {
  const void* func = nullptr;
  dim3 grid, block;
  unsigned lds = 0;
  int plan_blocks = 0, chunks = 0, row_blocks4 = 0, S = 0, D = 0, L = 0, zero_grad = 0;
  const int64_t* ids_base = nullptr;
  int64_t ids_stride = 0, plan_n = 0, n = 0;
  const int32_t* vocab = nullptr;
  int32_t *p_sorted_pos = nullptr, *p_uniq_rows = nullptr, *p_seg_start = nullptr, *p_num_uniq = nullptr, *p_error = nullptr;
  TableArgs tabs;
  const int32_t *uniq_rows = nullptr, *num_uniq = nullptr, *owner_flag = nullptr, *step_ptr = nullptr;
  const float *row_g2 = nullptr, *row_g1 = nullptr, *clip_coef = nullptr;
  float lr = 0, b1 = 0, b2 = 0, eps = 0;
  float *p = nullptr, *m = nullptr, *v = nullptr, *g = nullptr;
  void* params[36];
  void bind() {
    int k = 0;
    params[k++] = &plan_blocks; params[k++] = &ids_base; params[k++] = &ids_stride; params[k++] = &vocab;
    params[k++] = &plan_n; params[k++] = &chunks; params[k++] = &p_sorted_pos; params[k++] = &p_uniq_rows;
    params[k++] = &p_seg_start; params[k++] = &p_num_uniq; params[k++] = &p_error; params[k++] = &row_blocks4;
    params[k++] = &tabs; params[k++] = &S; params[k++] = &D; params[k++] = &L; params[k++] = &uniq_rows;
    params[k++] = &num_uniq; params[k++] = &row_g2; params[k++] = &row_g1; params[k++] = &owner_flag;
    params[k++] = &clip_coef; params[k++] = &lr; params[k++] = &b1; params[k++] = &b2; params[k++] = &eps;
    params[k++] = &step_ptr; params[k++] = &p; params[k++] = &m; params[k++] = &v; params[k++] = &g; params[k++] = &n;
    params[k++] = &zero_grad;
  }
};

int describe_apply_plan(const dfm_table* tables, int num_sparse, int dim, int num_lists, const int32_t* d_uniq_rows,
                        const int32_t* d_num_uniq, const float* d_row_g2, const float* d_row_g1,
                        const int32_t* d_owner_flag, const float* d_clip_coef, float lr, float beta1, float beta2,
                        float eps, const int32_t* d_step, float* d_p, float* d_m, float* d_v, float* d_g, int64_t n,
                        int zero_grad, const int64_t* d_next_ids, int64_t ids_stride, const int32_t* d_vocab,
                        int max_vocab, int64_t batch, int32_t* d_next_sorted_pos, int32_t* d_next_uniq_rows,
                        int32_t* d_next_seg_start, int32_t* d_next_num_uniq, int32_t* d_error_flag, ApplyPlanLaunch* a) {
  DFM_REQUIRE(tables && d_uniq_rows && d_num_uniq && d_row_g2 && d_row_g1 && d_owner_flag && d_step && d_p && d_m &&
                  d_v && d_g, "null argument");
  DFM_REQUIRE(d_next_ids && d_vocab && d_next_sorted_pos && d_next_uniq_rows && d_next_seg_start && d_next_num_uniq,
              "null row-plan argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS && num_lists > 0 && n > 0, "bad sizes");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 256, "dim must be a multiple of 4 and <= 256");
  DFM_REQUIRE(batch > 0 && batch < (int64_t(1) << 31) && ids_stride >= batch && max_vocab > 0, "bad batch / id stride");
  if (int rc = fill_tables(tables, num_sparse, dim, &a->tabs, true)) return rc;
  const bool narrow = max_vocab < (1 << 20) - 1;           // as dfm_rowplan_build
  a->func = narrow ? reinterpret_cast<const void*>(step_apply_plan_kernel<uint32_t, 12>)
                   : reinterpret_cast<const void*>(step_apply_plan_kernel<unsigned long long, 32>);
  a->lds = rowplan::lds_bytes(narrow);
  static bool allowed[2] = {false, false};
  if (!allowed[narrow ? 0 : 1]) {
    DFM_HIP_TRY(hipFuncSetAttribute(a->func, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a->lds)));
    allowed[narrow ? 0 : 1] = true;
  }
  constexpr int kPer = rowplan::SORT_THREADS / kTailThreads;
  const int64_t rb = row_blocks(num_sparse, dim, num_lists), ab = (n + kTailThreads - 1) / kTailThreads;
  const int64_t rb4 = (rb + kPer - 1) / kPer * kPer;
  a->chunks = static_cast<int>((batch + CH - 1) / CH);
  a->plan_blocks = num_sparse * a->chunks * (1 + rowplan::kTouchParts);
  a->row_blocks4 = static_cast<int>(rb4);
  a->grid = dim3(static_cast<unsigned>(a->plan_blocks + rb4 / kPer + (ab + kPer - 1) / kPer));
  a->block = dim3(rowplan::SORT_THREADS);
  a->S = num_sparse; a->D = dim; a->L = num_lists; a->zero_grad = zero_grad;
  a->ids_base = d_next_ids; a->ids_stride = ids_stride; a->vocab = d_vocab; a->plan_n = batch; a->n = n;
  a->p_sorted_pos = d_next_sorted_pos; a->p_uniq_rows = d_next_uniq_rows; a->p_seg_start = d_next_seg_start;
  a->p_num_uniq = d_next_num_uniq; a->p_error = d_error_flag;
  a->uniq_rows = d_uniq_rows; a->num_uniq = d_num_uniq; a->row_g2 = d_row_g2; a->row_g1 = d_row_g1;
  a->owner_flag = d_owner_flag; a->clip_coef = d_clip_coef; a->lr = lr; a->b1 = beta1; a->b2 = beta2; a->eps = eps;
  a->step_ptr = d_step; a->p = d_p; a->m = d_m; a->v = d_v; a->g = d_g;
  a->bind();
  return DFM_OK;
}
}  // namespace

#define DFM_APPLY_PLAN_ARGS                                                                                              \
  tables, num_sparse, dim, num_lists, d_uniq_rows, d_num_uniq, d_row_g2, d_row_g1, d_owner_flag, d_clip_coef, lr, beta1, \
      beta2, eps, d_step, d_p, d_m, d_v, d_g, n, zero_grad, d_next_ids, ids_stride, d_vocab, max_vocab, batch,           \
      d_next_sorted_pos, d_next_uniq_rows, d_next_seg_start, d_next_num_uniq, d_error_flag

extern "C" int dfm_step_apply_plan(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                                   const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                                   const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef,
                                   float lr, float beta1, float beta2, float eps, const int32_t* d_step, float* d_p,
                                   float* d_m, float* d_v, float* d_g, int64_t n, int zero_grad,
                                   const int64_t* d_next_ids, int64_t ids_stride, const int32_t* d_vocab, int max_vocab,
                                   int64_t batch, int32_t* d_next_sorted_pos, int32_t* d_next_uniq_rows,
                                   int32_t* d_next_seg_start, int32_t* d_next_num_uniq, int32_t* d_error_flag,
                                   dfm_stream_t stream) {
  ApplyPlanLaunch a;
  if (int rc = describe_apply_plan(DFM_APPLY_PLAN_ARGS, &a)) return rc;
  DFM_HIP_TRY(hipLaunchKernel(a.func, a.grid, a.block, a.params, a.lds, as_stream(stream)));
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_step_apply_plan_update(void* graph_exec, void* node, const dfm_table* tables, int num_sparse, int dim,
                                          int num_lists, const int32_t* d_uniq_rows, const int32_t* d_num_uniq,
                                          const float* d_row_g2, const float* d_row_g1, const int32_t* d_owner_flag,
                                          const float* d_clip_coef, float lr, float beta1, float beta2, float eps,
                                          const int32_t* d_step, float* d_p, float* d_m, float* d_v, float* d_g, int64_t n,
                                          int zero_grad, const int64_t* d_next_ids, int64_t ids_stride,
                                          const int32_t* d_vocab, int max_vocab, int64_t batch,
                                          int32_t* d_next_sorted_pos, int32_t* d_next_uniq_rows,
                                          int32_t* d_next_seg_start, int32_t* d_next_num_uniq, int32_t* d_error_flag) {
  DFM_REQUIRE(graph_exec && node, "null argument");
  ApplyPlanLaunch a;
  if (int rc = describe_apply_plan(DFM_APPLY_PLAN_ARGS, &a)) return rc;
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = const_cast<void*>(a.func);
  p.gridDim = a.grid;
  p.blockDim = a.block;
  p.sharedMemBytes = a.lds;
  p.kernelParams = a.params;
  p.extra = nullptr;
  DFM_HIP_TRY(hipGraphExecKernelNodeSetParams(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipGraphNode_t>(node), &p));
  return DFM_OK;
}
