// Field-sharded embedding tables under data parallelism (the reference trains on one
// device only, trainer.py:47-56, so the layout of the exchange is this build's own).
//
// Rank r of N owns the tables of the SPARSE fields [first_r, first_r + nf_r) — rows and Adam moments
// live on one GPU only — and serves them to every rank's minibatch; everything dense (DNN, heads,
// DENSE-field Linears) stays replicated.  One step moves, per rank and direction, the activations of
// its batch instead of row updates:
//
//   ids      all-to-all   the batch's ids of q's fields                       -> owner q
//   rows     all-to-all   [e (B, nf_r, D) | first-order w (B, nf_r)]          <- owner r   (shard_gather)
//   grads    all-to-all   [d e (B, nf_q, D) | d first (B) | dense grads (n)]  -> owner q   (shard_pack)
//
// after which the owner reduces the received gradients to one row per distinct id with the row plan of
// the GLOBAL batch (rowplan.hip, tail_bodies.h::rowgrad_body with SampleSegments) and runs the row-wise
// Adam on its own rows.  Segments are laid out so that both sides of every all-to-all are plain
// contiguous buffers with a fixed split per peer; all kernels here are streaming / gather copies,
// HBM-bound, one float4 per thread.
#include "tail_bodies.h"

using namespace dfm;
using namespace dfm::tail;

namespace {
constexpr int kThreads = 256;

struct ShardTables {
  const float* w2[DFM_MAX_FIELDS];
  const float* w1[DFM_MAX_FIELDS];
  int32_t stride2[DFM_MAX_FIELDS];
  int32_t stride1[DFM_MAX_FIELDS];
  int32_t vocab[DFM_MAX_FIELDS];
};

// The staged copy of a batch record: the first kernel node of a captured step, re-pointed at the next
// record by dfm_stage_record_update.
__global__ __launch_bounds__(kThreads) void stage_record_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                                int64_t n16) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

// ids (world, nf, B) as received (source-rank major) -> rows of the owned tables, written in the send
// layout of the rows all-to-all: segment p = [e (B, nf, D) | w (B, nf)], seg = B * nf * (D + 1) floats.
// Also leaves the ids field-major, gids (nf, world * B), for the owner's row plan.
__global__ __launch_bounds__(kThreads) void shard_gather_kernel(ShardTables tabs, int nf, int D, int world, int64_t B,
                                                                const int64_t* __restrict__ ids,
                                                                float* __restrict__ send, int64_t* __restrict__ gids,
                                                                int32_t* __restrict__ err) {
  const int lpr = D / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int q = static_cast<int>(t % lpr);
  const int64_t item = t / lpr;                       // (p, b, j), j fastest
  if (item >= static_cast<int64_t>(world) * B * nf) return;
  const int j = static_cast<int>(item % nf);
  const int64_t gb = item / nf;                       // p * B + b
  const int64_t p = gb / B, b = gb - p * B;
  const int64_t id = checked_id(ids[(p * nf + j) * B + b], tabs.vocab[j], err);
  const int64_t seg = B * nf * (D + 1);
  const float4 v = ld4(tabs.w2[j] + id * tabs.stride2[j] + 4 * q);
  st4(send + p * seg + (b * nf + j) * D + 4 * q, v);
  if (q == 0) {
    send[p * seg + B * nf * D + b * nf + j] = tabs.w1[j][id * tabs.stride1[j]];
    gids[static_cast<int64_t>(j) * world * B + gb] = id;
  }
}

struct PackPlan {
  int64_t seg_start[DFM_MAX_RANKS + 1];   // float offset of peer q's segment in the send buffer
  int32_t first[DFM_MAX_RANKS];           // q's first SPARSE field (index among the SPARSE fields)
  int32_t count[DFM_MAX_RANKS];           // how many it owns
  int32_t field_of_sparse[DFM_MAX_FIELDS];
  int32_t world;
};

// d field_embeddings (B, F, D), d first_order (B), the flat dense gradient (n) -> the send layout of the
// gradient all-to-all: segment q = [d e of q's fields (B, nf_q, D) | d first (B) | dense (n)].  The
// batch-split d-weight slabs of the tower (dfm_linear_backward) are added on the way, in slab order
// (what dfm_linear_backward_finish would have done in a launch of its own); the flat buffer itself is
// left as it is — the optimizer's prepare launch replaces it by the mean over ranks.
__global__ __launch_bounds__(kThreads) void shard_pack_kernel(PackPlan plan, int64_t B, int F, int D,
                                                              const float* __restrict__ g_field,
                                                              const float* __restrict__ g_first,
                                                              const float* __restrict__ dense, int64_t n,
                                                              SlabTable slabs, float* __restrict__ send) {
  // one float4 per thread, three ranges of threads: [d e of every SPARSE field | d first, once per peer |
  // the dense gradient, computed once and stored into every peer's segment]
  int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int S = plan.first[plan.world - 1] + plan.count[plan.world - 1];
  const int64_t emb4 = B * S * D / 4, first4 = B / 4, dense4 = n / 4;
  if (t < emb4) {
    const int64_t e = t * 4;                           // segments' row parts back to back: q's starts at B * first[q] * D
    int q = 0;
    while (q + 1 < plan.world && e >= B * plan.first[q + 1] * D) ++q;
    const int64_t o = e - B * plan.first[q] * D;
    const int nf = plan.count[q];
    const int64_t r = o / D;                           // b * nf + j
    const int64_t b = r / nf;
    const int j = static_cast<int>(r - b * nf);
    const int f = plan.field_of_sparse[plan.first[q] + j];
    st4(send + plan.seg_start[q] + o, ld4(g_field + (b * F + f) * D + (o - r * D)));
    return;
  }
  t -= emb4;
  if (t < first4 * plan.world) {
    const int q = static_cast<int>(t / first4);
    const int64_t o = (t - q * first4) * 4;
    st4(send + plan.seg_start[q] + B * plan.count[q] * D + o, ld4(g_first + o));
    return;
  }
  t -= first4 * plan.world;
  if (t >= dense4) return;
  const float* src = dense + t * 4;
  float4 v = ld4(src);
  add_slabs(v, src, slabs);
  for (int q = 0; q < plan.world; ++q) st4(send + plan.seg_start[q] + B * plan.count[q] * D + B + t * 4, v);
}

__global__ __launch_bounds__(kThreads) void shard_rowgrad_kernel(FieldMap fm, int S, int D, int lists,
                                                                 const float* __restrict__ g_first,
                                                                 const float* __restrict__ g_field,
                                                                 const int32_t* __restrict__ sorted_pos,
                                                                 int32_t* seg_start,
                                                                 const int32_t* __restrict__ num_uniq,
                                                                 float* __restrict__ row_g2, float* __restrict__ row_g1,
                                                                 SampleSegments segs) {
  rowgrad_body(blockIdx.x, fm, S, S, D, lists, g_first, g_field, sorted_pos, seg_start, num_uniq, row_g2, row_g1, segs);
}

// fixed-order sum of n floats (one workgroup): the owned rows' share of |g|^2, one float per rank
__global__ __launch_bounds__(1024) void sum_floats_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float wsum[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 16; ++i) tot += wsum[i];
    out[0] = tot;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int stage_params(const void* src, void* dst, int64_t nbytes, int64_t* n16, dim3* grid) {
  DFM_REQUIRE(src && dst && nbytes > 0 && nbytes % 16 == 0, "stage_record: non-null buffers, a positive multiple of 16 bytes");
  DFM_REQUIRE(aligned16(src) && aligned16(dst), "stage_record: 16-byte aligned buffers");
  *n16 = nbytes / 16;
  *grid = dim3(static_cast<unsigned>((*n16 + kThreads - 1) / kThreads));
  return DFM_OK;
}
}  // namespace

extern "C" int dfm_stage_record(const void* d_src, void* d_dst, int64_t nbytes, dfm_stream_t stream) {
  int64_t n16;
  dim3 grid;
  if (int rc = stage_params(d_src, d_dst, nbytes, &n16, &grid)) return rc;
  hipLaunchKernelGGL(stage_record_kernel, grid, dim3(kThreads), 0, as_stream(stream), static_cast<const uint4*>(d_src),
                     static_cast<uint4*>(d_dst), n16);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_stage_record_update(void* graph_exec, void* node, const void* d_src, void* d_dst, int64_t nbytes) {
  DFM_REQUIRE(graph_exec && node, "null argument");
  int64_t n16;
  dim3 grid;
  if (int rc = stage_params(d_src, d_dst, nbytes, &n16, &grid)) return rc;
  const uint4* src = static_cast<const uint4*>(d_src);
  uint4* dst = static_cast<uint4*>(d_dst);
  void* params[3] = {&src, &dst, &n16};
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = reinterpret_cast<void*>(stage_record_kernel);
  p.gridDim = grid;
  p.blockDim = dim3(kThreads);
  p.kernelParams = params;
  DFM_HIP_TRY(hipGraphExecKernelNodeSetParams(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipGraphNode_t>(node), &p));
  return DFM_OK;
}

extern "C" int dfm_shard_gather(const dfm_table* tables, const int32_t* vocab, int num_owned, int dim, int world,
                                int64_t batch, const int64_t* d_ids, float* d_send, int64_t* d_gids,
                                int32_t* d_error_flag, dfm_stream_t stream) {
  DFM_REQUIRE(tables && vocab && d_ids && d_send && d_gids, "null argument");
  DFM_REQUIRE(num_owned > 0 && num_owned <= DFM_MAX_FIELDS && world > 0 && world <= DFM_MAX_RANKS, "bad field / rank count");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && batch > 0 && batch % 4 == 0 && batch * world < (int64_t(1) << 31),
              "dim and batch must be multiples of 4");
  DFM_REQUIRE(aligned16(d_send), "send buffer must be 16-byte aligned");
  ShardTables st;
  memset(&st, 0, sizeof(st));
  for (int j = 0; j < num_owned; ++j) {
    DFM_REQUIRE(tables[j].w2 && tables[j].w1 && vocab[j] > 0, "table %d: null weights / bad vocabulary", j);
    st.w2[j] = tables[j].w2;
    st.w1[j] = tables[j].w1;
    st.stride2[j] = tables[j].stride2 ? tables[j].stride2 : dim;
    st.stride1[j] = tables[j].stride1 ? tables[j].stride1 : 1;
    st.vocab[j] = vocab[j];
    DFM_REQUIRE(st.stride2[j] % 4 == 0 && aligned16(st.w2[j]), "table %d: rows must be 16-byte aligned", j);
  }
  const int64_t threads = static_cast<int64_t>(world) * batch * num_owned * (dim / 4);
  hipLaunchKernelGGL(shard_gather_kernel, dim3(static_cast<unsigned>((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     as_stream(stream), st, num_owned, dim, world, batch, d_ids, d_send, d_gids, d_error_flag);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int64_t dfm_shard_pack_segment(int64_t batch, int num_owned, int dim, int64_t n_dense) {
  return batch * num_owned * dim + batch + n_dense;
}

extern "C" int dfm_shard_pack(const int32_t* first_field, const int32_t* field_count, int world,
                              const int32_t* field_of_sparse, int num_sparse, int num_fields, int dim, int64_t batch,
                              const float* d_g_field, const float* d_g_first, const float* d_dense, int64_t n_dense,
                              const dfm_slab_ref* slabs, int num_slabs, float* d_send, dfm_stream_t stream) {
  DFM_REQUIRE(first_field && field_count && field_of_sparse && d_g_field && d_g_first && d_dense && d_send, "null argument");
  DFM_REQUIRE(world > 0 && world <= DFM_MAX_RANKS && num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS &&
                  num_fields <= DFM_MAX_FIELDS, "bad field / rank count");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0 && batch > 0 && batch % 4 == 0 && n_dense >= 0 && n_dense % 4 == 0,
              "dim, batch and the dense size must be multiples of 4");
  DFM_REQUIRE(aligned16(d_g_field) && aligned16(d_g_first) && aligned16(d_dense) && aligned16(d_send),
              "16-byte aligned buffers only");
  PackPlan pp;
  memset(&pp, 0, sizeof(pp));
  pp.world = world;
  for (int s = 0; s < num_sparse; ++s) {
    DFM_REQUIRE(field_of_sparse[s] >= 0 && field_of_sparse[s] < num_fields, "field_of_sparse[%d] out of range", s);
    pp.field_of_sparse[s] = field_of_sparse[s];
  }
  for (int q = 0; q < world; ++q) {
    DFM_REQUIRE(first_field[q] >= 0 && field_count[q] > 0 && first_field[q] + field_count[q] <= num_sparse,
                "rank %d: owned fields outside [0, %d)", q, num_sparse);
    pp.first[q] = first_field[q];
    pp.count[q] = field_count[q];
    pp.seg_start[q + 1] = pp.seg_start[q] + dfm_shard_pack_segment(batch, field_count[q], dim, n_dense);
  }
  SlabTable st;
  if (int rc = fill_slab_table(slabs, num_slabs, d_dense, n_dense, &st)) return rc;
  DFM_REQUIRE(pp.first[0] == 0, "rank 0 owns the first SPARSE fields");
  for (int q = 1; q < world; ++q)
    DFM_REQUIRE(pp.first[q] == pp.first[q - 1] + pp.count[q - 1], "the ranks' field blocks must be contiguous and in rank order");
  DFM_REQUIRE(pp.first[world - 1] + pp.count[world - 1] == num_sparse, "the ranks' field blocks must cover every SPARSE field");
  const int64_t threads = batch * num_sparse * dim / 4 + batch / 4 * world + n_dense / 4;
  hipLaunchKernelGGL(shard_pack_kernel, dim3(static_cast<unsigned>((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     as_stream(stream), pp, batch, num_fields, dim, d_g_field, d_g_first, d_dense, n_dense, st, d_send);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_shard_rowgrad(int num_owned, int dim, int world, int64_t batch, const float* d_recv,
                                 int64_t segment, const int32_t* d_sorted_pos, int32_t* d_seg_start,
                                 const int32_t* d_num_uniq, float* d_row_g2, float* d_row_g1, dfm_stream_t stream) {
  DFM_REQUIRE(d_recv && d_sorted_pos && d_seg_start && d_num_uniq && d_row_g2 && d_row_g1, "null argument");
  DFM_REQUIRE(num_owned > 0 && num_owned <= DFM_MAX_FIELDS && world > 0 && dim > 0 && dim % 4 == 0 && batch > 0,
              "bad sizes");
  DFM_REQUIRE(segment >= batch * num_owned * dim + batch && segment % 4 == 0 && aligned16(d_recv),
              "segment too short / not 16-byte aligned");
  FieldMap fm;
  memset(&fm, 0, sizeof(fm));
  for (int j = 0; j < num_owned; ++j) fm.f[j] = j;
  const int64_t n = batch * world;
  const int lists = static_cast<int>((n + CH - 1) / CH) * num_owned;
  const int64_t threads = static_cast<int64_t>(lists) * CH * (dim / 4);
  hipLaunchKernelGGL(shard_rowgrad_kernel, dim3(static_cast<unsigned>((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     as_stream(stream), fm, num_owned, dim, lists, d_recv + batch * num_owned * dim, d_recv, d_sorted_pos,
                     d_seg_start, d_num_uniq, d_row_g2, d_row_g1, SampleSegments{batch, segment});
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_sum_floats(const float* d_x, int64_t n, float* d_out, dfm_stream_t stream) {
  DFM_REQUIRE(d_x && d_out && n >= 0 && n < (int64_t(1) << 31), "bad arguments");
  hipLaunchKernelGGL(sum_floats_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d_x, static_cast<int>(n), d_out);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}
