// Row plan: per SPARSE field, the batch's ids sorted into runs of equal id — the
// deterministic, atomic-free form of the embedding backward scatter-add
// (reference: autograd of embedding.py:95-98 = aten::embedding_dense_backward, which
// accumulates duplicate ids; row 0 = padding_idx gets no gradient).
//
// One 1024-thread workgroup sorts one chunk of up to 4096 ids of one field entirely in
// LDS (keys id<<shift | position: keys are distinct, so the order is "by id, then by sample
// position" = stable; a counting sort over 4096 id buckets for spread-out ids, a ballot-counting
// radix sort for skewed ones).  Ids 0 / out of range become a sentinel that sorts to the end.  Run heads are found with a block scan.  The plan depends on the
// ids only, so the host can enqueue it ahead of the forward pass.
//
// rowgrad: D/4 lanes per distinct id add that id's contributions in sorted (= sample)
// order and write one gradient row: plain 16-byte stores, bitwise reproducible.
#include "rowplan_body.h"

#include <cstdlib>

using namespace dfm;

namespace {
using tail::CH;
using rowplan::SORT_THREADS;
using rowplan::kTouchParts;

struct IdTable {
  const int64_t* p[DFM_MAX_FIELDS];
  int32_t vocab[DFM_MAX_FIELDS];
};
using tail::FieldMap;
// Row touch (round 3): extra workgroups of the row-plan launch (blockIdx.y >= chunks) read the chunk's ids and
// touch the first line of every row the batch will gather.  The plan sorts on 26 of 256 CUs for ~13 us; the
// other CUs pull the batch's 106 K table lines (13.6 MB out of 6.6 GB) and the id columns into the Infinity
// Cache meanwhile, so that the gather — launched right after, on the same record — finds its ids and rows
// on-die instead of paying three dependent HBM round trips.  Values are discarded; nothing is written.
struct TouchTable {
  const float* w2[DFM_MAX_FIELDS];
  int32_t stride2[DFM_MAX_FIELDS];
};
}  // namespace

template <typename KeyT, int SHIFT>
__global__ __launch_bounds__(SORT_THREADS) void rowplan_sort(
    IdTable ids, int S, int64_t n, int32_t* __restrict__ sorted_pos, int32_t* __restrict__ uniq_rows,
    int32_t* __restrict__ seg_start, int32_t* __restrict__ num_uniq, int32_t* error_flag, int ablate,
    TouchTable touch, int chunks) {
  const int s = blockIdx.x;
  if (static_cast<int>(blockIdx.y) >= chunks) {
    const int q = static_cast<int>(blockIdx.y) - chunks;
    rowplan::rowplan_touch_body(ids.p[s], ids.vocab[s], touch.w2[s], touch.stride2[s], q / kTouchParts, q % kTouchParts, n);
    return;
  }
  rowplan::rowplan_chunk_body<KeyT, SHIFT>(ids.p[s], ids.vocab[s], s, blockIdx.y, S, n, sorted_pos, uniq_rows, seg_start,
                                           num_uniq, error_flag, ablate);
}

__global__ __launch_bounds__(256) void rowgrad_kernel(
    FieldMap fmap, int S, int F, int D, int lists, const float* __restrict__ g_first,
    const float* __restrict__ g_field, const int32_t* __restrict__ sorted_pos,
    int32_t* seg_start, const int32_t* __restrict__ num_uniq,
    float* __restrict__ row_g2, float* __restrict__ row_g1) {
  tail::rowgrad_body(blockIdx.x, fmap, S, F, D, lists, g_first, g_field, sorted_pos, seg_start, num_uniq, row_g2, row_g1);
}

// timing-only ablation (0 in every product call): 1 = skip the sort, 2 = skip the run-head / output phase
// timing-only ablation of the sort (tools/rp_ablate.sh): exists only in a -DDFM_TUNING_ABLATE=<mask> build,
// never in the shipped library
#ifndef DFM_TUNING_ABLATE
#define DFM_TUNING_ABLATE 0
#endif
static constexpr int g_rp_ablate = DFM_TUNING_ABLATE;

extern "C" {

// One row-plan launch, fully described (kernel, geometry, argument values): goes to the stream, or rewrites the
// kernel node of an instantiated graph (dfm_rowplan_build_update), like the gather's GatherLaunch.
struct RowplanLaunch {
  const void* func = nullptr;
  dim3 grid, block;
  unsigned lds = 0;
  IdTable ids;
  TouchTable touch;
  int S = 0, ablate = 0, chunks = 0;
  int64_t n = 0;
  int32_t *sorted_pos = nullptr, *uniq_rows = nullptr, *seg_start = nullptr, *num_uniq = nullptr, *err = nullptr;
  void* params[12];
  void bind() {
    int k = 0;
    params[k++] = &ids; params[k++] = &S; params[k++] = &n; params[k++] = &sorted_pos; params[k++] = &uniq_rows;
    params[k++] = &seg_start; params[k++] = &num_uniq; params[k++] = &err; params[k++] = &ablate;
    params[k++] = &touch; params[k++] = &chunks;
  }
};

static int describe_rowplan(const int64_t* const* ids, const int32_t* vocab, int num_sparse, int64_t n,
                            int32_t* d_sorted_pos, int32_t* d_uniq_rows, int32_t* d_seg_start, int32_t* d_num_uniq,
                            int32_t* d_error_flag, const dfm_table* touch_tables, int dim, RowplanLaunch* r) {
  DFM_REQUIRE(ids && vocab && d_sorted_pos && d_uniq_rows && d_seg_start && d_num_uniq, "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS, "num_sparse %d outside [1, %d]", num_sparse, DFM_MAX_FIELDS);
  DFM_REQUIRE(n > 0 && n < (int64_t(1) << 31), "n out of range");
  memset(&r->ids, 0, sizeof(r->ids));
  memset(&r->touch, 0, sizeof(r->touch));
  for (int s = 0; s < num_sparse; ++s) {
    DFM_REQUIRE(ids[s] != nullptr && vocab[s] > 0, "sparse field %d: bad ids/vocab", s);
    r->ids.p[s] = ids[s];
    r->ids.vocab[s] = vocab[s];
    if (touch_tables) {
      DFM_REQUIRE(touch_tables[s].w2 && dim > 0, "row touch: table %d has no weights", s);
      r->touch.w2[s] = touch_tables[s].w2;
      r->touch.stride2[s] = touch_tables[s].stride2 ? touch_tables[s].stride2 : dim;
    }
  }
  const int chunks = static_cast<int>((n + CH - 1) / CH);
  int max_vocab = 0;
  for (int s = 0; s < num_sparse; ++s) max_vocab = vocab[s] > max_vocab ? vocab[s] : max_vocab;
  static_assert(CH == 4096, "the 32-bit key packs the position into 12 bits");
  // ids (< vocab) below 2^20 - 1 leave the all-ones 32-bit key free for the sentinel
  const bool narrow = max_vocab < (1 << 20) - 1;
  r->func = narrow ? reinterpret_cast<const void*>(rowplan_sort<uint32_t, 12>)
                   : reinterpret_cast<const void*>(rowplan_sort<unsigned long long, 32>);
  r->lds = rowplan::lds_bytes(narrow);
  static bool allowed[2] = {false, false};          // more than 64 KB of dynamic LDS: allowed once per kernel
  if (!allowed[narrow ? 0 : 1]) {
    DFM_HIP_TRY(hipFuncSetAttribute(r->func, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(r->lds)));
    allowed[narrow ? 0 : 1] = true;
  }
  r->grid = dim3(num_sparse, chunks * (touch_tables ? 1 + kTouchParts : 1));
  r->block = dim3(SORT_THREADS);
  r->S = num_sparse; r->n = n; r->chunks = chunks; r->ablate = g_rp_ablate;
  r->sorted_pos = d_sorted_pos; r->uniq_rows = d_uniq_rows; r->seg_start = d_seg_start; r->num_uniq = d_num_uniq;
  r->err = d_error_flag;
  r->bind();
  return DFM_OK;
}

int dfm_rowplan_build(const int64_t* const* ids, const int32_t* vocab, int num_sparse, int64_t n,
                      int32_t* d_sorted_pos, int32_t* d_uniq_rows, int32_t* d_seg_start,
                      int32_t* d_num_uniq, int32_t* d_error_flag, const dfm_table* touch_tables, int dim,
                      dfm_stream_t stream) {
  RowplanLaunch r;
  if (int rc = describe_rowplan(ids, vocab, num_sparse, n, d_sorted_pos, d_uniq_rows, d_seg_start, d_num_uniq,
                                d_error_flag, touch_tables, dim, &r)) return rc;
  DFM_HIP_TRY(hipLaunchKernel(r.func, r.grid, r.block, r.params, r.lds, as_stream(stream)));
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// The row plan was captured into a graph (dfm_graph_last_node right after dfm_rowplan_build returns its node):
// point the node of the INSTANTIATED graph at other id columns (the next batch record).  Host-side only;
// same rules as dfm_embedding_forward_staged_update.
int dfm_rowplan_build_update(void* graph_exec, void* node, const int64_t* const* ids, const int32_t* vocab,
                             int num_sparse, int64_t n, int32_t* d_sorted_pos, int32_t* d_uniq_rows,
                             int32_t* d_seg_start, int32_t* d_num_uniq, int32_t* d_error_flag,
                             const dfm_table* touch_tables, int dim) {
  DFM_REQUIRE(graph_exec && node, "null argument");
  RowplanLaunch r;
  if (int rc = describe_rowplan(ids, vocab, num_sparse, n, d_sorted_pos, d_uniq_rows, d_seg_start, d_num_uniq,
                                d_error_flag, touch_tables, dim, &r)) return rc;
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = const_cast<void*>(r.func);
  p.gridDim = r.grid;
  p.blockDim = r.block;
  p.sharedMemBytes = r.lds;
  p.kernelParams = r.params;
  p.extra = nullptr;
  DFM_HIP_TRY(hipGraphExecKernelNodeSetParams(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipGraphNode_t>(node), &p));
  return DFM_OK;
}

int dfm_rowgrad_build(const int32_t* field_of_sparse, int num_sparse, int num_fields, int dim,
                      int64_t n, const float* d_g_first, const float* d_g_field,
                      const int32_t* d_sorted_pos, int32_t* d_seg_start,
                      const int32_t* d_num_uniq, float* d_row_g2, float* d_row_g1,
                      dfm_stream_t stream) {
  DFM_REQUIRE(field_of_sparse && d_g_first && d_g_field && d_sorted_pos && d_seg_start && d_num_uniq &&
              d_row_g2 && d_row_g1, "null argument");
  DFM_REQUIRE(num_sparse > 0 && num_sparse <= DFM_MAX_FIELDS, "num_sparse out of range");
  DFM_REQUIRE(dim > 0 && dim % 4 == 0, "dim must be a multiple of 4");
  DFM_REQUIRE(n > 0, "n must be positive");
  FieldMap fm;
  memset(&fm, 0, sizeof(fm));
  for (int s = 0; s < num_sparse; ++s) {
    DFM_REQUIRE(field_of_sparse[s] >= 0 && field_of_sparse[s] < num_fields, "field_of_sparse[%d] out of range", s);
    fm.f[s] = field_of_sparse[s];
  }
  const int chunks = static_cast<int>((n + CH - 1) / CH);
  const int lists = chunks * num_sparse;
  const int64_t threads = static_cast<int64_t>(lists) * CH * (dim / 4);
  hipLaunchKernelGGL(rowgrad_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                     as_stream(stream), fm, num_sparse, num_fields, dim, lists, d_g_first, d_g_field,
                     d_sorted_pos, d_seg_start, d_num_uniq, d_row_g2, d_row_g1);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

}  // extern "C"
