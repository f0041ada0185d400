// FeatureEmbedding forward / dense-gradient backward for gfx950.
//
// Reference semantics: deepfm/models/layers/embedding.py:76-126 (forward) and its
// autograd (dense V x d gradients, nn.Embedding(sparse=False), embedding.py:35-40).
//
// Two forward paths:
//   * emb_fwd_uniform<D,W,US,UD>  — every field SPARSE or DENSE with dim == fm_dim == D and no
//     projection (the Criteo shape).  ONE launch gathers all tables: a workgroup of W waves owns
//     64/(D/4) consecutive samples, a wave US sparse + UD dense fields of them per pass, so ids are
//     read as one coalesced line and rows as 16-byte pieces, ALL loads of a pass in flight at once
//     (the kernel is one dependent chain kernarg -> ids -> rows -> stores; round-2 stamps:
//     profiles/r02_gather_microbench.txt).
//     first_order and the FM value are reduced across the block's waves through LDS in a fixed
//     order (W == 1: no LDS, no barrier).
//   * emb_fwd_general — any schema (mixed dims, projections, SEQUENCE bags): one thread
//     per (sample, field).  Correctness path for MovieLens-shaped schemas.
#include "tail_bodies.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <vector>

using namespace dfm;


struct dfm_embedding_plan {
  int num_fields = 0;
  int fm_dim = 0;
  int total_dim = 0;
  int uniform = 0;
  int max_dim = 0;
  std::vector<dfm_field> h_fields;
  std::vector<int32_t> h_sparse, h_dense, h_proj;
  dfm_field* d_fields = nullptr;
  int32_t* d_sparse = nullptr;
  int32_t* d_dense = nullptr;
  int32_t* d_proj = nullptr;
};

// ======================================================================================
// uniform fused gather
// ======================================================================================
// Per-call slot tables travel BY VALUE in the kernel-argument segment: a wave's slot index
// depends only on its wave id, so the slot (pointers + vocab) is one scalar load issued at
// kernel entry — the dependent chain is kernarg -> ids -> rows, nothing else.
struct SparseSlot {
  const int64_t* ids;
  const float* w2;
  const float* w1;
  int32_t vocab;
  int32_t field;
  int32_t stride2;  // floats between rows of w2 / w1 (packed row records: the record size)
  int32_t stride1;
  int64_t* ids_out;  // optional: the ids are also copied here (staging of the step's static inputs)
};
struct DenseSlot {
  const float* x;
  const float* w2;
  const float* b2;
  const float* w1;
  const float* b1;
  int32_t field;
  int32_t pad;
  float* x_out;      // optional: copy of x (staging)
};
constexpr int kMaxSparseSlots = 48, kMaxDenseSlots = 32;   // 48*48 + 32*56 B of kernel arguments
struct UniformArgs {
  SparseSlot sp[kMaxSparseSlots];
  DenseSlot de[kMaxDenseSlots];
};

typedef float v4f __attribute__((ext_vector_type(4)));
// Streaming (nt) stores of field_embeddings: -0.8 us for the isolated kernel (tools/microbench_gather2), but
// inside the training step they cost 0.4-1.2 us and 3.7 MB of extra write requests (the consumer is the next
// kernel of the graph) — the product kernel stores plainly; shape 6 keeps the nt variant for the tools.
__device__ __forceinline__ void st4_stream(float* p, const float4& v) {
  v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

template <int D, int W, int US, int UD>
__device__ __forceinline__ void emb_fwd_uniform_body(
    const UniformArgs& args, int ns, int nd, int64_t B, int F, float* __restrict__ first_order,
    float* __restrict__ fe, float* __restrict__ fm_out, float* __restrict__ fm_sum, int32_t* error_flag,
    const float* __restrict__ extra_src = nullptr, float* __restrict__ extra_dst = nullptr) {
  constexpr int LPR = D / 4;        // lanes per row (16 B each)
  constexpr int SPW = kWave / LPR;  // samples per wave == samples per block
  const int lane = lane_id();
  const int wave = W == 1 ? 0 : wave_id_uniform();
  const int s = lane / LPR, q = lane % LPR;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * SPW + s;
  const bool live = b < B;
  const int64_t bc = live ? b : B - 1;  // clamped: dead lanes load valid addresses

  float4 S = make_float4(0.f, 0.f, 0.f, 0.f);   // sum_f e
  float4 SQ = make_float4(0.f, 0.f, 0.f, 0.f);  // sum_f e^2
  float fo = 0.f;
  bool bad = false;

  // One straight-line block per pass: every slot, then every id / dense value, then every row, then
  // the arithmetic, and only then the (predicated) stores — so no branch sits between a load and
  // its first use and the waits stay counted, not vmcnt(0).  Slots past the end are clamped to
  // slot 0 (a duplicate, cache-hitting load) and masked.
  const int sp_iters = US ? (ns + W * US - 1) / (W * US) : 0;
  const int de_iters = UD ? (nd + W * UD - 1) / (W * UD) : 0;
  // W == 1 (the host launches it only when one pass covers the plan): slot indices are compile-time
  // constants, so every slot field is an immediate-offset scalar load from the kernel arguments
  const int iters = W == 1 ? 1 : (sp_iters > de_iters ? sp_iters : de_iters);
  for (int it = 0; it < iters; ++it) {
    bool oks[US + 1], okd[UD + 1];
    SparseSlot sl[US + 1];
    DenseSlot dl[UD + 1];
#pragma unroll
    for (int u = 0; u < US; ++u) {
      const int i = W == 1 ? u : wave + (it * US + u) * W;
      oks[u] = i < ns;
      sl[u] = args.sp[oks[u] ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const int i = W == 1 ? u : wave + (it * UD + u) * W;
      okd[u] = i < nd;
      dl[u] = args.de[okd[u] ? i : 0];
    }
    int64_t id[US + 1];
    float x[UD + 1];
#pragma unroll
    for (int u = 0; u < US; ++u) id[u] = sl[u].ids[bc];
#pragma unroll
    for (int u = 0; u < UD; ++u) x[u] = dl[u].x[bc];
    // staging: the raw inputs also go to the step's static buffers (row plan, embedding backward)
#pragma unroll
    for (int u = 0; u < US; ++u)
      if (sl[u].ids_out && live && q == 0 && oks[u]) sl[u].ids_out[b] = id[u];
#pragma unroll
    for (int u = 0; u < UD; ++u)
      if (dl[u].x_out && live && q == 0 && okd[u]) dl[u].x_out[b] = x[u];
    float4 row[US + 1];
    float w1v[US + 1];
#pragma unroll
    for (int u = 0; u < US; ++u) {
      const bool oob = static_cast<uint64_t>(id[u]) >= static_cast<uint64_t>(sl[u].vocab);
      bad |= oob && oks[u];
      id[u] = oob ? 0 : id[u];
      row[u] = ld4(sl[u].w2 + id[u] * sl[u].stride2 + q * 4);
      w1v[u] = sl[u].w1[id[u] * sl[u].stride1];   // same address on the row's lanes: one request
    }
    float4 dw[UD + 1], db[UD + 1];
    float dw1[UD + 1], db1[UD + 1];
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      dw[u] = ld4(dl[u].w2 + q * 4);
      db[u] = ld4(dl[u].b2 + q * 4);
      dw1[u] = dl[u].w1[0];
      db1[u] = dl[u].b1[0];
    }
    float* out = fe + b * F * D + q * 4;
#pragma unroll
    for (int u = 0; u < US; ++u) {
      const float m = oks[u] ? 1.f : 0.f;
      const float4 e = row[u];
      S.x = fmaf(m, e.x, S.x); S.y = fmaf(m, e.y, S.y); S.z = fmaf(m, e.z, S.z); S.w = fmaf(m, e.w, S.w);
      SQ.x = fmaf(m * e.x, e.x, SQ.x); SQ.y = fmaf(m * e.y, e.y, SQ.y);
      SQ.z = fmaf(m * e.z, e.z, SQ.z); SQ.w = fmaf(m * e.w, e.w, SQ.w);
      fo = fmaf(m, w1v[u], fo);
      if (live && oks[u]) st4(out + sl[u].field * D, e);
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const float m = okd[u] ? 1.f : 0.f;
      float4 e;
      e.x = fmaf(x[u], dw[u].x, db[u].x); e.y = fmaf(x[u], dw[u].y, db[u].y);
      e.z = fmaf(x[u], dw[u].z, db[u].z); e.w = fmaf(x[u], dw[u].w, db[u].w);
      S.x = fmaf(m, e.x, S.x); S.y = fmaf(m, e.y, S.y); S.z = fmaf(m, e.z, S.z); S.w = fmaf(m, e.w, S.w);
      SQ.x = fmaf(m * e.x, e.x, SQ.x); SQ.y = fmaf(m * e.y, e.y, SQ.y);
      SQ.z = fmaf(m * e.z, e.z, SQ.z); SQ.w = fmaf(m * e.w, e.w, SQ.w);
      fo = fmaf(m, fmaf(x[u], dw1[u], db1[u]), fo);
      if (live && okd[u]) st4(out + dl[u].field * D, e);
    }
  }
  if (bad && error_flag) atomicOr(error_flag, 1);
  if (q != 0) fo = 0.f;  // every lane of a row loaded the same first-order value: count it once
  float acc[9] = {S.x, S.y, S.z, S.w, SQ.x, SQ.y, SQ.z, SQ.w, fo};
  if (W > 1) {
    // ---- fixed-order reduction over the block's waves ------------------------------------
    __shared__ float red[W][9][kWave];
#pragma unroll
    for (int c = 0; c < 9; ++c) red[wave][c][lane] = acc[c];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = red[0][c][lane];
    for (int w = 1; w < W; ++w) {
#pragma unroll
      for (int c = 0; c < 9; ++c) acc[c] += red[w][c][lane];
    }
  }
  // 0.5 * sum_d (S_d^2 - SQ_d)   (fm.py:20-22)
  float t = (acc[0] * acc[0] - acc[4]) + (acc[1] * acc[1] - acc[5]) +
            (acc[2] * acc[2] - acc[6]) + (acc[3] * acc[3] - acc[7]);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, kWave);
  if (live && q == 0) {
    first_order[b] = acc[8];
    if (fm_out) fm_out[b] = 0.5f * t;
  }
  // S[b, :] = sum_f e[b, f, :] for the FM backward g * (S - e)
  if (live && fm_sum) st4(fm_sum + b * D + q * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
  if (live && q == 0 && extra_dst) extra_dst[b] = extra_src[b];      // per-sample payload (labels)
}

// --------------------------------------------------------------------------------------------------
// emb_fwd_pair<D, NS, ND, STAGE>: the same gather compiled for an EXACT field count (the Criteo shape:
// NS = 26 SPARSE + ND = 13 DENSE).  A workgroup of TWO waves owns 64/(D/4) samples; wave w takes the
// slots 2u + w.  Every slot index and both counts are compile-time constants: slot fields are
// immediate-offset scalar loads, there is no run-time selection, masking or branch between a load and
// its use (staging is a template flag), and all of a wave's loads are in flight in three rounds:
// ids + dense values -> rows + first-order scalars + dense weights -> math + streaming stores.
// Measured against the 8-wave shape on the same tables: profiles/r02_gather_shapes.csv.
template <int D, int NS, int ND, bool STAGE, int WAVE, bool NT>
__device__ __forceinline__ void emb_fwd_pair_wave(
    const UniformArgs& args, int64_t B, int F, float* __restrict__ fe, int32_t* error_flag, float (&acc)[9]) {
  constexpr int LPR = D / 4;
  constexpr int SPW = kWave / LPR;
  constexpr int HS = (NS - WAVE + 1) / 2, HD = (ND - WAVE + 1) / 2;   // this wave's slots: 2u + WAVE
  const int lane = lane_id();
  const int s = lane / LPR, q = lane % LPR;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * SPW + s;
  const bool live = b < B;
  const int64_t bc = live ? b : B - 1;
  int64_t id[HS + 1];
  float x[HD + 1];
#pragma unroll
  for (int u = 0; u < HS; ++u) id[u] = args.sp[2 * u + WAVE].ids[bc];
#pragma unroll
  for (int u = 0; u < HD; ++u) x[u] = args.de[2 * u + WAVE].x[bc];
  float4 row[HS + 1];
  float w1v[HS + 1];
  bool bad = false;
#pragma unroll
  for (int u = 0; u < HS; ++u) {
    const SparseSlot& sl = args.sp[2 * u + WAVE];
    const bool oob = static_cast<uint64_t>(id[u]) >= static_cast<uint64_t>(sl.vocab);
    bad |= oob;
    const int64_t i = oob ? 0 : id[u];
    row[u] = ld4(sl.w2 + i * sl.stride2 + q * 4);
    w1v[u] = sl.w1[i * sl.stride1];
  }
  float4 dw[HD + 1], db[HD + 1];
  float dw1[HD + 1], db1[HD + 1];
#pragma unroll
  for (int u = 0; u < HD; ++u) {
    const DenseSlot& dl = args.de[2 * u + WAVE];
    dw[u] = ld4(dl.w2 + q * 4);
    db[u] = ld4(dl.b2 + q * 4);
    dw1[u] = dl.w1[0];
    db1[u] = dl.b1[0];
  }
  if (STAGE && live && q == 0) {
#pragma unroll
    for (int u = 0; u < HS; ++u) args.sp[2 * u + WAVE].ids_out[b] = id[u];
#pragma unroll
    for (int u = 0; u < HD; ++u) args.de[2 * u + WAVE].x_out[b] = x[u];
  }
  float4 S = make_float4(0.f, 0.f, 0.f, 0.f), SQ = make_float4(0.f, 0.f, 0.f, 0.f);
  float fo = 0.f;
  float* out = fe + b * F * D + q * 4;
#pragma unroll
  for (int u = 0; u < HS; ++u) {
    const float4 e = row[u];
    S.x += e.x; S.y += e.y; S.z += e.z; S.w += e.w;
    SQ.x = fmaf(e.x, e.x, SQ.x); SQ.y = fmaf(e.y, e.y, SQ.y);
    SQ.z = fmaf(e.z, e.z, SQ.z); SQ.w = fmaf(e.w, e.w, SQ.w);
    fo += w1v[u];
    if (live) { if (NT) st4_stream(out + args.sp[2 * u + WAVE].field * D, e); else st4(out + args.sp[2 * u + WAVE].field * D, e); }
  }
#pragma unroll
  for (int u = 0; u < HD; ++u) {
    float4 e;
    e.x = fmaf(x[u], dw[u].x, db[u].x); e.y = fmaf(x[u], dw[u].y, db[u].y);
    e.z = fmaf(x[u], dw[u].z, db[u].z); e.w = fmaf(x[u], dw[u].w, db[u].w);
    S.x += e.x; S.y += e.y; S.z += e.z; S.w += e.w;
    SQ.x = fmaf(e.x, e.x, SQ.x); SQ.y = fmaf(e.y, e.y, SQ.y);
    SQ.z = fmaf(e.z, e.z, SQ.z); SQ.w = fmaf(e.w, e.w, SQ.w);
    fo += fmaf(x[u], dw1[u], db1[u]);
    if (live) { if (NT) st4_stream(out + args.de[2 * u + WAVE].field * D, e); else st4(out + args.de[2 * u + WAVE].field * D, e); }
  }
  if (bad && error_flag) atomicOr(error_flag, 1);
  if (q != 0) fo = 0.f;
  acc[0] = S.x; acc[1] = S.y; acc[2] = S.z; acc[3] = S.w;
  acc[4] = SQ.x; acc[5] = SQ.y; acc[6] = SQ.z; acc[7] = SQ.w; acc[8] = fo;
}

template <int D, int NS, int ND, bool STAGE, bool NT = false>
__global__ __launch_bounds__(128) void emb_fwd_pair(
    UniformArgs args, int64_t B, int F, float* __restrict__ first_order, float* __restrict__ fe,
    float* __restrict__ fm_out, float* __restrict__ fm_sum, int32_t* error_flag,
    const float* __restrict__ extra_src, float* __restrict__ extra_dst) {
  constexpr int LPR = D / 4;
  constexpr int SPW = kWave / LPR;
  __shared__ float red[9][kWave];
  const int lane = lane_id();
  float acc[9];
  if (wave_id_uniform() == 1) {
    emb_fwd_pair_wave<D, NS, ND, STAGE, 1, NT>(args, B, F, fe, error_flag, acc);
#pragma unroll
    for (int c = 0; c < 9; ++c) red[c][lane] = acc[c];
    __syncthreads();
    return;
  }
  emb_fwd_pair_wave<D, NS, ND, STAGE, 0, NT>(args, B, F, fe, error_flag, acc);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] += red[c][lane];      // fixed order: wave 0 + wave 1
  const int s = lane / LPR, q = lane % LPR;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * SPW + s;
  const bool live = b < B;
  float t = (acc[0] * acc[0] - acc[4]) + (acc[1] * acc[1] - acc[5]) +
            (acc[2] * acc[2] - acc[6]) + (acc[3] * acc[3] - acc[7]);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, kWave);
  if (live && q == 0) {
    first_order[b] = acc[8];
    if (fm_out) fm_out[b] = 0.5f * t;
    if (STAGE && extra_dst) extra_dst[b] = extra_src[b];
  }
  if (live && fm_sum) st4(fm_sum + b * D + q * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// slot tables in the kernel-argument segment (first touch measured at 0-40 ns: the segment is hot
// when the waves start)
template <int D, int W, int US, int UD>
__global__ __launch_bounds__(W * 64) void emb_fwd_uniform(
    UniformArgs args, int ns, int nd, int64_t B, int F, float* __restrict__ first_order,
    float* __restrict__ fe, float* __restrict__ fm_out, float* __restrict__ fm_sum, int32_t* error_flag,
    const float* __restrict__ extra_src, float* __restrict__ extra_dst) {
  emb_fwd_uniform_body<D, W, US, UD>(args, ns, nd, B, F, first_order, fe, fm_out, fm_sum, error_flag, extra_src, extra_dst);
}

// ======================================================================================
// general path
// ======================================================================================
__device__ __forceinline__ float bag_pool(const float* __restrict__ table, int stride, int j,
                                          const int64_t* __restrict__ ids, int L, int vocab,
                                          int combiner, int32_t* error_flag) {
  float acc = 0.f;
  int count = 0;
  bool first = true;
  for (int l = 0; l < L; ++l) {
    const int64_t id = checked_id(ids[l], vocab, error_flag);
    if (id == 0) continue;
    const float v = table[id * stride + j];
    if (combiner == DFM_MAX) {
      if (first || v > acc) acc = v;
      first = false;
    } else {
      acc += v;
    }
    ++count;
  }
  if (combiner == DFM_MEAN && count > 0) acc = acc / static_cast<float>(count);
  return acc;
}

__global__ void emb_fwd_general(const dfm_field* __restrict__ fields, PtrTable in, int64_t B, int F,
                                int fm_dim, int total_dim, float* __restrict__ fo_parts,
                                float* __restrict__ fe, float* __restrict__ flat,
                                int32_t* error_flag) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= B * F) return;
  const int f = static_cast<int>(t % F);
  const int64_t b = t / F;
  const dfm_field fd = fields[f];
  const int d = fd.dim;
  float* flat_row = flat + b * total_dim + fd.flat_offset;
  float* fe_row = fe + (b * F + f) * fm_dim;
  float fo;
  if (fd.kind == DFM_SPARSE) {
    const int64_t id = checked_id(static_cast<const int64_t*>(in.p[f])[b], fd.vocab, error_flag);
    const float* row = fd.w2 + id * fd.stride2;
    for (int j = 0; j < d; ++j) flat_row[j] = row[j];
    fo = fd.w1[id * fd.stride1];
  } else if (fd.kind == DFM_DENSE) {
    const float x = static_cast<const float*>(in.p[f])[b];
    for (int j = 0; j < d; ++j) flat_row[j] = fmaf(x, fd.w2[j], fd.b2[j]);
    fo = fmaf(x, fd.w1[0], fd.b1[0]);
  } else {
    const int64_t* ids = static_cast<const int64_t*>(in.p[f]) + b * fd.max_len;
    for (int j = 0; j < d; ++j)
      flat_row[j] = bag_pool(fd.w2, fd.stride2, j, ids, fd.max_len, fd.vocab, fd.combiner, error_flag);
    fo = bag_pool(fd.w1, fd.stride1, 0, ids, fd.max_len, fd.vocab, fd.combiner, error_flag);
  }
  fo_parts[b * F + f] = fo;
  if (fd.proj) {
    for (int k = 0; k < fm_dim; ++k) {
      float acc = 0.f;
      for (int j = 0; j < d; ++j) acc = fmaf(flat_row[j], fd.proj[k * d + j], acc);
      fe_row[k] = acc;
    }
  } else {
    for (int j = 0; j < d; ++j) fe_row[j] = flat_row[j];
  }
}

__global__ void first_order_sum(const float* __restrict__ fo_parts, int64_t B, int F,
                                float* __restrict__ first_order) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int f = 0; f < F; ++f) acc += fo_parts[b * F + f];
  first_order[b] = acc;
}

// ======================================================================================
// dense-gradient backward
// ======================================================================================
// gradient w.r.t. the raw (pre-projection) embedding element j of (b, f)
__device__ __forceinline__ float raw_grad(const dfm_field& fd, int f, int64_t b, int j, int F,
                                          int fm_dim, int total_dim,
                                          const float* __restrict__ g_field,
                                          const float* __restrict__ g_flat) {
  float g = g_flat ? g_flat[b * total_dim + fd.flat_offset + j] : 0.f;
  const float* gf = g_field + (b * F + f) * fm_dim;
  if (fd.proj) {
    for (int k = 0; k < fm_dim; ++k) g = fmaf(gf[k], fd.proj[k * fd.dim + j], g);
  } else {
    g += gf[j];
  }
  return g;
}

// SPARSE + SEQUENCE rows: float atomics into the dense (V, d) gradients.
__global__ void emb_bwd_scatter(const dfm_field* __restrict__ fields, PtrTable in, GradTable gt,
                                int64_t B, int F, int fm_dim, int total_dim,
                                const float* __restrict__ g_first,
                                const float* __restrict__ g_field,
                                const float* __restrict__ g_flat) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= B * F) return;
  const int f = static_cast<int>(t % F);
  const int64_t b = t / F;
  const dfm_field fd = fields[f];
  if (fd.kind == DFM_DENSE) return;
  const int d = fd.dim;
  const dfm_field_grad g = gt.g[f];
  const float gfo = g_first[b];
  if (fd.kind == DFM_SPARSE) {
    int64_t id = static_cast<const int64_t*>(in.p[f])[b];
    if (id <= 0 || id >= fd.vocab) return;  // padding row: no gradient
    for (int j = 0; j < d; ++j)
      atomicAdd(g.w2 + id * d + j, raw_grad(fd, f, b, j, F, fm_dim, total_dim, g_field, g_flat));
    atomicAdd(g.w1 + id, gfo);
    return;
  }
  // SEQUENCE
  const int64_t* ids = static_cast<const int64_t*>(in.p[f]) + b * fd.max_len;
  const int L = fd.max_len;
  int count = 0;
  for (int l = 0; l < L; ++l) count += (ids[l] > 0 && ids[l] < fd.vocab) ? 1 : 0;
  if (count == 0) return;
  if (fd.combiner == DFM_MAX) {
    for (int j = 0; j <= d; ++j) {  // j == d: the (V,1) first-order table
      const float* table = j < d ? fd.w2 : fd.w1;
      const int stride = j < d ? fd.stride2 : fd.stride1, col = j < d ? j : 0;
      int64_t best = 0;
      float bestv = 0.f;
      for (int l = 0; l < L; ++l) {
        const int64_t id = ids[l];
        if (id <= 0 || id >= fd.vocab) continue;
        const float v = table[id * stride + col];
        if (best == 0 || v > bestv) { best = id; bestv = v; }
      }
      if (j < d)
        atomicAdd(g.w2 + best * d + j, raw_grad(fd, f, b, j, F, fm_dim, total_dim, g_field, g_flat));
      else
        atomicAdd(g.w1 + best, gfo);
    }
    return;
  }
  const float scale = fd.combiner == DFM_MEAN ? 1.f / static_cast<float>(count) : 1.f;
  for (int j = 0; j < d; ++j) {
    const float gj = raw_grad(fd, f, b, j, F, fm_dim, total_dim, g_field, g_flat) * scale;
    for (int l = 0; l < L; ++l) {
      const int64_t id = ids[l];
      if (id > 0 && id < fd.vocab) atomicAdd(g.w2 + id * d + j, gj);
    }
  }
  for (int l = 0; l < L; ++l) {
    const int64_t id = ids[l];
    if (id > 0 && id < fd.vocab) atomicAdd(g.w1 + id, gfo * scale);
  }
}

// Block-wide fixed-order sum of two values (256 threads).
using tail::block_sum2;

// DENSE fields: dW2[j] = sum_b x_b * g_raw[b,j], db2[j] = sum_b g_raw[b,j]; block (i, j);
// j == dim handles the first-order Linear(1,1).
__global__ __launch_bounds__(256) void emb_bwd_dense_fields(
    const dfm_field* __restrict__ fields, const int32_t* __restrict__ dense_list, PtrTable in,
    GradTable gt, int64_t B, int F, int fm_dim, int total_dim, const float* __restrict__ g_first,
    const float* __restrict__ g_field, const float* __restrict__ g_flat) {
  const int f = dense_list[blockIdx.x];
  const dfm_field fd = fields[f];
  const int j = blockIdx.y;
  if (j > fd.dim) return;
  const float* x = static_cast<const float*>(in.p[f]);
  float sw = 0.f, sb = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float g = j < fd.dim ? raw_grad(fd, f, b, j, F, fm_dim, total_dim, g_field, g_flat)
                               : g_first[b];
    sw = fmaf(x[b], g, sw);
    sb += g;
  }
  block_sum2(sw, sb);
  if (threadIdx.x == 0) {
    const dfm_field_grad g = gt.g[f];
    if (j < fd.dim) { g.w2[j] += sw; g.b2[j] += sb; }
    else            { g.w1[0] += sw; g.b1[0] += sb; }
  }
}

// Uniform plans (dim == fm_dim, no projection, g_flat folded into g_field): block (i, jq) sums 4
// columns of field i's gradient with 16-byte loads — the 4 column groups of a field walk the same
// 64-byte segments, so the second to fourth hit L2 — instead of one 4-byte load per line and column.
// jq == dim/4 handles the first-order Linear(1,1).
__global__ __launch_bounds__(256) void emb_bwd_dense_fields_uniform(
    const int32_t* __restrict__ dense_list, PtrTable in, GradTable gt, int64_t B, int F, int D,
    const float* __restrict__ g_first, const float* __restrict__ g_field) {
  tail::dense_fields_uniform_body(blockIdx.x, dense_list, in, gt, B, F, D, g_first, g_field);
}

// Projection gradient dP[k,j] = sum_b g_field[b,f,k] * raw[b,j]; block (i, k*max_dim + j).
__global__ __launch_bounds__(256) void emb_bwd_proj(
    const dfm_field* __restrict__ fields, const int32_t* __restrict__ proj_list, GradTable gt,
    int64_t B, int F, int fm_dim, int total_dim, int max_dim, const float* __restrict__ g_field,
    const float* __restrict__ flat_saved) {
  const int f = proj_list[blockIdx.x];
  const dfm_field fd = fields[f];
  const int k = blockIdx.y / max_dim, j = blockIdx.y % max_dim;
  if (j >= fd.dim) return;
  float acc = 0.f, unused = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 256)
    acc = fmaf(g_field[(b * F + f) * fm_dim + k], flat_saved[b * total_dim + fd.flat_offset + j], acc);
  block_sum2(acc, unused);
  if (threadIdx.x == 0) gt.g[f].proj[k * fd.dim + j] += acc;
}

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" int dfm_embedding_plan_create(const dfm_field* fields, int num_fields, int fm_dim,
                              dfm_embedding_plan** out_plan) {
  DFM_REQUIRE(fields && out_plan, "null argument");
  DFM_REQUIRE(num_fields > 0 && num_fields <= DFM_MAX_FIELDS,
              "num_fields %d outside [1, %d]", num_fields, DFM_MAX_FIELDS);
  DFM_REQUIRE(fm_dim > 0, "fm_dim must be positive");
  auto* plan = new dfm_embedding_plan();
  plan->num_fields = num_fields;
  plan->fm_dim = fm_dim;
  plan->h_fields.assign(fields, fields + num_fields);
  bool uniform = (fm_dim % 4 == 0) && (kWave % (fm_dim / 4) == 0) && fm_dim <= 256;
  int off = 0;
  for (int f = 0; f < num_fields; ++f) {
    dfm_field& fd = plan->h_fields[f];
    if (fd.dim <= 0 || !fd.w2 || !fd.w1 || (fd.kind == DFM_DENSE && (!fd.b2 || !fd.b1)) ||
        (fd.kind != DFM_DENSE && fd.vocab <= 0) || (fd.kind == DFM_SEQUENCE && fd.max_len <= 0) ||
        fd.kind < 0 || fd.kind > 2 || fd.combiner < 0 || fd.combiner > 2 ||
        ((fd.dim != fm_dim) != (fd.proj != nullptr))) {
      delete plan;
      return fail(DFM_ERR_INVALID, "field %d: inconsistent descriptor", f);
    }
    if (fd.stride2 == 0) fd.stride2 = fd.dim;
    if (fd.stride1 == 0) fd.stride1 = 1;
    if (fd.kind != DFM_DENSE && (fd.stride2 < fd.dim || fd.stride1 < 1 || (uniform && fd.stride2 % 4 != 0))) {
      delete plan;
      return fail(DFM_ERR_INVALID, "field %d: bad row strides", f);
    }
    fd.flat_offset = off;
    off += fd.dim;
    plan->max_dim = fd.dim > plan->max_dim ? fd.dim : plan->max_dim;
    if (fd.kind == DFM_SPARSE) plan->h_sparse.push_back(f);
    if (fd.kind == DFM_DENSE) plan->h_dense.push_back(f);
    if (fd.proj) plan->h_proj.push_back(f);
    if (fd.kind == DFM_SEQUENCE || fd.proj || fd.dim != fm_dim) uniform = false;
  }
  plan->total_dim = off;
  if (plan->h_sparse.size() > (size_t)kMaxSparseSlots || plan->h_dense.size() > (size_t)kMaxDenseSlots) uniform = false;
  plan->uniform = uniform ? 1 : 0;
  auto upload = [](const void* src, size_t bytes, void** dst) -> hipError_t {
    if (bytes == 0) { *dst = nullptr; return hipSuccess; }
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  hipError_t e = upload(plan->h_fields.data(), sizeof(dfm_field) * num_fields, (void**)&plan->d_fields);
  if (e == hipSuccess) e = upload(plan->h_sparse.data(), 4 * plan->h_sparse.size(), (void**)&plan->d_sparse);
  if (e == hipSuccess) e = upload(plan->h_dense.data(), 4 * plan->h_dense.size(), (void**)&plan->d_dense);
  if (e == hipSuccess) e = upload(plan->h_proj.data(), 4 * plan->h_proj.size(), (void**)&plan->d_proj);
  if (e != hipSuccess) {
    dfm_embedding_plan_destroy(plan);
    return fail(DFM_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
  }
  *out_plan = plan;
  return DFM_OK;
}

extern "C" int dfm_embedding_plan_destroy(dfm_embedding_plan* plan) {
  if (!plan) return DFM_OK;
  (void)hipFree(plan->d_fields);
  (void)hipFree(plan->d_sparse);
  (void)hipFree(plan->d_dense);
  (void)hipFree(plan->d_proj);
  delete plan;
  return DFM_OK;
}

extern "C" int dfm_embedding_plan_is_uniform(const dfm_embedding_plan* plan) { return plan ? plan->uniform : 0; }

extern "C" size_t dfm_embedding_workspace_bytes(const dfm_embedding_plan* plan, int64_t batch) {
  if (!plan || plan->uniform || batch <= 0) return 0;
  return sizeof(float) * static_cast<size_t>(batch) * plan->num_fields;  // fo_parts
}

static int fill_ptrs(const dfm_embedding_plan* plan, const void* const* inputs, PtrTable* t) {
  memset(t, 0, sizeof(*t));
  for (int f = 0; f < plan->num_fields; ++f) {
    DFM_REQUIRE(inputs[f] != nullptr, "input %d is null", f);
    t->p[f] = inputs[f];
  }
  return DFM_OK;
}

// Per-launch kernel timing for bench.py: when armed, the uniform gather is launched with
// hipExtLaunchKernelGGL, whose start/stop events are recorded by the command processor exactly
// around the dispatch (the same interval rocprofv3's kernel trace reports) instead of around the
// host-visible launch call.
namespace {
struct GatherTimer {
  std::vector<hipEvent_t> start, stop;
  int used = 0;
} g_gather_timer;
int g_gather_shape = 0;   // 0 = automatic; tools only (dfm_gather_set_shape)
}  // namespace

// Launch shapes (W waves per workgroup, US sparse + UD dense slots in flight per wave and pass):
//   1: W = 1, 26 + 13 — a lane group owns its sample across all fields: no LDS, no barrier
//   2: W = 2, 13 + 7
//   3: W = 4,  8 + 4
//   4: W = 8,  4 + 2  (round 1's shape; more slots than a pass holds simply take more passes)
//   5: emb_fwd_pair — two waves, compile-time field counts (26 SPARSE + 13 DENSE, D = 16 / 32)
//   6: shape 5 with streaming stores
// Automatic choice: 5 where it applies, else 4.  Shapes 1-3 exist for tools/time_gather.py.
extern "C" int dfm_gather_set_shape(int shape) {
  DFM_REQUIRE(shape >= 0 && shape <= 6, "gather shape %d outside [0, 6]", shape);
  g_gather_shape = shape;
  return DFM_OK;
}

// One gather launch, fully described: the kernel, its geometry and its argument values.  The same
// description either goes to the stream (optionally with start/stop events attached to the dispatch) or
// rewrites the kernel node of an instantiated graph (dfm_embedding_forward_staged_update).
struct GatherLaunch {
  const void* func = nullptr;
  dim3 grid, block;
  UniformArgs args;
  int ns = 0, nd = 0, F = 0;
  int64_t B = 0;
  float *fo = nullptr, *fe = nullptr, *fm_out = nullptr, *fm_sum = nullptr;
  int32_t* err = nullptr;
  const float* extra_src = nullptr;
  float* extra_dst = nullptr;
  bool pair = false;
  void* params[12];
  void bind() {
    int n = 0;
    params[n++] = &args;
    if (!pair) { params[n++] = &ns; params[n++] = &nd; }
    params[n++] = &B; params[n++] = &F; params[n++] = &fo; params[n++] = &fe; params[n++] = &fm_out;
    params[n++] = &fm_sum; params[n++] = &err; params[n++] = &extra_src; params[n++] = &extra_dst;
  }
};

template <int D>
static int describe_uniform(const dfm_embedding_plan* plan, const PtrTable& in, int64_t B,
                            float* fo, float* fe, float* fm_out, float* fm_sum, int32_t* err,
                            void* const* stage_out, const float* extra_src, float* extra_dst, GatherLaunch* g) {
  constexpr int SPW = kWave / (D / 4);
  memset(&g->args, 0, sizeof(g->args));
  const int ns = static_cast<int>(plan->h_sparse.size()), nd = static_cast<int>(plan->h_dense.size());
  for (int i = 0; i < ns; ++i) {
    const int f = plan->h_sparse[i];
    const dfm_field& fd = plan->h_fields[f];
    g->args.sp[i] = SparseSlot{static_cast<const int64_t*>(in.p[f]), fd.w2, fd.w1, fd.vocab, f, fd.stride2, fd.stride1,
                               stage_out ? static_cast<int64_t*>(stage_out[f]) : nullptr};
  }
  for (int i = 0; i < nd; ++i) {
    const int f = plan->h_dense[i];
    const dfm_field& fd = plan->h_fields[f];
    g->args.de[i] = DenseSlot{static_cast<const float*>(in.p[f]), fd.w2, fd.b2, fd.w1, fd.b1, f, 0,
                              stage_out ? static_cast<float*>(stage_out[f]) : nullptr};
  }
  g->ns = ns; g->nd = nd; g->B = B; g->F = plan->num_fields;
  g->fo = fo; g->fe = fe; g->fm_out = fm_out; g->fm_sum = fm_sum; g->err = err;
  g->extra_src = extra_src; g->extra_dst = extra_dst;
  g->grid = dim3(static_cast<unsigned>((B + SPW - 1) / SPW));
  int shape = g_gather_shape;
  // automatic: the exact-count two-wave kernel for the Criteo field counts (D = 16 / 32), else 8 waves
  const bool pair_ok = (D == 16 || D == 32) && ns == 26 && nd == 13;
  if (shape == 0) shape = pair_ok ? 5 : 4;
  if (shape == 5 && !pair_ok) shape = 4;
  if (shape == 6 && !pair_ok) shape = 4;
  g->pair = shape == 5 || shape == 6;
#define DFM_UNIFORM_KERNEL(WV, US_, UD_)                                               \
  do {                                                                                 \
    g->func = reinterpret_cast<const void*>(&emb_fwd_uniform<D, WV, US_, UD_>);        \
    g->block = dim3(WV * 64);                                                          \
  } while (0)
  switch (shape) {
    case 1: DFM_UNIFORM_KERNEL(1, 26, 13); break;
    case 2: DFM_UNIFORM_KERNEL(2, 13, 7); break;
    case 3: DFM_UNIFORM_KERNEL(4, 8, 4); break;
    case 5:
      if constexpr (D == 16 || D == 32) {
        g->func = stage_out ? reinterpret_cast<const void*>(&emb_fwd_pair<D, 26, 13, true>)
                            : reinterpret_cast<const void*>(&emb_fwd_pair<D, 26, 13, false>);
        g->block = dim3(128);
      }
      break;
    case 6:     // shape 5 with streaming (nt) stores of field_embeddings
      if constexpr (D == 16 || D == 32) {
        g->func = stage_out ? reinterpret_cast<const void*>(&emb_fwd_pair<D, 26, 13, true, true>)
                            : reinterpret_cast<const void*>(&emb_fwd_pair<D, 26, 13, false, true>);
        g->block = dim3(128);
      }
      break;
    default: DFM_UNIFORM_KERNEL(8, 4, 2);
  }
#undef DFM_UNIFORM_KERNEL
  g->bind();
  return DFM_OK;
}

static int describe_gather(const dfm_embedding_plan* plan, const PtrTable& in, int64_t B, float* fo, float* fe,
                           float* fm_out, float* fm_sum, int32_t* err, void* const* stage_out,
                           const float* extra_src, float* extra_dst, GatherLaunch* g) {
#define DFM_DESCRIBE(DD) \
  case DD: return describe_uniform<DD>(plan, in, B, fo, fe, fm_out, fm_sum, err, stage_out, extra_src, extra_dst, g)
  switch (plan->fm_dim) {
    DFM_DESCRIBE(4); DFM_DESCRIBE(8); DFM_DESCRIBE(16); DFM_DESCRIBE(32); DFM_DESCRIBE(64); DFM_DESCRIBE(128); DFM_DESCRIBE(256);
    default: break;
  }
#undef DFM_DESCRIBE
  return fail(DFM_ERR_UNSUPPORTED, "no uniform gather for fm_dim %d", plan->fm_dim);
}

static int launch_gather(GatherLaunch* g, hipStream_t st) {
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (g_gather_timer.used < static_cast<int>(g_gather_timer.start.size())) {
    ev0 = g_gather_timer.start[g_gather_timer.used];
    ev1 = g_gather_timer.stop[g_gather_timer.used];
    ++g_gather_timer.used;
  }
  if (ev0) DFM_HIP_TRY(hipExtLaunchKernel(g->func, g->grid, g->block, g->params, 0, st, ev0, ev1, 0));
  else     DFM_HIP_TRY(hipLaunchKernel(g->func, g->grid, g->block, g->params, 0, st));
  return DFM_OK;
}

extern "C" int dfm_embedding_forward(const dfm_embedding_plan* plan, const void* const* inputs, int64_t batch,
                          float* d_first_order, float* d_field_emb, float* d_flat_emb,
                          float* d_fm_out, float* d_fm_sum, void* d_workspace, int32_t* d_error_flag,
                          dfm_stream_t stream) {
  DFM_REQUIRE(plan && inputs && d_first_order && d_field_emb, "null argument");
  DFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch %lld out of range", (long long)batch);
  if (batch == 0) return DFM_OK;
  PtrTable in;
  if (int rc = fill_ptrs(plan, inputs, &in)) return rc;
  hipStream_t st = as_stream(stream);
  if (plan->uniform && (d_flat_emb == nullptr || d_flat_emb == d_field_emb)) {
    GatherLaunch g;
    if (int rc = describe_gather(plan, in, batch, d_first_order, d_field_emb, d_fm_out, d_fm_sum, d_error_flag,
                                 nullptr, nullptr, nullptr, &g)) return rc;
    return launch_gather(&g, st);
  }
  DFM_REQUIRE(d_fm_out == nullptr && d_fm_sum == nullptr, "fused FM outputs need a uniform plan");
  DFM_REQUIRE(d_flat_emb && d_flat_emb != d_field_emb, "general plan needs a separate flat_embeddings buffer");
  DFM_REQUIRE(d_workspace, "general plan needs workspace (dfm_embedding_workspace_bytes)");
  float* fo_parts = static_cast<float*>(d_workspace);
  const int64_t total = batch * plan->num_fields;
  hipLaunchKernelGGL(emb_fwd_general, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                     plan->d_fields, in, batch, plan->num_fields, plan->fm_dim, plan->total_dim,
                     fo_parts, d_field_emb, d_flat_emb, d_error_flag);
  DFM_LAUNCH_CHECK();
  hipLaunchKernelGGL(first_order_sum, dim3(static_cast<unsigned>((batch + 255) / 256)), dim3(256), 0, st,
                     fo_parts, batch, plan->num_fields, d_first_order);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

static int fill_grads(const dfm_embedding_plan* plan, const dfm_field_grad* grads, GradTable* gt,
                      bool dense_only) {
  memset(gt, 0, sizeof(*gt));
  for (int f = 0; f < plan->num_fields; ++f) {
    const dfm_field& fd = plan->h_fields[f];
    const dfm_field_grad& g = grads[f];
    if (fd.kind == DFM_DENSE) {
      DFM_REQUIRE(g.w2 && g.b2 && g.w1 && g.b1, "field %d: missing DENSE gradient buffer", f);
    } else if (!dense_only) {
      DFM_REQUIRE(g.w2 && g.w1, "field %d: missing table gradient buffer", f);
    }
    if (fd.proj && !dense_only) DFM_REQUIRE(g.proj, "field %d: missing projection gradient buffer", f);
    gt->g[f] = g;
  }
  return DFM_OK;
}

static int launch_dense_fields(const dfm_embedding_plan* plan, const PtrTable& in, const GradTable& gt,
                               int64_t batch, const float* g_first, const float* g_field,
                               const float* g_flat, hipStream_t st) {
  const int nd = static_cast<int>(plan->h_dense.size());
  if (nd == 0) return DFM_OK;
  if (plan->uniform && g_flat == nullptr && (reinterpret_cast<uintptr_t>(g_field) & 15) == 0) {
    hipLaunchKernelGGL(emb_bwd_dense_fields_uniform, dim3(nd * (plan->fm_dim / 4 + 1)), dim3(256), 0, st, plan->d_dense,
                       in, gt, batch, plan->num_fields, plan->fm_dim, g_first, g_field);
    DFM_LAUNCH_CHECK();
    return DFM_OK;
  }
  hipLaunchKernelGGL(emb_bwd_dense_fields, dim3(nd, plan->max_dim + 1), dim3(256), 0, st,
                     plan->d_fields, plan->d_dense, in, gt, batch, plan->num_fields, plan->fm_dim,
                     plan->total_dim, g_first, g_field, g_flat);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

extern "C" int dfm_embedding_backward_dense(const dfm_embedding_plan* plan, const void* const* inputs,
                                 int64_t batch, const float* d_g_first, const float* d_g_field,
                                 const float* d_g_flat, const dfm_field_grad* grads,
                                 const void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(plan && inputs && d_g_first && d_g_field && grads, "null argument");
  DFM_REQUIRE(plan->uniform || d_g_flat, "general plan needs d_g_flat");
  if (batch == 0) return DFM_OK;
  PtrTable in;
  GradTable gt;
  if (int rc = fill_ptrs(plan, inputs, &in)) return rc;
  if (int rc = fill_grads(plan, grads, &gt, false)) return rc;
  hipStream_t st = as_stream(stream);
  const int64_t total = batch * plan->num_fields;
  hipLaunchKernelGGL(emb_bwd_scatter, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                     plan->d_fields, in, gt, batch, plan->num_fields, plan->fm_dim, plan->total_dim,
                     d_g_first, d_g_field, d_g_flat);
  DFM_LAUNCH_CHECK();
  if (int rc = launch_dense_fields(plan, in, gt, batch, d_g_first, d_g_field, d_g_flat, st)) return rc;
  const int np = static_cast<int>(plan->h_proj.size());
  if (np > 0) {
    // projections need the forward's flat_embeddings (raw rows): passed through d_workspace
    DFM_REQUIRE(d_workspace, "projection gradients need the saved flat_embeddings in d_workspace");
    hipLaunchKernelGGL(emb_bwd_proj, dim3(np, plan->fm_dim * plan->max_dim), dim3(256), 0, st,
                       plan->d_fields, plan->d_proj, gt, batch, plan->num_fields, plan->fm_dim,
                       plan->total_dim, plan->max_dim, d_g_field, static_cast<const float*>(d_workspace));
    DFM_LAUNCH_CHECK();
  }
  return DFM_OK;
}

extern "C" int dfm_embedding_backward_dense_fields(const dfm_embedding_plan* plan, const void* const* inputs,
                                        int64_t batch, const float* d_g_first,
                                        const float* d_g_field, const float* d_g_flat,
                                        const dfm_field_grad* grads, dfm_stream_t stream) {
  DFM_REQUIRE(plan && inputs && d_g_first && d_g_field && grads, "null argument");
  if (batch == 0) return DFM_OK;
  PtrTable in;
  GradTable gt;
  if (int rc = fill_ptrs(plan, inputs, &in)) return rc;
  if (int rc = fill_grads(plan, grads, &gt, true)) return rc;
  return launch_dense_fields(plan, in, gt, batch, d_g_first, d_g_field, d_g_flat, as_stream(stream));
}


extern "C" int dfm_embedding_forward_staged(const dfm_embedding_plan* plan, const void* const* inputs,
                                           void* const* stage_out, const float* d_extra_src, float* d_extra_dst,
                                           int64_t batch, float* d_first_order, float* d_field_emb, float* d_fm_out,
                                           float* d_fm_sum, int32_t* d_error_flag, dfm_stream_t stream) {
  DFM_REQUIRE(plan && inputs && stage_out && d_first_order && d_field_emb, "null argument");
  DFM_REQUIRE(plan->uniform, "staged gather needs a uniform plan");
  DFM_REQUIRE((d_extra_src == nullptr) == (d_extra_dst == nullptr), "extra source and destination go together");
  DFM_REQUIRE(batch >= 0 && batch < (int64_t(1) << 31), "batch %lld out of range", (long long)batch);
  if (batch == 0) return DFM_OK;
  PtrTable in;
  if (int rc = fill_ptrs(plan, inputs, &in)) return rc;
  for (int f = 0; f < plan->num_fields; ++f)
    DFM_REQUIRE(stage_out[f] != nullptr && stage_out[f] != inputs[f], "stage_out[%d] must be a distinct buffer", f);
  GatherLaunch g;
  if (int rc = describe_gather(plan, in, batch, d_first_order, d_field_emb, d_fm_out, d_fm_sum, d_error_flag,
                               stage_out, d_extra_src, d_extra_dst, &g)) return rc;
  return launch_gather(&g, as_stream(stream));
}

// The staged gather was captured into a graph (dfm_graph_last_node right after the call returns its
// node): point the node of the INSTANTIATED graph at another batch record / other buffers.  Host-side
// only (hipGraphExecKernelNodeSetParams), nothing is enqueued; takes effect at the next launch of the
// exec.  The caller must not update an exec whose previous launch may still be pending (the training
// step alternates two execs for that reason).
extern "C" int dfm_embedding_forward_staged_update(const dfm_embedding_plan* plan, void* graph_exec, void* node,
                                                  const void* const* inputs, void* const* stage_out,
                                                  const float* d_extra_src, float* d_extra_dst, int64_t batch,
                                                  float* d_first_order, float* d_field_emb, float* d_fm_out,
                                                  float* d_fm_sum, int32_t* d_error_flag) {
  DFM_REQUIRE(plan && graph_exec && node && inputs && stage_out && d_first_order && d_field_emb, "null argument");
  DFM_REQUIRE(plan->uniform, "staged gather needs a uniform plan");
  DFM_REQUIRE(batch > 0 && batch < (int64_t(1) << 31), "batch %lld out of range", (long long)batch);
  PtrTable in;
  if (int rc = fill_ptrs(plan, inputs, &in)) return rc;
  GatherLaunch g;
  if (int rc = describe_gather(plan, in, batch, d_first_order, d_field_emb, d_fm_out, d_fm_sum, d_error_flag,
                               stage_out, d_extra_src, d_extra_dst, &g)) return rc;
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = const_cast<void*>(g.func);
  p.gridDim = g.grid;
  p.blockDim = g.block;
  p.sharedMemBytes = 0;
  p.kernelParams = g.params;
  p.extra = nullptr;
  DFM_HIP_TRY(hipGraphExecKernelNodeSetParams(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipGraphNode_t>(node), &p));
  return DFM_OK;
}

extern "C" int dfm_gather_timing_begin(int launches) {
  DFM_REQUIRE(launches > 0 && launches <= (1 << 20), "bad launch count");
  DFM_REQUIRE(g_gather_timer.start.empty(), "gather timing already armed");
  g_gather_timer.start.resize(launches);
  g_gather_timer.stop.resize(launches);
  for (int i = 0; i < launches; ++i) {
    DFM_HIP_TRY(hipEventCreate(&g_gather_timer.start[i]));
    DFM_HIP_TRY(hipEventCreate(&g_gather_timer.stop[i]));
  }
  g_gather_timer.used = 0;
  return DFM_OK;
}

extern "C" int dfm_gather_timing_end(float* h_us, int capacity, int* h_count) {
  DFM_REQUIRE(h_count, "null argument");
  const int n = g_gather_timer.used < capacity ? g_gather_timer.used : capacity;
  for (int i = 0; i < g_gather_timer.used; ++i) {
    DFM_HIP_TRY(hipEventSynchronize(g_gather_timer.stop[i]));
    float ms = 0.f;
    DFM_HIP_TRY(hipEventElapsedTime(&ms, g_gather_timer.start[i], g_gather_timer.stop[i]));
    if (i < n && h_us) h_us[i] = ms * 1e3f;
  }
  *h_count = n;
  for (size_t i = 0; i < g_gather_timer.start.size(); ++i) {
    (void)hipEventDestroy(g_gather_timer.start[i]);
    (void)hipEventDestroy(g_gather_timer.stop[i]);
  }
  g_gather_timer.start.clear();
  g_gather_timer.stop.clear();
  g_gather_timer.used = 0;
  return DFM_OK;
}
