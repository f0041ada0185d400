// Device bodies of the training step's tail kernels (row gradients, DENSE-field gradients, row-wise
// and dense Adam), shared by their stand-alone launches (rowplan.hip, embedding.hip, rowadam.hip,
// dense_opt.hip) and by the grouped launches of step_tail.hip, where independent kernels of the step
// share one dispatch (a dependent launch costs ~4.5 us on an MI355X whatever it does).
// Every body is written for 256-thread workgroups and takes its logical workgroup index `blk`.
#pragma once

#include "common.h"

namespace dfm {
namespace tail {

constexpr int CH = DFM_ROWPLAN_CHUNK;  // 4096
constexpr int kTailThreads = 256;

struct FieldMap {
  int32_t f[DFM_MAX_FIELDS];
};
struct TableArgs {
  dfm_table t[DFM_MAX_FIELDS];
};

// Batch-split d-weight products of dfm_linear_backward (tower.hip): `splits` slabs of `elems` floats
// each, to be added in slab order into the weight's view `g` of the flat dense gradient.
constexpr int kMaxSlabs = 16;
struct SlabTable {
  const float* slabs[kMaxSlabs];
  float* g[kMaxSlabs];
  int64_t elems[kMaxSlabs];
  int splits[kMaxSlabs];
  int count;
};

// gi += the slabs of the float4 at `gptr` (an address inside the flat gradient buffer; parameters start
// on 64-byte boundaries, so a float4 belongs to at most one slab-backed weight); true if any were added.
__device__ __forceinline__ bool add_slabs(float4& gi, const float* gptr, const SlabTable& st) {
  bool any = false;
  for (int r = 0; r < st.count; ++r) {
    const int64_t off = gptr - st.g[r];
    if (off >= 0 && off < st.elems[r]) {
      const float* sl = st.slabs[r] + off;
      const int64_t stride = st.elems[r];
      const int splits = st.splits[r];
      int q = 0;
      for (; q + 8 <= splits; q += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld4(sl + (q + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { gi.x += t[u].x; gi.y += t[u].y; gi.z += t[u].z; gi.w += t[u].w; }
      }
      for (; q < splits; ++q) {
        const float4 t = ld4(sl + q * stride);
        gi.x += t.x; gi.y += t.y; gi.z += t.z; gi.w += t.w;
      }
      any = true;
    }
  }
  return any;
}

// fixed-order sum of `sq` over the workgroup's 4 waves -> partial[blk]
__device__ __forceinline__ void block_partial(float sq, float* __restrict__ partial, int blk) {
  __shared__ float wsum[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, kWave);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[blk] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__device__ __forceinline__ void block_sum2(float& a, float& b2) {
  __shared__ float sa[4], sb[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m, kWave);
    b2 += __shfl_xor(b2, m, kWave);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 0) { sa[w] = a; sb[w] = b2; }
  __syncthreads();
  a = sa[0] + sa[1] + sa[2] + sa[3];
  b2 = sb[0] + sb[1] + sb[2] + sb[3];
}

// DENSE fields of a uniform plan: workgroup (field i, column group jq) sums 4 columns of the field's
// gradient over the batch with 16-byte loads; jq == D/4 handles the first-order Linear(1,1).
// Batch slices (DensePartials): slice p = blk / (fields * groups) covers rows [p * rows, (p + 1) * rows) and
// STORES its sums at out[p * elems + (address of the gradient element - base)] instead of adding them in place.
struct DensePartials {
  float* out;            // nullptr: one slice over the whole batch, added into the gradient buffers
  const float* base;
  int64_t elems, rows;
  int per_slice;         // workgroups per slice = fields * groups
};
__device__ __forceinline__ void dense_fields_uniform_body(int blk, const int32_t* __restrict__ dense_list, PtrTable in, GradTable gt, int64_t B, int F, int D, const float* __restrict__ g_first, const float* __restrict__ g_field, DensePartials dp = DensePartials{nullptr, nullptr, 0, 0, 0}) {
  const int groups = D / 4 + 1;      // column groups per field (+ the first-order Linear)
  int64_t b_lo = 0, b_hi = B;
  int slice = 0;
  if (dp.out) {
    slice = blk / dp.per_slice;
    blk -= slice * dp.per_slice;
    b_lo = slice * dp.rows;
    b_hi = b_lo + dp.rows < B ? b_lo + dp.rows : B;
  }
  const int f = dense_list[blk / groups];
  const int jq = blk % groups;
  const float* x = static_cast<const float*>(in.p[f]);
  float sw[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  if (jq * 4 < D) {
    const float* g = g_field + static_cast<int64_t>(f) * D + jq * 4;
#pragma unroll 8
    for (int64_t b = b_lo + threadIdx.x; b < b_hi; b += 256) {
      const float4 v = ld4(g + b * F * D);
      const float xb = x[b];
      sw[0] = fmaf(xb, v.x, sw[0]); sw[1] = fmaf(xb, v.y, sw[1]);
      sw[2] = fmaf(xb, v.z, sw[2]); sw[3] = fmaf(xb, v.w, sw[3]);
      sb[0] += v.x; sb[1] += v.y; sb[2] += v.z; sb[3] += v.w;
    }
  } else {
    for (int64_t b = b_lo + threadIdx.x; b < b_hi; b += 256) {
      const float v = g_first[b];
      sw[0] = fmaf(x[b], v, sw[0]);
      sb[0] += v;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) block_sum2(sw[u], sb[u]);
  if (threadIdx.x == 0) {
    const dfm_field_grad g = gt.g[f];
    if (dp.out) {
      float* o = dp.out + slice * dp.elems;
      if (jq * 4 < D) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[(g.w2 + jq * 4 + u) - dp.base] = sw[u]; o[(g.b2 + jq * 4 + u) - dp.base] = sb[u]; }
      } else {
        o[g.w1 - dp.base] = sw[0]; o[g.b1 - dp.base] = sb[0];
      }
    } else if (jq * 4 < D) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { g.w2[jq * 4 + u] += sw[u]; g.b2[jq * 4 + u] += sb[u]; }
    } else {
      g.w1[0] += sw[0]; g.b1[0] += sb[0];
    }
  }
}

// Where sample b's gradients live: g_field + (b * F + f) * D and g_first + b in one array each, or —
// field-sharded tables (shard.hip): one received segment per source rank — every `samples` samples a
// new segment starts `stride` floats further on, in both arrays.
struct SampleSegments {
  int64_t samples;   // 0: one array
  int64_t stride;
};

// One row gradient per distinct id.  Runs of up to kLongRun contributions are summed in sample order by the
// row's own D/4 lanes (bit-exact against a sequential reduction).  Longer runs — a hot id under a skewed
// distribution: Zipf(1.05) clipped to the vocabulary puts ~2 000 of a batch's 4 096 ids of a field on ONE
// row, and one lane group walking them took 0.9 ms — are summed by the whole workgroup: 256 / (D/4) lane
// groups take every (256 / (D/4))-th contribution each, the partial sums are added in group order (fixed
// tree: still bitwise reproducible, 64 x shorter chain at D = 16).
//
// Runs of more than kSplitRun contributions — half a batch on one id: a default / missing-value category — are
// summed by SEVERAL workgroups of the list (one workgroup alone takes 3-5 us per 512 contributions, 42 us for a
// run of 4096, while the list's other 63 workgroups have nothing to do).  The row plan lists those runs (at
// most one of 4096 ids) in the unused tail of the list's seg_start:
//     seg[CH] = number of split runs (valid when num_uniq <= CH - kSplitRun; otherwise there are none)
//     seg[CH - 1 - e] = entry u of split run e,   seg[CH - 1 - kMaxSplitRuns] = arrival counter (0)
// Workgroup bl < nsl of the list sums slice bl of every split run (contributions p0 + bl*per ... , same lane
// group tree as above) into row CH - 1 - (e * nsl + bl) of the list's row_g2 / row_g1 — rows behind
// num_uniq, free because a split run of > 2048 contributions leaves > 2047 rows unused; the last workgroup to
// arrive adds the nsl partial rows of every split run in slice order (fixed association again) and writes
// the row.  This hand-over between workgroups costs ~10 us (three dependent trips to memory: partial rows
// out, arrival counter, partial rows in), paid only by lists that have such a run.
constexpr int kLongRun = 64;
constexpr int kSplitRun = 2048;        // below this the hand-over costs more than it saves
constexpr int kMaxSplitRuns = 2;        // > 4096 / 2049
constexpr int kMaxSlices = 64;
constexpr int kRowgradLds = 2048;       // floats: groups x (D + 4) = 1024 + 1024 / (D/4) <= 2048

__device__ __forceinline__ void rowgrad_load(const float* __restrict__ g_field, const float* __restrict__ g_first,
                                             int64_t b, int F, int f, int D, int q, const SampleSegments& segs,
                                             float4& g, float& g1) {
  int64_t off = 0;
  if (segs.samples > 0) {
    const int64_t sg = b / segs.samples;
    off = sg * segs.stride;
    b -= sg * segs.samples;
  }
  g = ld4(g_field + off + (b * F + f) * D + q * 4);
  g1 = q == 0 ? g_first[off + b] : 0.f;
}
// contributions pos[p], p = p0, p0 + step, ... < p1, added in that order; eight positions, then their eight
// rows in flight at a time (the loads do not depend on each other — only the additions are ordered)
constexpr int kRunBatch = 8;
// agent-scope accesses for data handed from one workgroup to another inside a launch (tail of rowgrad_body)
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rowgrad_run(const float* __restrict__ g_field, const float* __restrict__ g_first,
                                            const int32_t* __restrict__ pos, int p0, int p1, int step, int F, int f,
                                            int D, int q, const SampleSegments& segs, float4& acc, float& acc1) {
  // every batch is predicated, the last one included: a tail walked one contribution at a time pays two
  // dependent memory latencies per contribution (7 of a 31-long strided run cost more than its 3 batches).
  // The positions of the NEXT batch are requested before the rows of this one are waited for: a long run
  // then pays one memory latency per batch instead of two (positions and rows both come from another XCD's
  // kernel, i.e. from memory: ~2.5 us each).
  int32_t b[kRunBatch];
#pragma unroll
  for (int u = 0; u < kRunBatch; ++u) b[u] = p0 + u * step < p1 ? pos[p0 + u * step] : 0;
  for (int p = p0; p < p1; p += kRunBatch * step) {
    float4 g[kRunBatch];
    float g1[kRunBatch];
#pragma unroll
    for (int u = 0; u < kRunBatch; ++u) {
      g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      g1[u] = 0.f;
      if (p + u * step < p1) rowgrad_load(g_field, g_first, b[u], F, f, D, q, segs, g[u], g1[u]);
    }
    const int pn = p + kRunBatch * step;
#pragma unroll
    for (int u = 0; u < kRunBatch; ++u) b[u] = pn + u * step < p1 ? pos[pn + u * step] : 0;
#pragma unroll
    for (int u = 0; u < kRunBatch; ++u) {
      if (p + u * step < p1) {                   // (no "+ 0.f": -0.f + 0.f would turn into +0.f)
        acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
        acc1 += g1[u];
      }
    }
  }
}

__device__ __forceinline__ void rowgrad_body(int blk, FieldMap fmap, int S, int F, int D, int lists, const float* __restrict__ g_first,
    const float* __restrict__ g_field, const int32_t* __restrict__ sorted_pos,
    int32_t* seg_start, const int32_t* __restrict__ num_uniq,
    float* row_g2, float* row_g1, SampleSegments segs = SampleSegments{0, 0}) {
  __shared__ float red[kRowgradLds];
  __shared__ int s_last;
  __shared__ int s_long, s_ent[kTailThreads], s_p0[kTailThreads], s_p1[kTailThreads];   // the workgroup's long runs
  const int lpr = D / 4, groups = kTailThreads / lpr;       // lane groups (= list entries) per workgroup
  // (lane groups that do not tile the workgroup — D/4 not a power of two — keep the linear mapping and the
  // sequential sum)
  const bool coop = kTailThreads % lpr == 0;
  const int q = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  // entry of this lane group.  The workgroups of a list INTERLEAVE its entries (workgroup bl of the list's nbl
  // takes u = bl, bl + nbl, ...): ids are sorted, so under a skewed distribution the hot rows are neighbours
  // (ids 1, 2, 3, ... of a Zipf law; every row of a 50-id field) and a workgroup that owned 64 consecutive
  // entries walked all of their long runs one after the other (238 us for 64 ids x 64 contributions).
  const int nbl = coop ? CH / groups : 1;
  int64_t list;
  int u;
  if (coop) {
    list = blk / nbl;
    u = grp * nbl + blk % nbl;
  } else {
    const int64_t entry = (static_cast<int64_t>(blk) * kTailThreads + threadIdx.x) / lpr;
    list = entry / CH;
    u = static_cast<int>(entry % CH);
  }
  const int nu = list < lists ? num_uniq[list] : 0;
  const bool valid = threadIdx.x < groups * lpr && u < nu;
  int32_t* seg = seg_start + (list < lists ? list : 0) * (CH + 1);       // (its tail holds the arrival counter)
  const int n_split = coop && nu > 0 && nu <= CH - kSplitRun ? seg[CH] : 0;     // uniform over the workgroup
  int p0 = 0, p1 = 0;
  if (valid) {
    p0 = seg[u]; p1 = seg[u + 1];
  }
  if (threadIdx.x == 0) s_long = 0;
  __syncthreads();
  const bool is_split = coop && valid && p1 - p0 > kSplitRun;
  const bool is_long = coop && valid && !is_split && p1 - p0 > kLongRun;
  if (is_long && q == 0) {                 // (the order of this list does not matter: the runs are independent)
    const int i = atomicAdd(&s_long, 1);
    s_ent[i] = u; s_p0[i] = p0; s_p1[i] = p1;
  }
  if (valid && !is_long && !is_split) {
    const int f = fmap.f[static_cast<int>(list % S)];
    const int32_t* pos = sorted_pos + list * CH;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc1 = 0.f;
    rowgrad_run(g_field, g_first, pos, p0, p1, 1, F, f, D, q, segs, acc, acc1);
    st4(row_g2 + (list * CH + u) * D + q * 4, acc);
    if (q == 0) row_g1[list * CH + u] = acc1;
  }
  __syncthreads();
  if (!s_long && !n_split) return;                          // uniform: the usual case ends here
  // ---- long runs of this workgroup's entries, one after the other, by all lane groups ----
  // (found in LDS: asking global memory for every entry's run length again was 64 dependent loads, ~40 us)
  const int n_long = s_long;
  for (int e = 0; e < n_long; ++e) {
    const int64_t ls = list;                                // coop: one list per workgroup
    const int ue = s_ent[e], q0 = s_p0[e], q1 = s_p1[e];
    const int f = fmap.f[static_cast<int>(ls % S)];
    const int32_t* pos = sorted_pos + ls * CH;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc1 = 0.f;
    if (grp < groups) rowgrad_run(g_field, g_first, pos, q0 + grp, q1, groups, F, f, D, q, segs, acc, acc1);
    if (grp < groups) {
      st4(red + grp * (D + 4) + q * 4, acc);
      if (q == 0) red[grp * (D + 4) + D] = acc1;
    }
    __syncthreads();
    if (grp == 0) {                                         // group 0 adds the partial sums in group order
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      float tot1 = 0.f;
      for (int gi = 0; gi < groups; ++gi) {
        const float4 v = ld4(red + gi * (D + 4) + q * 4);
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
        if (q == 0) tot1 += red[gi * (D + 4) + D];
      }
      st4(row_g2 + (ls * CH + ue) * D + q * 4, tot);
      if (q == 0) row_g1[ls * CH + ue] = tot1;
    }
    __syncthreads();
  }
  if (!n_split) return;
  // ---- split runs of the list: this workgroup's slice of each ----
  const int nsl = nbl < kMaxSlices ? nbl : kMaxSlices, bl = blk % nbl;
  if (bl >= nsl) return;
  const int f = fmap.f[static_cast<int>(list % S)];
  const int32_t* pos = sorted_pos + list * CH;
  for (int e = 0; e < n_split; ++e) {
    const int ue = seg[CH - 1 - e];
    const int q0 = seg[ue], q1 = seg[ue + 1];
    const int per = (q1 - q0 + nsl - 1) / nsl;
    const int s0 = q0 + bl * per, s1 = s0 + per < q1 ? s0 + per : q1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc1 = 0.f;
    if (grp < groups) {
      rowgrad_run(g_field, g_first, pos, s0 + grp, s1, groups, F, f, D, q, segs, acc, acc1);
      st4(red + grp * (D + 4) + q * 4, acc);
      if (q == 0) red[grp * (D + 4) + D] = acc1;
    }
    __syncthreads();
    if (grp == 0) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      float tot1 = 0.f;
      for (int gi = 0; gi < groups; ++gi) {
        const float4 v = ld4(red + gi * (D + 4) + q * 4);
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
        if (q == 0) tot1 += red[gi * (D + 4) + D];
      }
      const int64_t prow = list * CH + CH - 1 - (e * nsl + bl);
      st_agent(row_g2 + prow * D + q * 4 + 0, tot.x);
      st_agent(row_g2 + prow * D + q * 4 + 1, tot.y);
      st_agent(row_g2 + prow * D + q * 4 + 2, tot.z);
      st_agent(row_g2 + prow * D + q * 4 + 3, tot.w);
      if (q == 0) st_agent(row_g1 + prow, tot1);
    }
    __syncthreads();
  }
  // hand-over: the list's workgroups sit on different XCDs = different, non-coherent L2s.  The partial rows
  // are written and read with agent-scope accesses (write-through / L2-bypassing: a handful of floats), the
  // barrier waits for the stores, then the arrival is counted.  (__threadfence() instead makes every
  // workgroup write back and invalidate its XCD's whole L2: 133 us for this kernel.)
  int* counter = seg + (CH - 1 - kMaxSplitRuns);
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsl - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  // the nsl partial rows of every split run: one row per lane group (all loads of a run in flight together —
  // they come from memory, ~2 us each), then the same LDS tree as above
  const int nred = groups < nsl ? groups : nsl;
  for (int e = 0; e < n_split; ++e) {
    const int ue = seg[CH - 1 - e];
    if (grp < nred) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float acc1 = 0.f;
      for (int r = grp; r < nsl; r += groups) {
        const int64_t prow = list * CH + CH - 1 - (e * nsl + r);
        const float* pr = row_g2 + prow * D + q * 4;
        acc.x += ld_agent(pr); acc.y += ld_agent(pr + 1); acc.z += ld_agent(pr + 2); acc.w += ld_agent(pr + 3);
        if (q == 0) acc1 += ld_agent(row_g1 + prow);
      }
      st4(red + grp * (D + 4) + q * 4, acc);
      if (q == 0) red[grp * (D + 4) + D] = acc1;
    }
    __syncthreads();
    if (grp == 0) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      float tot1 = 0.f;
      for (int gi = 0; gi < nred; ++gi) {
        const float4 v = ld4(red + gi * (D + 4) + q * 4);
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
        if (q == 0) tot1 += red[gi * (D + 4) + D];
      }
      st4(row_g2 + (list * CH + ue) * D + q * 4, tot);
      if (q == 0) row_g1[list * CH + ue] = tot1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the row plan zeroes it too: this is for a plan used twice)
}

__device__ __forceinline__ int find_row(const int32_t* __restrict__ rows, int n, int32_t row) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = rows[mid];
    if (v < row) lo = mid + 1; else hi = mid;
  }
  return (lo < n && rows[lo] == row) ? lo : -1;
}

// Membership of every entry of list lq in list ln (same field), for all ordered pairs lq != ln:
// workgroup (s, ln, lq) holds list ln's sorted rows in LDS (16 KB) and answers list lq's <= 4096
// queries with LDS binary searches; match[(lq*S+s)*CH + u][ln] = 1 iff found.  Replaces the
// O(L^2) global-memory binary searches of the merge when many lists (data-parallel ranks) meet.
// THREADS per workgroup: the queries are independent chains of ~12 dependent LDS reads, so the kernel wants
// as many of them in flight as a workgroup can hold (1024: 4 queries per thread instead of 16).
template <int THREADS>
__device__ __forceinline__ void rowadam_match_body(int blk, int S, int L, const int32_t* __restrict__ uniq_rows,
                                                   const int32_t* __restrict__ num_uniq,
                                                   uint8_t* __restrict__ match) {
  __shared__ int32_t rows[CH];
  const int s = blk % S;
  const int pair = blk / S;
  const int ln = pair / L, lq = pair % L;
  if (ln == lq) return;
  const int64_t lt = static_cast<int64_t>(ln) * S + s, lqs = static_cast<int64_t>(lq) * S + s;
  const int nt = num_uniq[lt], nq = num_uniq[lqs];
  for (int i = threadIdx.x; i < nt; i += THREADS) rows[i] = uniq_rows[lt * CH + i];
  __syncthreads();
  for (int u = threadIdx.x; u < nq; u += THREADS) {
    const int32_t row = uniq_rows[lqs * CH + u];
    int lo = 0, hi = nt;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rows[mid] < row) lo = mid + 1; else hi = mid;
    }
    match[(lqs * CH + u) * L + ln] = (lo < nt && rows[lo] == row) ? 1 : 0;
  }
}

// pass A of the row-wise Adam: ownership merge of the L lists, lazy L2, |g|^2 partial per workgroup
__device__ __forceinline__ void rowadam_merge_body(int blk, TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, float* __restrict__ row_g2, float* __restrict__ row_g1,
    int32_t* __restrict__ owner_flag, float grad_scale, float l2, float* __restrict__ partial,
    const uint8_t* __restrict__ match = nullptr) {
  const int lpr = D / 4;
  const int64_t t = static_cast<int64_t>(blk) * kTailThreads + threadIdx.x;
  const int q = static_cast<int>(t % lpr);
  const int64_t entry = t / lpr;
  const int64_t list = entry / CH;  // l * S + s
  const int u = static_cast<int>(entry % CH);
  float sq = 0.f;
  if (list < static_cast<int64_t>(L) * S && u < num_uniq[list]) {
    const int l = static_cast<int>(list / S), s = static_cast<int>(list % S);
    const int32_t row = uniq_rows[list * CH + u];
    // match (optional, from rowadam_match_body): byte [entry][ln] != 0 iff list ln of this field also
    // holds the row — the L-1 binary searches per entry become L byte reads, and a search is only
    // repeated for the (rare) lists that do hold it
    const uint8_t* mt = match ? match + (list * CH + u) * L : nullptr;
    bool owner = true;
    for (int lp = 0; lp < l && owner; ++lp) {
      const int64_t other = static_cast<int64_t>(lp) * S + s;
      if (mt ? mt[lp] != 0 : find_row(uniq_rows + other * CH, num_uniq[other], row) >= 0) owner = false;
    }
    if (q == 0) owner_flag[list * CH + u] = owner ? 1 : 0;
    if (owner) {
      float4 g = ld4(row_g2 + (list * CH + u) * D + q * 4);
      float g1 = q == 0 ? row_g1[list * CH + u] : 0.f;
      for (int ln = l + 1; ln < L; ++ln) {
        if (mt && mt[ln] == 0) continue;
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
  block_partial(sq, partial, blk);
}

__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, float b1, float b2,
                                      float step_size, float inv_bc2_sqrt, float eps) {
  m = fmaf(b1, m, (1.f - b1) * g);
  v = fmaf(b2, v, (1.f - b2) * g * g);
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  w -= step_size * (m / denom);
}

// pass B: Adam on the rows this list owns
__device__ __forceinline__ void rowadam_apply_body(int blk, TableArgs tabs, int S, int D, int L, const int32_t* __restrict__ uniq_rows,
    const int32_t* __restrict__ num_uniq, const float* __restrict__ row_g2,
    const float* __restrict__ row_g1, const int32_t* __restrict__ owner_flag,
    const float* __restrict__ clip_coef, float lr, float b1, float b2, float eps,
    const int32_t* __restrict__ step_ptr) {
  const int lpr = D / 4;
  const int64_t t = static_cast<int64_t>(blk) * kTailThreads + threadIdx.x;
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

// dense parameters: g[i] += 2*l2*p[i] for i < n_l2; |g|^2 partial per workgroup (16 elements per thread)
constexpr int kPrepPerThread = 16;
__device__ __forceinline__ void dense_prepare_body(int blk, float* __restrict__ g, const float* __restrict__ p,
                                                   int64_t n, int64_t n_l2, float l2, float* __restrict__ partial) {
  const int64_t base = (static_cast<int64_t>(blk) * kTailThreads + threadIdx.x) * kPrepPerThread;
  float sq = 0.f;
  const float k = 2.f * l2;
#pragma unroll 4
  for (int j = 0; j < kPrepPerThread; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      float gi = g[i];
      if (i < n_l2) { gi = fmaf(k, p[i], gi); g[i] = gi; }
      sq = fmaf(gi, gi, sq);
    }
  }
  block_partial(sq, partial, blk);
}

// torch.optim.Adam on one element per thread; g_zero != NULL also clears the gradient
__device__ __forceinline__ void dense_adam_body(int blk, float* __restrict__ p, float* __restrict__ m,
                                                float* __restrict__ v, const float* __restrict__ g, int64_t n,
                                                const float* __restrict__ clip_coef, float lr, float b1, float b2,
                                                float eps, const int32_t* __restrict__ step_ptr,
                                                float* __restrict__ g_zero) {
  const int64_t i = static_cast<int64_t>(blk) * kTailThreads + threadIdx.x;
  if (i >= n) return;
  const float clip = clip_coef ? clip_coef[0] : 1.f;
  const float step = static_cast<float>(step_ptr[0]);
  const float step_size = lr / (1.f - powf(b1, step));
  const float inv_bc2_sqrt = 1.f / sqrtf(1.f - powf(b2, step));
  const float gi = g[i] * clip;
  const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
  const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
  m[i] = mi;
  v[i] = vi;
  p[i] -= step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
  if (g_zero) g_zero[i] = 0.f;       // the gradient buffer is ready for the next step's accumulation
}

}  // namespace tail
}  // namespace dfm

extern "C" int dfm_linear_backward_splits(int64_t batch, int out_features, int in_features);

namespace dfm {
namespace tail {
// host: dfm_slab_ref[] -> SlabTable (every g_w must be a 64-byte aligned view of d_g[0 .. n))
inline int fill_slab_table(const dfm_slab_ref* slabs, int num_slabs, const float* d_g, int64_t n, SlabTable* st) {
  DFM_REQUIRE(num_slabs >= 0 && num_slabs <= kMaxSlabs && (num_slabs == 0 || slabs), "0..%d slab references", kMaxSlabs);
  memset(st, 0, sizeof(*st));
  for (int i = 0; i < num_slabs; ++i) {
    const dfm_slab_ref& h = slabs[i];
    DFM_REQUIRE(h.workspace && h.g_w && h.batch > 0 && h.out_features > 0 && h.in_features > 0, "incomplete dfm_slab_ref");
    const int64_t elems = static_cast<int64_t>(h.out_features) * h.in_features;
    DFM_REQUIRE(elems % 4 == 0 && (reinterpret_cast<uintptr_t>(h.workspace) & 15) == 0 && h.g_w >= d_g &&
                    h.g_w + elems <= d_g + n && ((h.g_w - d_g) % 16) == 0,
                "slab-backed weights must be 64-byte aligned views of the dense gradient buffer");
    st->slabs[i] = static_cast<const float*>(h.workspace);
    st->g[i] = h.g_w;
    st->elems[i] = elems;
    st->splits[i] = h.splits > 0 ? h.splits : dfm_linear_backward_splits(h.batch, h.out_features, h.in_features);
  }
  st->count = num_slabs;
  return DFM_OK;
}
}  // namespace tail
}  // namespace dfm
