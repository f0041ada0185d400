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
#include "tail_bodies.h"

#include <cstdlib>

using namespace dfm;

namespace {
using tail::CH;
constexpr int SORT_THREADS = 1024;
constexpr int PER_THREAD = CH / SORT_THREADS;  // 4

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
constexpr int kTouchParts = 2;            // touch workgroups per (field, chunk): 2 048 ids each, 2 per thread
}  // namespace

// KeyT = uint64 (id << 32 | pos) for any vocabulary, or uint32 (id << 12 | pos) when every id of the
// launch fits 20 bits (vocabulary < 2^20 - 1: the 1M-row Criteo tables): the sorts are LDS-bound, so half the
// key bytes is a good part of the time.  Both sorts run with either key width (dynamic LDS: 64 / 96 KB).
template <typename KeyT, int SHIFT>
__global__ __launch_bounds__(SORT_THREADS) void rowplan_sort(
    IdTable ids, int S, int64_t n, int32_t* __restrict__ sorted_pos, int32_t* __restrict__ uniq_rows,
    int32_t* __restrict__ seg_start, int32_t* __restrict__ num_uniq, int32_t* error_flag, int ablate,
    TouchTable touch, int chunks) {
  if (static_cast<int>(blockIdx.y) >= chunks) {
    // row-touch workgroup: part p of chunk c of field s
    const int s = blockIdx.x, q = static_cast<int>(blockIdx.y) - chunks;
    const int c = q / kTouchParts, part = q % kTouchParts;
    const int64_t base = static_cast<int64_t>(c) * CH;
    const int len = static_cast<int>(n - base < CH ? n - base : CH);
    const int64_t* src = ids.p[s];
    const int vocab = ids.vocab[s];
    const float* w2 = touch.w2[s];
    const int64_t stride = touch.stride2[s];
    float keep = 0.f;
    for (int i = part * (CH / kTouchParts) + static_cast<int>(threadIdx.x); i < (part + 1) * (CH / kTouchParts);
         i += SORT_THREADS) {
      if (i < len) {
        const int64_t id = src[base + i];
        if (id > 0 && id < vocab) keep += w2[id * stride];
      }
    }
    asm volatile("" :: "v"(keep));          // the loads must be issued; their values are not used
    return;
  }
  constexpr KeyT SENTINEL = static_cast<KeyT>(~static_cast<KeyT>(0));
  constexpr KeyT POS_MASK = (static_cast<KeyT>(1) << SHIFT) - 1;
  // dynamic LDS (rowplan_lds_bytes): [keys | tmp | cnt | start] — 64 KB with 32-bit keys, 96 KB with 64-bit ones
  extern __shared__ __attribute__((aligned(16))) unsigned char rp_lds[];
  KeyT* keys = reinterpret_cast<KeyT*>(rp_lds);
  __shared__ int wave_tot[SORT_THREADS / kWave];
  const int s = blockIdx.x, c = blockIdx.y;
  const int tid = threadIdx.x;
  const int64_t base = static_cast<int64_t>(c) * CH;
  const int len = static_cast<int>(n - base < CH ? n - base : CH);
  const int64_t* src = ids.p[s];
  const int vocab = ids.vocab[s];

  // load (coalesced) and form keys
  for (int i = tid; i < CH; i += SORT_THREADS) {
    KeyT k = SENTINEL;
    if (i < len) {
      const int64_t id = src[base + i];
      if (id < 0 || id >= vocab) {
        if (error_flag) atomicOr(error_flag, 1);
      } else if (id != 0) {
        k = (static_cast<KeyT>(id) << SHIFT) | static_cast<KeyT>(i);
      }
    }
    keys[i] = k;
  }
  __syncthreads();

  // ---- fast path: counting sort over 4096 id buckets -------------------------------------------
  // With ids spread over the vocabulary a bucket (id * 4096 / vocab: monotone in id, so bucket
  // order is key order) holds about one key: histogram, exclusive scan, scatter into the bucket's
  // range, then every key ranks itself among the handful of keys of its bucket.  LDS atomics decide
  // only the transient slot inside a bucket; the final position depends on key comparisons alone, so
  // the result is THE sorted array.  Skewed ids (any bucket with more than kMaxBucket keys) take the
  // radix sort instead.
  bool sorted = false;
  // ---- stable LSD radix sort on the id bits, 7 bits a pass (the input is in position order, so sorting by id
  // stably IS the (id, position) order).  A key's slot inside its digit is a COUNT — keys of the same digit in
  // earlier rounds + lower waves of the round + lower lanes of the wave, from ballots — so the cost does not
  // depend on the distribution: hot ids (Zipf), a field with three ids, every bucket of the counting sort
  // below overflowing ... all take bits / 7 passes of ~3 us (20-bit ids: 22 us a launch against 14 for the
  // counting sort).  Used for vocabularies of <= 128 ids (one pass)
  // and whenever the counting sort's buckets overflow (that case used to run a bitonic network of 78
  // compare-exchange sweeps: 27 ... 30 us).
  constexpr int kDigits = 128, kW = SORT_THREADS / kWave;
  __shared__ int dstart[kDigits + 1];
  __shared__ int segsum[SORT_THREADS];
  KeyT* const buf1 = keys + CH;
  int* const wcnt = reinterpret_cast<int*>(keys + 2 * CH);     // [PER_THREAD rounds][16 waves][128 digits] = 32 KB
  // one pass: src -> dst by digit (key >> (SHIFT + shift)) & 127; nvalid < 0: the valid keys are the
  // non-sentinels of all CH slots (first pass), else slots [0, nvalid).  Returns the number of valid keys.
  auto radix_pass = [&](const KeyT* src, KeyT* dst, int shift, int nvalid) -> int {
    const int lane = lane_id(), w = tid >> 6;
    for (int i = tid; i < PER_THREAD * kW * kDigits; i += SORT_THREADS) wcnt[i] = 0;
    __syncthreads();
    KeyT mykey[PER_THREAD];
    int mydig[PER_THREAD], myrank[PER_THREAD];
#pragma unroll
    for (int r = 0; r < PER_THREAD; ++r) {
      const int i = tid + r * SORT_THREADS;                   // round r: array slots r * 1024 + tid, in order
      mykey[r] = src[i];
      const bool ok = nvalid < 0 ? mykey[r] != SENTINEL : i < nvalid;
      mydig[r] = ok ? static_cast<int>((mykey[r] >> (SHIFT + shift)) & (kDigits - 1)) : -1;
      // lanes of the wave holding the same digit: one ballot per digit bit (cost independent of the ids)
      unsigned long long mask = __ballot(ok);
#pragma unroll
      for (int bit = 0; bit < 7; ++bit) {
        const unsigned long long bm = __ballot((mydig[r] >> bit) & 1);
        mask &= ((mydig[r] >> bit) & 1) ? bm : ~bm;
      }
      myrank[r] = __popcll(mask & ((1ull << lane) - 1ull));
      if (ok && myrank[r] == 0) wcnt[(r * kW + w) * kDigits + mydig[r]] = __popcll(mask);
    }
    __syncthreads();
    {                               // per digit: exclusive prefix of its counts over (round, wave) = slot order;
      constexpr int kSeg = SORT_THREADS / kDigits, kPer = PER_THREAD * kW / kSeg;   // 8 threads a digit, 8 counts each
      const int d = tid & (kDigits - 1), sg = tid / kDigits;
      int cn[kPer], sum = 0;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        cn[q] = wcnt[(sg * kPer + q) * kDigits + d];
        sum += cn[q];
      }
      segsum[sg * kDigits + d] = sum;
      __syncthreads();
      int run = 0, tot = 0;
#pragma unroll
      for (int q = 0; q < kSeg; ++q) {
        const int v = segsum[q * kDigits + d];
        run += q < sg ? v : 0;
        tot += v;
      }
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        wcnt[(sg * kPer + q) * kDigits + d] = run;
        run += cn[q];
      }
      if (sg == 0) dstart[d + 1] = tot;
    }
    __syncthreads();
    if (tid < kWave) {              // exclusive scan of the 128 digit totals by one wave (two digits a lane)
      const int a0 = dstart[2 * tid + 1], a1 = dstart[2 * tid + 2];
      int incl = a0 + a1;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(incl, o, kWave);
        if (tid >= o) incl += t;
      }
      const int excl = incl - (a0 + a1);
      dstart[2 * tid + 1] = excl + a0;
      dstart[2 * tid + 2] = excl + a0 + a1;
      if (tid == 0) dstart[0] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER_THREAD; ++r)
      if (mydig[r] >= 0) dst[dstart[mydig[r]] + wcnt[(r * kW + w) * kDigits + mydig[r]] + myrank[r]] = mykey[r];
    const int total = dstart[kDigits];
    __syncthreads();
    return total;
  };
  // the whole sort: `passes` passes over the id bits, ping-pong keys <-> buf1, sentinels behind the valid keys
  auto radix_sort = [&]() {
    int bits = 1;
    while ((1ll << bits) < vocab) ++bits;                     // ids < vocab
    int nvalid = -1;
    const KeyT* src = keys;
    KeyT* dst = buf1;
    for (int shift = 0; shift < bits; shift += 7) {
      nvalid = radix_pass(src, dst, shift, nvalid);
      const KeyT* t = src; src = dst; dst = const_cast<KeyT*>(t);
    }
    // src holds the sorted valid keys
    if (src != keys)
      for (int i = tid; i < CH; i += SORT_THREADS) keys[i] = i < nvalid ? src[i] : SENTINEL;
    else
      for (int i = tid; i < CH; i += SORT_THREADS) if (i >= nvalid) keys[i] = SENTINEL;
    __syncthreads();
  };
  if (vocab <= kDigits) {
    radix_sort();
    sorted = true;
  }
  if (!sorted) {
    constexpr int kMaxBucket = 64;
    KeyT* tmp = keys + CH;
    int* cnt = reinterpret_cast<int*>(tmp + CH);       // keys per bucket, then the scatter cursor
    int* start = cnt + CH;                             // first output slot of every bucket
    __shared__ int wave_sum[SORT_THREADS / kWave];
    __shared__ int s_max;
#pragma unroll
    for (int r = 0; r < PER_THREAD; ++r) cnt[tid * PER_THREAD + r] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    KeyT mykey[PER_THREAD];
    int mybucket[PER_THREAD];
#pragma unroll
    for (int r = 0; r < PER_THREAD; ++r) {
      mykey[r] = keys[tid + r * SORT_THREADS];
      mybucket[r] = -1;
      if (mykey[r] != SENTINEL) {
        const unsigned long long id = mykey[r] >> SHIFT;
        mybucket[r] = static_cast<int>((id << 12) / static_cast<unsigned long long>(vocab));   // < 4096 since id < vocab
        atomicAdd(&cnt[mybucket[r]], 1);
      }
    }
    __syncthreads();
    // exclusive scan of cnt over the block: thread t owns buckets [4t, 4t+4)
    int c4[PER_THREAD], local = 0, mx = 0;
#pragma unroll
    for (int r = 0; r < PER_THREAD; ++r) {
      c4[r] = cnt[tid * PER_THREAD + r];
      local += c4[r];
      mx = c4[r] > mx ? c4[r] : mx;
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane_id() >= o) incl += t;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int t = __shfl_xor(mx, m, kWave);
      mx = t > mx ? t : mx;
    }
    if (lane_id() == kWave - 1) wave_sum[tid >> 6] = incl;
    if (lane_id() == 0) atomicMax(&s_max, mx);
    __syncthreads();
    int woff = 0, total_valid = 0;
    for (int i = 0; i < SORT_THREADS / kWave; ++i) {
      if (i < (tid >> 6)) woff += wave_sum[i];
      total_valid += wave_sum[i];
    }
    if (s_max <= kMaxBucket && !(ablate & 1)) {
      int run = woff + incl - local;
#pragma unroll
      for (int r = 0; r < PER_THREAD; ++r) {
        start[tid * PER_THREAD + r] = run;
        cnt[tid * PER_THREAD + r] = 0;           // becomes the scatter cursor
        run += c4[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PER_THREAD; ++r)
        if (mybucket[r] >= 0) tmp[start[mybucket[r]] + atomicAdd(&cnt[mybucket[r]], 1)] = mykey[r];
      __syncthreads();
      // every key places itself: rank among the keys of its bucket
#pragma unroll
      for (int r = 0; r < PER_THREAD; ++r) {
        const int i = tid + r * SORT_THREADS;
        KeyT out = SENTINEL;
        int dst = i;                                // slots >= total_valid keep the sentinel
        if (i < total_valid) {
          out = tmp[i];
          const int b = static_cast<int>(((static_cast<unsigned long long>(out >> SHIFT)) << 12) /
                                         static_cast<unsigned long long>(vocab));
          const int b0 = start[b], nb = cnt[b];
          int rank = 0;
          for (int q = 0; q < nb; ++q) rank += tmp[b0 + q] < out ? 1 : 0;
          dst = b0 + rank;
        }
        keys[dst] = out;
      }
      __syncthreads();
      sorted = true;
    }
  }

  if (!sorted && !(ablate & 1)) radix_sort();                  // overflowing buckets: skew-proof path

  if (ablate & 2) return;
  // run heads over the valid prefix; thread owns PER_THREAD consecutive entries
  const int e0 = tid * PER_THREAD;
  KeyT mine[PER_THREAD];
  KeyT prev = e0 > 0 ? keys[e0 - 1] : SENTINEL;
  int head[PER_THREAD];
  int cnt = 0, valid = 0;
#pragma unroll
  for (int r = 0; r < PER_THREAD; ++r) {
    mine[r] = keys[e0 + r];
    const bool ok = mine[r] != SENTINEL;
    const bool h = ok && (e0 + r == 0 || (mine[r] >> SHIFT) != (prev >> SHIFT));
    head[r] = h ? 1 : 0;
    cnt += head[r];
    valid += ok ? 1 : 0;
    prev = mine[r];
  }
  // block-wide exclusive scan of cnt, and total of valid
  const int lane = lane_id(), w = tid >> 6;
  int incl = cnt, vsum = valid;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) vsum += __shfl_xor(vsum, m, kWave);
  __shared__ int wave_valid[SORT_THREADS / kWave];
  if (lane == kWave - 1) wave_tot[w] = incl;
  if (lane == 0) wave_valid[w] = vsum;
  __syncthreads();
  int wave_off = 0, total_u = 0, total_valid = 0;
  for (int i = 0; i < SORT_THREADS / kWave; ++i) {
    if (i < w) wave_off += wave_tot[i];
    total_u += wave_tot[i];
    total_valid += wave_valid[i];
  }
  int slot = wave_off + incl - cnt;  // exclusive prefix

  const size_t list = static_cast<size_t>(c) * S + s;
  int32_t* o_pos = sorted_pos + list * CH;
  int32_t* o_rows = uniq_rows + list * CH;
  int32_t* o_seg = seg_start + list * (CH + 1);
#pragma unroll
  for (int r = 0; r < PER_THREAD; ++r) {
    const int p = e0 + r;
    const bool ok = mine[r] != SENTINEL;
    o_pos[p] = ok ? static_cast<int32_t>(base + static_cast<int64_t>(mine[r] & POS_MASK)) : -1;
    if (head[r]) {
      o_rows[slot] = static_cast<int32_t>(mine[r] >> SHIFT);
      o_seg[slot] = p;
      ++slot;
    }
  }
  if (tid == 0) {
    o_seg[total_u] = total_valid;
    num_uniq[list] = total_u;
  }
  // runs of more than kSplitRun ids, for the row gradients to split over workgroups: listed in the free tail of
  // seg_start (layout and reasons: tail_bodies.h::rowgrad_body).  Run lengths from the run starts in LDS.
  if (total_u > CH - tail::kSplitRun) return;               // (then no run can be that long)
  int* hstart = reinterpret_cast<int*>(keys + 2 * CH);
  __shared__ int s_nsplit, s_split[tail::kMaxSplitRuns];
  if (tid == 0) { s_nsplit = 0; hstart[total_u] = total_valid; }
  slot = wave_off + incl - cnt;
#pragma unroll
  for (int r = 0; r < PER_THREAD; ++r)
    if (head[r]) hstart[slot++] = e0 + r;
  __syncthreads();
  for (int j = tid; j < total_u; j += SORT_THREADS)
    if (hstart[j + 1] - hstart[j] > tail::kSplitRun) s_split[atomicAdd(&s_nsplit, 1)] = j;
  __syncthreads();
  if (tid < s_nsplit) o_seg[CH - 1 - tid] = s_split[tid];
  if (tid == 0) {
    o_seg[CH] = s_nsplit;
    o_seg[CH - 1 - tail::kMaxSplitRuns] = 0;                // arrival counter of the list's workgroups
  }
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
  r->lds = static_cast<unsigned>((narrow ? 4 : 8) * 2 * CH + 2 * CH * sizeof(int));
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
