// Device bodies of the row plan (csrc/rowplan.hip), shared with the fused optimizer launch (csrc/step_tail.hip).
#pragma once
#include "tail_bodies.h"

namespace dfm {
namespace rowplan {
using tail::CH;
constexpr int SORT_THREADS = 1024;
constexpr int PER_THREAD = CH / SORT_THREADS;  // 4
constexpr int kTouchParts = 2;            // touch workgroups per (field, chunk): 2 048 ids each, 2 per thread
// dynamic LDS of a chunk workgroup: [keys | tmp | cnt | start]
__host__ __device__ constexpr unsigned lds_bytes(bool narrow) {
  return static_cast<unsigned>((narrow ? 4 : 8) * 2 * CH + 2 * CH * sizeof(int));
}

// KeyT = uint64 (id << 32 | pos) for any vocabulary, or uint32 (id << 12 | pos) when every id of the
// launch fits 20 bits (vocabulary < 2^20 - 1: the 1M-row Criteo tables): the sorts are LDS-bound, so half the
// key bytes is a good part of the time.  Both sorts run with either key width (dynamic LDS: 64 / 96 KB).
// row-touch workgroup: part `part` of chunk c of one field (see TouchTable in rowplan.hip)
__device__ __forceinline__ void rowplan_touch_body(const int64_t* __restrict__ src, int vocab, const float* __restrict__ w2,
                                                   int64_t stride, int c, int part, int64_t n) {
  const int64_t base = static_cast<int64_t>(c) * CH;
  const int len = static_cast<int>(n - base < CH ? n - base : CH);
  float keep = 0.f;
  for (int i = part * (CH / kTouchParts) + static_cast<int>(threadIdx.x); i < (part + 1) * (CH / kTouchParts);
       i += SORT_THREADS) {
    if (i < len) {
      const int64_t id = src[base + i];
      if (id > 0 && id < vocab) keep += w2[id * stride];
    }
  }
  asm volatile("" :: "v"(keep));          // the loads must be issued; their values are not used
}

// One chunk (<= 4096 ids) of one field, by ONE 1024-thread workgroup with rowplan_lds_bytes of dynamic LDS:
// the body of rowplan_sort, callable from any launch of that shape (csrc/step_tail.hip fuses it into the
// optimizer's apply launch for the NEXT step's ids).
template <typename KeyT, int SHIFT>
__device__ __forceinline__ void rowplan_chunk_body(
    const int64_t* __restrict__ src, int vocab, int s, int c, int S, int64_t n, int32_t* __restrict__ sorted_pos,
    int32_t* __restrict__ uniq_rows, int32_t* __restrict__ seg_start, int32_t* __restrict__ num_uniq,
    int32_t* error_flag, int ablate) {
  constexpr KeyT SENTINEL = static_cast<KeyT>(~static_cast<KeyT>(0));
  constexpr KeyT POS_MASK = (static_cast<KeyT>(1) << SHIFT) - 1;
  // dynamic LDS (rowplan_lds_bytes): [keys | tmp | cnt | start] — 64 KB with 32-bit keys, 96 KB with 64-bit ones
  extern __shared__ __attribute__((aligned(16))) unsigned char rp_lds[];
  KeyT* keys = reinterpret_cast<KeyT*>(rp_lds);
  __shared__ int wave_tot[SORT_THREADS / kWave];
  const int tid = threadIdx.x;
  const int64_t base = static_cast<int64_t>(c) * CH;
  const int len = static_cast<int>(n - base < CH ? n - base : CH);

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


}  // namespace rowplan
}  // namespace dfm
