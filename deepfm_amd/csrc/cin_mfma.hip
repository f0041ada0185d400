// CIN forward on the gfx950 matrix cores: the WHOLE layer stack in one launch.
// Reference: deepfm/models/layers/cin.py:66-105.
//
// GEMM view of one layer:  Y[c, n] = relu(bias[c] + sum_k W[c,k] Z[k,n]),
// k = (h, f), n = (b, d), Z[(h,f),(b,d)] = hidden[b,h,d] * x0[b,f,d].  Every column n depends only
// on column n of hidden and x0, so a wave that owns 32 columns (two samples at D = 16) can run all
// layers back to back: the "next" half of its accumulator tile IS the hidden input of the following
// layer (handed over through a wave-private LDS image), and Z — 398-654 MB per layer in the
// reference — is generated in registers, one 16-deep k-step at a time, as the MFMA B operand.
//
// Mapping onto v_mfma_f32_32x32x16_bf16 (D = A*B + C, A: 32 rows x 16 k, B: 16 k x 32 cols):
//   * rows   = output channels c, up to four 32-row blocks (C <= 128) -> 64 accumulator VGPRs;
//   * cols   = the wave's 32 (b,d) columns;
//   * k-step = one field group fg (8 consecutive f) for TWO hidden rows: lane half 0 takes hidden
//     row 2*hp, half 1 takes row 2*hp+1, so a lane's 8 B-operand values are
//     hidden[h][n] * x0[fg*8 + j][n]: one LDS scalar times 8 registers of the lane's x0 column.
//   * A operand: W repacked by cin_pack_weights into exactly the fragment order
//     [k-step][row block][lane][8 bf16] — one 16-B load per lane per MFMA, staged through LDS once
//     per workgroup (4 waves share a slab), double buffered.
// Numerics: bf16 x 3 split — W = Wh + Wl, Z = Zh + Zl, acc += Wh*Zh + Wh*Zl + Wl*Zh (fp32
// accumulate) — which holds the 1e-4 logit bar (plain bf16 is 4e-3, the split
// 7e-6); SPLIT = false is the plain-bf16 throughput mode (own tolerance, never used for parity).
#include <type_traits>

#include "common.h"

using namespace dfm;

namespace dfm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CIN_FWD_ABLATE
#define CIN_FWD_ABLATE 0      // timing-only ablations (tools/build_variant.sh); 0 in the product
#endif
constexpr int kCinMaxLayers = 8;
constexpr int kCinWaves = 8;          // waves per workgroup = per CU (share the weight slabs)
constexpr int kCinCols = 32;          // columns per wave

struct CinMfmaLayer {
  const __bf16* w_hi;   // packed (HP*FG, MB, 64, 8)
  const __bf16* w_lo;
  const float* bias;    // (C)
  float* Y;             // (B, C, D) post-ReLU activations for the backward, or null
  uint32_t* mask;       // (B*D, 4) ReLU masks, bit c of a column's 128 = (Y[c] > 0), or null; with it only the rows
  int y_from;           // >= y_from of Y are stored (the "next" half: the hidden input of the following layer)
  int C, H, HP, MB, direct, next_off, next_count, out_col;
};
struct CinMfmaArgs {
  const float* x0;
  float* out;
  int64_t B;
  int F, L, out_dim, hid_rows;  // hid_rows: rows of the per-wave hidden image in LDS
  int pad_;
  unsigned long long* stamps;   // tools/microbench_cin only (DFM_CIN_STAMPS build): per-wave phase sums
  CinMfmaLayer layer[kCinMaxLayers];
};

// ---- weight packing -----------------------------------------------------------------
// W (C, H*F) fp32 -> hi/lo bf16 fragments [ks = hp*FG+fg][mb][lane = hf*32+r][j]:
//   value = W[c = mb*32+r][h = 2*hp+hf][f = fg*8+j]  (0 outside C/H/F)
// Every layer of the stack in ONE launch (the weights change every training step: three launches of ~5 us
// each were 1 % of an xDeepFM step): workgroup -> layer by a prefix table of workgroup counts.
struct CinPackJob {
  const float* W;
  __bf16* hi;
  __bf16* lo;
  int C, H, p0, p1;          // forward: p0 = HP, p1 = MB; backward (W^T fragments): p0 = HQ, p1 = KS
};
struct CinPackJobs {
  CinPackJob job[kCinMaxLayers];
  int first_block[kCinMaxLayers + 1];
  int count, F, FG;
};
__device__ __forceinline__ int cin_pack_find(const CinPackJobs& jobs, int blk) {
  int i = 0;
  while (i + 1 < jobs.count && blk >= jobs.first_block[i + 1]) ++i;
  return i;
}

__device__ __forceinline__ void cin_pack_weights_body(int blk, const float* __restrict__ W, int C, int H, int F,
                                                      int HP, int FG, int MB,
                                                      __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const int64_t t = static_cast<int64_t>(blk) * 256 + threadIdx.x;
  const int64_t total = static_cast<int64_t>(HP) * FG * MB * 64 * 8;
  if (t >= total) return;
  const int j = t & 7;
  const int lane = (t >> 3) & 63;
  const int64_t rest = t >> 9;
  const int mb = static_cast<int>(rest % MB);
  const int ks = static_cast<int>(rest / MB);
  const int hp = ks / FG, fg = ks % FG;
  const int c = mb * 32 + (lane & 31), h = 2 * hp + (lane >> 5), f = fg * 8 + j;
  float v = 0.f;
  if (c < C && h < H && f < F) v = W[static_cast<int64_t>(c) * H * F + h * F + f];
  const __bf16 vh = static_cast<__bf16>(v);
  hi[t] = vh;
  lo[t] = static_cast<__bf16>(v - static_cast<float>(vh));
}
__global__ __launch_bounds__(256) void cin_pack_weights(const float* __restrict__ W, int C, int H, int F,
                                                        int HP, int FG, int MB,
                                                        __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  cin_pack_weights_body(blockIdx.x, W, C, H, F, HP, FG, MB, hi, lo);
}
__global__ __launch_bounds__(256) void cin_pack_weights_all(CinPackJobs jobs) {
  const int i = cin_pack_find(jobs, blockIdx.x);
  const CinPackJob& j = jobs.job[i];
  cin_pack_weights_body(blockIdx.x - jobs.first_block[i], j.W, j.C, j.H, jobs.F, j.p0, jobs.FG, j.p1, j.hi, j.lo);
}

// ---- the fused forward ----------------------------------------------------------------
template <int D, int FG, bool SPLIT>
__global__ __launch_bounds__(kCinWaves * 64, 1) void cin_fwd_mfma(CinMfmaArgs args) {
  static_assert(kCinCols % D == 0, "a 32-column tile must hold whole samples");
  constexpr int SPT = kCinCols / D;                       // samples per tile
  // one slab = the A fragments of ONE hidden-row pair: FG k-steps x (up to) 4 row blocks x 64 lanes x 16 B
  constexpr int SLAB = FG * 4 /*MB max*/ * 64 * 16;       // bytes of one hi (or lo) slab
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // [2 buffers][hi, lo][MB*64 lanes][16 B]  then per-wave hidden images
  unsigned char* wbuf = lds_raw;
  float* hid_all = reinterpret_cast<float*>(lds_raw + 2 * 2 * SLAB);
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int tid = threadIdx.x;
  float* hid = hid_all + static_cast<size_t>(wave) * args.hid_rows * kCinCols;
  const int n = lane & 31, hf = lane >> 5;
  const int64_t col0 = (static_cast<int64_t>(blockIdx.x) * kCinWaves + wave) * kCinCols;
  const int64_t ncols = args.B * D;
  const int64_t col = col0 + n;
  const bool live = col < ncols;
  const int64_t b = live ? col / D : 0;
  const int d = static_cast<int>(col % D);
  const int F = args.F;

  // the lane's x0 column in registers (zero beyond F / beyond the batch)
  float x0r[FG * 8];
#pragma unroll
  for (int f = 0; f < FG * 8; ++f)
    x0r[f] = (live && f < F) ? args.x0[(b * F + f) * D + d] : 0.f;
  // layer 0: hidden = x0 (wave-private LDS image, both lane halves write the same values)
  for (int r = 0; r < args.hid_rows; ++r) {
    if (hf == (r & 1)) hid[r * kCinCols + n] = 0.f;
  }
#pragma unroll
  for (int f = 0; f < FG * 8; ++f)
    if (hf == 0 && f < args.hid_rows) hid[f * kCinCols + n] = x0r[f];

#ifdef DFM_CIN_STAMPS
  unsigned long long st_work = 0, st_barrier = 0, st_pre = 0, st_epi = 0;
  const unsigned long long st_begin = wall_clock64();
  unsigned long long st_mark = st_begin;
#endif
  for (int li = 0; li < args.L; ++li) {
    const CinMfmaLayer ly = args.layer[li];
    const int MB = ly.MB;
    f32x16 acc[4];
    {
      // accumulators start at the bias; rows of this lane: mb*32 + (r&3) + 8*(r>>2) + 4*hf
      const float* brow = ly.bias + 4 * hf;
      const int c_lim = ly.C - 4 * hf;          // row < C  <=>  (row - 4*hf) < c_lim
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = mb * 32 + (r & 3) + 8 * (r >> 2);   // compile-time
          acc[mb][r] = rr < c_lim ? brow[rr] : 0.f;
        }
      }
    }
    // Weight pipeline, one slab (FG k-steps: every k-step of a hidden-row pair) at a time: slab t is loaded
    // global -> registers during iteration t-2, written to LDS at the start of iteration t-1 (the loads
    // had a whole iteration to land), made visible by the ONE barrier that ends iteration t-1, read
    // during iteration t.  Two LDS buffers, one register set of up to 3 x 16 B per thread and half.
    // (Round 1 staged and synchronised per k-step: 420 barriers per wave at the Criteo shape, 74 ns of
    // barrier wait + lock-step between the two waves of a SIMD each — tools/microbench_cin; now 84.)
    constexpr int kStageRounds = (FG * 4 * 64 + kCinWaves * 64 - 1) / (kCinWaves * 64);
    const int slab_u4 = FG * MB * 64;                       // uint4 per slab and half
    // (named registers, not arrays: arrays handed to a lambda went to scratch memory)
    static_assert(kStageRounds <= 3, "stage registers");
    uint4 vh0 = {}, vh1 = {}, vh2 = {}, vl0 = {}, vl1 = {}, vl2 = {};
    auto stage_load = [&](int hp) {
      const uint4* gh = reinterpret_cast<const uint4*>(ly.w_hi) + static_cast<int64_t>(hp) * slab_u4;
      const uint4* gl = reinterpret_cast<const uint4*>(ly.w_lo) + static_cast<int64_t>(hp) * slab_u4;
      constexpr int T = kCinWaves * 64;
      if (tid < slab_u4) { vh0 = gh[tid]; if (SPLIT) vl0 = gl[tid]; }
      if (kStageRounds > 1 && tid + T < slab_u4) { vh1 = gh[tid + T]; if (SPLIT) vl1 = gl[tid + T]; }
      if (kStageRounds > 2 && tid + 2 * T < slab_u4) { vh2 = gh[tid + 2 * T]; if (SPLIT) vl2 = gl[tid + 2 * T]; }
    };
    auto stage_store = [&](int buf) {
      uint4* bh_ = reinterpret_cast<uint4*>(wbuf + buf * 2 * SLAB);
      uint4* bl_ = reinterpret_cast<uint4*>(wbuf + buf * 2 * SLAB + SLAB);
      constexpr int T = kCinWaves * 64;
      if (tid < slab_u4) { bh_[tid] = vh0; if (SPLIT) bl_[tid] = vl0; }
      if (kStageRounds > 1 && tid + T < slab_u4) { bh_[tid + T] = vh1; if (SPLIT) bl_[tid + T] = vl1; }
      if (kStageRounds > 2 && tid + 2 * T < slab_u4) { bh_[tid + 2 * T] = vh2; if (SPLIT) bl_[tid + 2 * T] = vl2; }
    };
    __syncthreads();                       // previous layer's readers are done with both buffers / hid
    stage_load(0);
    stage_store(0);
    if (ly.HP > 1) stage_load(1);
    __syncthreads();
    // One k-step = one 16-deep slice of the reduction: 4 (MB) x 3 (split) MFMAs per wave.  All A
    // fragments of the step are requested from LDS first, the B operand is generated while they are
    // in flight, and the MFMAs run hh / hl / lh across the four independent accumulators.  FULL
    // (MB == 4, the usual 128-channel layer) is free of branches, so nothing waits early: with a
    // uniform `mb < MB` test around every accumulator the compiler issued read - wait - MFMA eight
    // times per step (LDS latency exposed each time).  No barrier between the FG k-steps of a slab:
    // the scheduler is free to overlap one step's B generation with another's MFMAs.
    auto kloop = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      for (int hp = 0; hp < ly.HP; ++hp) {
        const float hv = hid[(2 * hp + hf) * kCinCols + n];
        const int cur = hp & 1;
        const unsigned char* base = wbuf + cur * 2 * SLAB;
#ifdef DFM_CIN_STAMPS
        const unsigned long long ts0 = wall_clock64();
#endif
        if (hp + 1 < ly.HP) stage_store(cur ^ 1);   // slab hp+1 (loaded one iteration ago)
        if (hp + 2 < ly.HP) stage_load(hp + 2);
#pragma unroll
        for (int fg = 0; fg < FG; ++fg) {
          bf16x8 ah[4], al[4];
          if constexpr (FULL) {
            const int fga = (CIN_FWD_ABLATE & 2) ? 0 : fg;        // timing-only ablation: one fragment set per slab
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              ah[mb] = reinterpret_cast<const bf16x8*>(base)[(fga * 4 + mb) * 64 + lane];
              if (SPLIT) al[mb] = reinterpret_cast<const bf16x8*>(base + SLAB)[(fga * 4 + mb) * 64 + lane];
            }
          }
          // B operand: Z values of this k-step for the lane's column
          bf16x8 bh, bl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float z = hv * x0r[((CIN_FWD_ABLATE & 1) ? 0 : fg) * 8 + j];   // timing-only ablation: one B per slab
            bh[j] = static_cast<__bf16>(z);
            if (SPLIT) bl[j] = static_cast<__bf16>(z - static_cast<float>(bh[j]));
          }
          if constexpr (FULL) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb], 0, 0, 0);
            if (SPLIT) {
#pragma unroll
              for (int mb = 0; mb < 4; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl, acc[mb], 0, 0, 0);
#pragma unroll
              for (int mb = 0; mb < 4; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh, acc[mb], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              if (mb < MB) {
                const bf16x8 a_h = reinterpret_cast<const bf16x8*>(base)[(fg * MB + mb) * 64 + lane];
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh, acc[mb], 0, 0, 0);
                if (SPLIT) {
                  const bf16x8 a_l = reinterpret_cast<const bf16x8*>(base + SLAB)[(fg * MB + mb) * 64 + lane];
                  acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl, acc[mb], 0, 0, 0);
                  acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh, acc[mb], 0, 0, 0);
                }
              }
            }
          }
        }
#ifdef DFM_CIN_STAMPS
        const unsigned long long ts1 = wall_clock64();      // all of the slab's work issued
        __syncthreads();
        const unsigned long long ts2 = wall_clock64();
        st_work += ts1 - ts0; st_barrier += ts2 - ts1;
#else
        __syncthreads();
#endif
      }
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#ifdef DFM_CIN_STAMPS
    { const unsigned long long t = wall_clock64(); st_pre += t - st_mark; }
#endif
    if (MB == 4) kloop(std::true_type{});
    else kloop(std::false_type{});
#ifdef DFM_CIN_STAMPS
    st_mark = wall_clock64();
#endif
    // ---- epilogue: ReLU, sum-pool of the direct channels, hand the next channels over ------
    const bool last = li == args.L - 1;
    // the hidden image is rewritten below: every row of the next layer's (padded) image
    if (!last) {
      const int rows_next = 2 * args.layer[li + 1].HP;
      for (int r = ly.next_count; r < rows_next; ++r)
        if (hf == (r & 1)) hid[r * kCinCols + n] = 0.f;
    }
    {
      // per-lane bases so that every row below is a compile-time offset from them
      float* ybase = ly.Y ? ly.Y + (b * ly.C + 4 * hf) * D + d : nullptr;
      float* obase = args.out + b * args.out_dim + ly.out_col + 4 * hf;
      float* hbase = hid + (4 * hf - ly.next_off) * kCinCols + n;
      const int c_lim = ly.C - 4 * hf, d_lim = ly.direct - 4 * hf;
      const int n_lo = ly.next_off - 4 * hf, n_hi = n_lo + (last ? 0 : ly.next_count);
      const int y_lo = ly.y_from - 4 * hf;        // rows below are not stored (their masks are)
      uint32_t mw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = mb * 32 + (r & 3) + 8 * (r >> 2);   // compile-time; row = rr + 4*hf
          const float y = fmaxf(acc[mb][r], 0.f);
          if (y > 0.f) mw[mb] |= 1u << ((r & 3) + 8 * (r >> 2));
          if (!(CIN_FWD_ABLATE & 4) && ybase && live && rr < c_lim && rr >= y_lo) ybase[rr * D] = y;
          if (rr >= n_lo && rr < n_hi) hbase[rr * kCinCols] = y;
          // sum-pool over the D columns of each sample (D consecutive lanes; xor partners share
          // the lane half, hence the row, so the branch is taken pairwise)
          if (rr < d_lim) {
            float sum = group_sum<(D < 16 ? D : 16)>(live ? y : 0.f);
            if (D == 32) sum += __shfl_xor(sum, 16, kWave);
            if (live && d == 0) obase[rr] = sum;
          }
        }
      }
      if (ly.mask) {
        // the two lane halves hold the rows with bit 2 clear / set: OR them, one 16-byte store per column
        uint4 m;
        uint32_t* mp = reinterpret_cast<uint32_t*>(&m);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const uint32_t mine = mw[mb] << (4 * hf);
          mp[mb] = mine | static_cast<uint32_t>(__shfl_xor(static_cast<int>(mine), 32, kWave));
        }
        if (live && hf == 0) reinterpret_cast<uint4*>(ly.mask)[col] = m;
      }
    }
    (void)SPT;
#ifdef DFM_CIN_STAMPS
    { const unsigned long long t = wall_clock64(); st_epi += t - st_mark; st_mark = t; }
#endif
  }
#ifdef DFM_CIN_STAMPS
  if (args.stamps && lane == 0) {
    unsigned long long* p = args.stamps + (static_cast<size_t>(blockIdx.x) * kCinWaves + wave) * 8;
    p[0] = st_begin; p[1] = wall_clock64(); p[2] = st_work; p[3] = st_barrier; p[4] = st_pre; p[5] = st_epi;
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------
bool cin_mfma_supported(int F, int D, const int* C, const int* H, int L) {
  if (D != 16 && D != 8 && D != 32) return false;
  if (F > 40 || L > kCinMaxLayers) return false;
  for (int i = 0; i < L; ++i)
    if (C[i] > 128 || H[i] > 128) return false;
  return true;
}

size_t cin_mfma_packed_elems(int H, int F, int C) {
  const int HP = (H + 1) / 2, FG = (F + 7) / 8, MB = (C + 31) / 32;
  return static_cast<size_t>(HP) * FG * MB * 64 * 8;
}

int cin_mfma_pack(const float* W, int C, int H, int F, __bf16* hi, __bf16* lo, hipStream_t st) {
  const int HP = (H + 1) / 2, FG = (F + 7) / 8, MB = (C + 31) / 32;
  const size_t total = cin_mfma_packed_elems(H, F, C);
  hipLaunchKernelGGL(cin_pack_weights, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, W, C,
                     H, F, HP, FG, MB, hi, lo);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

// all layers' forward fragments: hi / lo + offs[i] (elements) receive layer i
int cin_mfma_pack_all(const float* const* W, const int* C, const int* H, int L, int F, __bf16* hi, __bf16* lo,
                      const size_t* offs, hipStream_t st) {
  DFM_REQUIRE(L > 0 && L <= kCinMaxLayers, "1..%d layers", kCinMaxLayers);
  CinPackJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  jobs.count = L; jobs.F = F; jobs.FG = (F + 7) / 8;
  int blocks = 0;
  for (int i = 0; i < L; ++i) {
    jobs.job[i] = CinPackJob{W[i], hi + offs[i], lo + offs[i], C[i], H[i], (H[i] + 1) / 2, (C[i] + 31) / 32};
    jobs.first_block[i] = blocks;
    blocks += static_cast<int>((cin_mfma_packed_elems(H[i], F, C[i]) + 255) / 256);
  }
  jobs.first_block[L] = blocks;
  hipLaunchKernelGGL(cin_pack_weights_all, dim3(blocks), dim3(256), 0, st, jobs);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

template <int D, int FG>
static int launch_fwd(const CinMfmaArgs& args, bool split, hipStream_t st) {
  const int64_t ncols = args.B * D;
  const int64_t blocks = (ncols + kCinWaves * kCinCols - 1) / (kCinWaves * kCinCols);
  const size_t lds = 2 * 2 * (static_cast<size_t>(FG) * 4 * 64 * 16) + sizeof(float) * kCinWaves * args.hid_rows * kCinCols;
  DFM_REQUIRE(lds <= 160 * 1024, "CIN MFMA kernel needs %zu bytes of LDS", lds);
  if (split) {
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwd_mfma<D, FG, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((cin_fwd_mfma<D, FG, true>), dim3(static_cast<unsigned>(blocks)), dim3(kCinWaves * 64),
                       lds, st, args);
  } else {
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwd_mfma<D, FG, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((cin_fwd_mfma<D, FG, false>), dim3(static_cast<unsigned>(blocks)), dim3(kCinWaves * 64),
                       lds, st, args);
  }
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int cin_mfma_forward(const CinMfmaArgs& args, int D, bool split, hipStream_t st) {
  const int FG = (args.F + 7) / 8;
#define DFM_CIN_CASE(DD, GG) \
  if (D == DD && FG == GG) return launch_fwd<DD, GG>(args, split, st);
  DFM_CIN_CASE(16, 1) DFM_CIN_CASE(16, 2) DFM_CIN_CASE(16, 3) DFM_CIN_CASE(16, 4) DFM_CIN_CASE(16, 5)
  DFM_CIN_CASE(8, 1) DFM_CIN_CASE(8, 2) DFM_CIN_CASE(8, 3) DFM_CIN_CASE(8, 4) DFM_CIN_CASE(8, 5)
  DFM_CIN_CASE(32, 1) DFM_CIN_CASE(32, 2) DFM_CIN_CASE(32, 3) DFM_CIN_CASE(32, 4) DFM_CIN_CASE(32, 5)
#undef DFM_CIN_CASE
  return fail(DFM_ERR_UNSUPPORTED, "no MFMA CIN kernel for D=%d, F=%d", D, args.F);
}

}  // namespace dfm
