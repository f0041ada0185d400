// CIN backward on the gfx950 matrix cores (autograd of deepfm/models/layers/cin.py:66-105).
//
// Per layer, with dY = relu'(Y) * upstream (upstream = pooled gradient broadcast over d on the
// direct channels + d hidden_{i+1} on the channels that fed the next layer):
//   G[k,n]        = sum_c W[c,k] dY[c,n]                      k = (h,f), n = (b,d)
//   d hidden[h,n] = sum_f x0[f,n] G[(h,f),n]
//   d x0[f,n]    += sum_h hidden[h,n] G[(h,f),n]
//   dW[c,k]       = sum_n dY[c,n] hidden[h,n] x0[f,n],   db[c] = sum_n dY[c,n]
//
// Kernel 1 (cin_dgrad_mfma): like the forward, everything except dW is column-local, so one
// wave owns 32 columns and walks the layers in reverse.  MFMA rows are the k index in blocks
// of 32 = 4 hidden rows x 8 fields (one field group), the reduction runs over the channels c
// (the dY fragments of the column stay in registers for the whole layer), and each 32x32 G tile
// is consumed straight out of the accumulator: a lane holds, for its column, 4 f x 4 h values,
// which it folds into 4 d-hidden partial sums (combined across the two lane halves once per
// 4 hidden rows) and 4 lane-private d-x0 registers per field group.  G is never stored.
// Kernel 2 (cin_wgrad_mfma): dW as a GEMM with rows c, columns k and the reduction over
// n = (b,d): one k-step is one sample (16 d), the A operand is dY (split to bf16 hi/lo on its way
// into LDS), the B operand hidden*x0 is generated in registers.  Split over batch slices,
// partial slabs reduced in a fixed order (bitwise reproducible).
// Numerics: the same bf16 x 3 split as the forward (SPLIT) or plain bf16 (throughput mode).
#include <type_traits>

#include "common.h"

using namespace dfm;

namespace dfm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CIN_BWD_ABLATE
#define CIN_BWD_ABLATE 0      // timing-only ablations (tools/build_variant.sh); 0 in the product
#endif
constexpr int kBwdMaxLayers = 8;
#ifndef CIN_BWD_WAVES
#define CIN_BWD_WAVES 4
#endif
constexpr int kBwdWaves = CIN_BWD_WAVES;
constexpr int kBwdCols = 32;

struct CinBwdLayer {
  const __bf16* wt_hi;     // packed (HQ*FG, KS, 64, 8): W^T fragments
  const __bf16* wt_lo;
  const float* Y;          // (B, C, D) post-ReLU activations of this layer (read only when `mask` is null)
  const uint32_t* mask;    // (B*D, 4) ReLU masks written by cin_fwd_mfma: bit c of a column's 128 = (Y[c] > 0)
  const float* hidden;     // hidden input of this layer: x0 (layer 0) or Y_{i-1} + next_off*D
  int64_t hidden_stride;   // floats between samples of `hidden`
  float* dY;               // (B, C, D) fp32 out, for the weight gradient
  int C, H, HQ, KS, direct, next_off, next_count, out_col;
};
struct CinBwdArgs {
  const float* x0;
  const float* g_out;
  float* g_x0;
  int64_t B;
  int F, L, out_dim, dh_rows;
  CinBwdLayer layer[kBwdMaxLayers];
};

// W (C, H*F) fp32 -> W^T fragments [blk = hq*FG+fg][ks][lane = hf*32 + r][j]:
//   value = W[c = ks*16 + 8*hf + j][h = 4*hq + (r>>3)][f = fg*8 + (r&7)]
__device__ __forceinline__ void cin_pack_wt_body(int wg, const float* __restrict__ W, int C, int H, int F, int HQ,
                                                 int FG, int KS, __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const int64_t t = static_cast<int64_t>(wg) * 256 + threadIdx.x;
  const int64_t total = static_cast<int64_t>(HQ) * FG * KS * 64 * 8;
  if (t >= total) return;
  const int j = t & 7;
  const int lane = (t >> 3) & 63;
  const int64_t rest = t >> 9;
  const int ks = static_cast<int>(rest % KS);
  const int blk = static_cast<int>(rest / KS);
  const int hq = blk / FG, fg = blk % FG;
  const int r = lane & 31, hf = lane >> 5;
  const int c = ks * 16 + 8 * hf + j, h = 4 * hq + (r >> 3), f = fg * 8 + (r & 7);
  float v = 0.f;
  if (c < C && h < H && f < F) v = W[static_cast<int64_t>(c) * H * F + h * F + f];
  const __bf16 vh = static_cast<__bf16>(v);
  hi[t] = vh;
  lo[t] = static_cast<__bf16>(v - static_cast<float>(vh));
}
__global__ __launch_bounds__(256) void cin_pack_wt(const float* __restrict__ W, int C, int H, int F, int HQ,
                                                   int FG, int KS, __bf16* __restrict__ hi,
                                                   __bf16* __restrict__ lo) {
  cin_pack_wt_body(blockIdx.x, W, C, H, F, HQ, FG, KS, hi, lo);
}
// every layer in one launch (see cin_pack_weights_all in cin_mfma.hip)
struct CinPackWtJobs {
  const float* W[kBwdMaxLayers];
  __bf16* hi[kBwdMaxLayers];
  __bf16* lo[kBwdMaxLayers];
  int C[kBwdMaxLayers], H[kBwdMaxLayers], HQ[kBwdMaxLayers], KS[kBwdMaxLayers];
  int first_block[kBwdMaxLayers + 1];
  int count, F, FG;
};
__global__ __launch_bounds__(256) void cin_pack_wt_all(CinPackWtJobs jobs) {
  int i = 0;
  while (i + 1 < jobs.count && static_cast<int>(blockIdx.x) >= jobs.first_block[i + 1]) ++i;
  cin_pack_wt_body(blockIdx.x - jobs.first_block[i], jobs.W[i], jobs.C[i], jobs.H[i], jobs.F, jobs.HQ[i], jobs.FG,
                   jobs.KS[i], jobs.hi[i], jobs.lo[i]);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int D, int FG, bool SPLIT>
__global__ __launch_bounds__(kBwdWaves * 64, 2) void cin_dgrad_mfma(CinBwdArgs args) {
  constexpr int kThreads = kBwdWaves * 64;
  constexpr int SLAB = 8 /*KS max*/ * 64 * 16;           // bytes of one hi (or lo) block slab
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* wbuf = lds_raw;                          // [2 buffers][hi, lo][KS*64][16 B]
  float* dh_all = reinterpret_cast<float*>(lds_raw + 2 * 2 * SLAB);
  const int lane = lane_id();
  const int wave = wave_id_uniform();
  const int tid = threadIdx.x;
  float* dH = dh_all + static_cast<size_t>(wave) * args.dh_rows * kBwdCols;
  const int n = lane & 31, hf = lane >> 5;
  const int64_t col = (static_cast<int64_t>(blockIdx.x) * kBwdWaves + wave) * kBwdCols + n;
  const bool live = col < args.B * D;
  const int64_t b = live ? col / D : 0;
  const int d = static_cast<int>(col % D);
  const int F = args.F;

  // this lane's 4 fields of every field group: f = fg*8 + 4*hf + j
  float x0q[FG * 4], dx0[FG * 4];
#pragma unroll
  for (int i = 0; i < FG * 4; ++i) {
    const int f = (i >> 2) * 8 + 4 * hf + (i & 3);
    x0q[i] = (live && f < F) ? args.x0[(b * F + f) * D + d] : 0.f;
    dx0[i] = 0.f;
  }

  for (int li = args.L - 1; li >= 0; --li) {
    const CinBwdLayer ly = args.layer[li];
    const int KS = ly.KS;
    // ---- dY fragments of this column (B operand, resident for the whole layer) -------------
    bf16x8 dyh[8], dyl[8];
    {
      const float* ybase = ly.Y + (b * ly.C + 8 * hf) * D + d;
      const float* gbase = args.g_out + b * args.out_dim + ly.out_col + 8 * hf;
      float* dybase = ly.dY + (b * ly.C + 8 * hf) * D + d;
      const float* hbase = dH + (8 * hf - ly.next_off) * kBwdCols + n;
      const int c_lim = ly.C - 8 * hf, d_lim = ly.direct - 8 * hf;
      const int n_lo = ly.next_off - 8 * hf, n_hi = n_lo + (li < args.L - 1 ? ly.next_count : 0);
      // ReLU mask of the column: one 16-byte load (64 scalar loads of Y before round 3: ~15 us per layer)
      uint32_t mh[4] = {0u, 0u, 0u, 0u};
      if (ly.mask && live) {
        const uint4 m = reinterpret_cast<const uint4*>(ly.mask)[col];
        mh[0] = m.x >> (8 * hf); mh[1] = m.y >> (8 * hf); mh[2] = m.z >> (8 * hf); mh[3] = m.w >> (8 * hf);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = ks * 16 + j;                     // compile-time; c = cc + 8*hf
          float g = 0.f;
          if (live && cc < c_lim && ks < KS) {
            if (cc < d_lim) g = gbase[cc];
            if (cc >= n_lo && cc < n_hi) g += hbase[cc * kBwdCols];
            if (ly.mask) g = ((mh[ks >> 1] >> (16 * (ks & 1) + j)) & 1u) ? g : 0.f;   // bit c = 16 ks + 8 hf + j
            else if (!(CIN_BWD_ABLATE & 2)) g = ybase[cc * D] > 0.f ? g : 0.f;
            if (!(CIN_BWD_ABLATE & 1)) dybase[cc * D] = g;
          }
          dyh[ks][j] = static_cast<__bf16>(g);
          if (SPLIT) dyl[ks][j] = static_cast<__bf16>(g - static_cast<float>(dyh[ks][j]));
        }
      }
    }
    // ---- G tiles, consumed from the accumulator ----------------------------------------------
    const int nblk = ly.HQ * FG;
    // Weight pipeline as in the forward: block t is loaded to registers during block t-2,
    // written to LDS at the start of block t-1, read during block t.
    // (named registers, not arrays: the compiler kept `uint4 sh[2]` in scratch memory, which put a
    // scratch round trip and a full wait on the global load into every block)
    uint4 sh0 = {}, sh1 = {}, sl0 = {}, sl1 = {};
    const int p0 = tid, p1 = kThreads + tid;
    const bool has0 = p0 < KS * 64, has1 = p1 < KS * 64;
    auto stage_load = [&](int blk) {
      const int64_t e = static_cast<int64_t>(blk) * KS * 64;
      if (has0) {
        sh0 = reinterpret_cast<const uint4*>(ly.wt_hi)[e + p0];
        if (SPLIT) sl0 = reinterpret_cast<const uint4*>(ly.wt_lo)[e + p0];
      }
      if (has1) {
        sh1 = reinterpret_cast<const uint4*>(ly.wt_hi)[e + p1];
        if (SPLIT) sl1 = reinterpret_cast<const uint4*>(ly.wt_lo)[e + p1];
      }
    };
    auto stage_store = [&](int buf) {
      unsigned char* base = wbuf + buf * 2 * SLAB;
      if (has0) {
        reinterpret_cast<uint4*>(base)[p0] = sh0;
        if (SPLIT) reinterpret_cast<uint4*>(base + SLAB)[p0] = sl0;
      }
      if (has1) {
        reinterpret_cast<uint4*>(base)[p1] = sh1;
        if (SPLIT) reinterpret_cast<uint4*>(base + SLAB)[p1] = sl1;
      }
    };
    __syncthreads();
    stage_load(0);
    stage_store(0);
    if (nblk > 1) stage_load(1);
    __syncthreads();
    // FULL (KS == 8: a 128-channel layer): no branch inside a block, the A fragments are requested
    // four k-steps at a time and the waits are counted; the generic path tests `ks < KS` around every
    // k-step (read - wait - MFMA each time).  The explicit wait in front of the loop only tells the
    // compiler that the layer preamble's loads are complete: without it the loop header inherits them
    // as pending and the first block of every iteration waits for vmcnt(0) right after issuing its
    // own weight loads.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    auto blocks = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      int blk = 0;
      for (int hq = 0; hq < ly.HQ; ++hq) {
        float hv[4], dhq[4];
#pragma unroll
        for (int hl = 0; hl < 4; ++hl) {
          const int h = 4 * hq + hl;
          hv[hl] = (live && h < ly.H) ? ly.hidden[b * ly.hidden_stride + h * D + d] : 0.f;
          dhq[hl] = 0.f;
        }
        // Two accumulators, alternating (round 3): the 32 FMAs that fold block fg - 1's G tile into d hidden / d x0
        // issue BEHIND block fg's 24 MFMAs, so the matrix pipe keeps running while the VALU consumes the previous
        // tile (one accumulator: MFMA chain -> wait -> fold -> barrier, 60 us of the launch by ablation).  Only the
        // last block of a hidden-row quad folds in the open.
        f32x16 accs[2];
        auto fold = [&](const f32x16& acc, auto fg_tag) {
          constexpr int fg = decltype(fg_tag)::value;
          if (CIN_BWD_ABLATE & 4) {
            dhq[0] += acc[0]; dx0[fg * 4] += acc[5];
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dhq[r >> 2] = fmaf(x0q[fg * 4 + (r & 3)], acc[r], dhq[r >> 2]);
              dx0[fg * 4 + (r & 3)] = fmaf(hv[r >> 2], acc[r], dx0[fg * 4 + (r & 3)]);
            }
          }
        };
        static_for<0, FG>([&](auto fg_tag) {
          constexpr int fg = decltype(fg_tag)::value;
          const int cur = blk & 1;
          const unsigned char* base = wbuf + cur * 2 * SLAB;
          f32x16& acc = accs[fg & 1];
          acc = f32x16{};
          if constexpr (FULL) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ah[q] = reinterpret_cast<const bf16x8*>(base)[q * 64 + lane];
              if (SPLIT) al[q] = reinterpret_cast<const bf16x8*>(base + SLAB)[q * 64 + lane];
            }
            if (!(CIN_BWD_ABLATE & 16)) {
              if (blk + 1 < nblk) stage_store(cur ^ 1);          // block blk+1 (loaded one block ago)
              if (blk + 2 < nblk) stage_load(blk + 2);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int ks = half * 4 + q;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], dyh[ks], acc, 0, 0, 0);
                if (SPLIT) {
                  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], dyl[ks], acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[q], dyh[ks], acc, 0, 0, 0);
                }
              }
              if (half == 0 && !(CIN_BWD_ABLATE & 8)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  ah[q] = reinterpret_cast<const bf16x8*>(base)[(4 + q) * 64 + lane];
                  if (SPLIT) al[q] = reinterpret_cast<const bf16x8*>(base + SLAB)[(4 + q) * 64 + lane];
                }
              }
            }
          } else {
            if (blk + 1 < nblk) stage_store(cur ^ 1);
            if (blk + 2 < nblk) stage_load(blk + 2);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              if (ks < KS) {
                const bf16x8 a_h = reinterpret_cast<const bf16x8*>(base)[ks * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, dyh[ks], acc, 0, 0, 0);
                if (SPLIT) {
                  const bf16x8 a_l = reinterpret_cast<const bf16x8*>(base + SLAB)[ks * 64 + lane];
                  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, dyl[ks], acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, dyh[ks], acc, 0, 0, 0);
                }
              }
            }
          }
          // accumulator register r: hidden row 4*hq + (r>>2), field fg*8 + 4*hf + (r&3)
          if constexpr (fg > 0) fold(accs[(fg - 1) & 1], std::integral_constant<int, fg - 1>{});
          if constexpr (fg == FG - 1) fold(acc, fg_tag);
          __syncthreads();
          ++blk;
        });
        // the other lane half holds the other 4 fields of every group
#pragma unroll
        for (int hl = 0; hl < 4; ++hl) {
          const float tot = dhq[hl] + __shfl_xor(dhq[hl], 32, kWave);
          if ((hl & 1) == hf) dH[(4 * hq + hl) * kBwdCols + n] = tot;
        }
      }
    };
    if (KS == 8) blocks(std::true_type{});
    else blocks(std::false_type{});
    if (li == 0) {   // hidden_0 is x0 itself: its gradient joins d x0
#pragma unroll
      for (int i = 0; i < FG * 4; ++i) {
        const int f = (i >> 2) * 8 + 4 * hf + (i & 3);
        if (f < 4 * ly.HQ) dx0[i] += dH[f * kBwdCols + n];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FG * 4; ++i) {
    const int f = (i >> 2) * 8 + 4 * hf + (i & 3);
    if (live && f < F) args.g_x0[(b * F + f) * D + d] = dx0[i];
  }
}

// ---- weight gradient -----------------------------------------------------------------------
struct CinWgradArgs {
  const float* dY;         // (B, C, 16) fp32, straight from the dgrad kernel
  int C;
  const float* x0;         // (B, F, 16)
  const float* hidden;
  int64_t hidden_stride;
  float* slabs;            // (slices, MB*32, KT*32) fp32 partials
  int64_t B;
  int F, H, MB, KT, slices, FP;  // KT: 32-wide column tiles over k' = h*FP + f (FP = padded F)
  int bias_col;                  // 1: column k' = F carries the bias gradient (needs FP > F)
  int64_t per;                   // samples per batch slice (even when D == 8: a k-step then spans two samples)
  unsigned long long* stamps;    // tools/microbench_cin_wgrad only (DFM_CIN_STAMPS build): per-wave phase sums
};

// grid (ceil(KT/4), slices); a wave owns one 32-column tile of k', all C rows; one pipeline step = one
// sample (= one 32x32x16 k-step per row block).  dY fragments (A) go global -> registers -> LDS one step
// ahead (split to bf16 hi / lo on the way), the hidden / x0 rows (B) global -> registers one step ahead.
//
// The loop is written to be ISSUE-lean (round 2: it ran 124 VALU + 66 SALU instructions per 12 MFMAs, 57 of
// them v_mov — zero fills and exec-masked paths around conditional loads — and every unit's time simply
// added up: 1.4 us per sample and workgroup).  Now every load is unconditional: rows / columns / samples
// that do not exist are CLAMPED to ones that do (their accumulator rows and slab columns are never read
// by the reduce kernel), the bias column reads a constant row of ones through a per-lane stride of 0, the
// two LDS buffers are two unrolled copies of the step (immediate offsets), and three workgroups share a CU.
// 16-byte load through a GLOBAL-address-space pointer: a select between two objects (kWgOnes / a tensor)
// leaves the compiler with a generic pointer, and flat loads count against lgkmcnt too — every wait for an
// LDS read would then also wait for the HBM loads in flight
typedef float gf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4g(const float* p) {
  const gf4 v = *reinterpret_cast<const __attribute__((address_space(1))) gf4*>(reinterpret_cast<uintptr_t>(p));
  return float4{v.x, v.y, v.z, v.w};
}
__device__ const float kWgOnes[32] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                      1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
#ifndef WG_OCC
#define WG_OCC 2
#endif
constexpr int kWgOcc = WG_OCC;          // workgroups per CU (<= 256 VGPRs)
#ifndef WG_LB
#define WG_LB 2
#endif
constexpr int kWgBound = WG_LB;
#ifndef WG_DEPTH
#define WG_DEPTH 1
#endif
constexpr int kWgDepth = WG_DEPTH;      // samples of global-load look-ahead (build-time tunables: tools/microbench_cin_wgrad)


// D = embedding dim (8, 16 or 32): one pipeline step is one MFMA k-step = 16 consecutive (b, d) columns —
// a sample (D = 16), half a sample (D = 32) or two samples (D = 8, lane half hf takes sample 2s + hf).
template <bool SPLIT, int D>
__global__ __launch_bounds__(256, kWgBound) void cin_wgrad_mfma(CinWgradArgs a) {
  static_assert(D == 8 || D == 16 || D == 32, "k-steps of 16 columns need D in {8, 16, 32}");
  constexpr int SLAB = 4 * 64 * 16;                       // one sample: 4 row blocks x 64 lanes x 16 B (hi or lo)
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * 2 * SLAB];   // [buf][hi, lo]
  // B operand through LDS (round 3, D == 16): one k-step is one sample, and a workgroup's 128 k' columns need only
  // the sample's x0 rows (F x 64 B, contiguous) and 128 / FP + 1 hidden rows — 2.9 KB at F = 39, loaded ONCE per workgroup, coalesced,
  // one 16-byte load per thread, instead of 64 scattered bytes per lane (16 KB per workgroup and sample through
  // the texture path: 77 us of the three launches by ablation).  Rows are 80 bytes apart: 16 consecutive rows
  // read with ds_read_b128 touch every bank once.  Row kOnes holds ones (the bias column's operand).
  constexpr bool LB = (D == 16) && !(CIN_BWD_ABLATE & 128);
  // (hidden rows per workgroup: 128 / FP + 1 <= 17 for FP >= 8; 5 at the Criteo shape's FP = 40)
  constexpr int kBRow = 20, kXRows = 40, kHRows = 17, kOnes = kXRows + kHRows, kBRows = kOnes + 1;
  static_assert(4 * kOnes <= 256, "one staging thread per 16 bytes of the B image");
  __shared__ __attribute__((aligned(16))) float bbuf[LB ? 2 * kBRows * kBRow : 4];
  const int lane = lane_id(), wave = wave_id_uniform(), tid = threadIdx.x;
  const int kt = blockIdx.x * 4 + wave;
  const int kcol = kt * 32 + (lane & 31), hf = lane >> 5;
  const int h = kcol / a.FP, f = kcol % a.FP;
  const bool kvalid = kt < a.KT && h < a.H && f < a.F;
  // bias gradient for free: the first padding column of hidden row 0 (k' = F, present when F is not a
  // multiple of 8) multiplies dY by ones, so its slab column is db[c] = sum_{b,d} dY[c,b,d]
  const bool kbias = a.bias_col && kt < a.KT && h == 0 && f == a.F;
  const int64_t per = a.per;
  const int64_t b0 = blockIdx.y * per, b1 = b0 + per < a.B ? b0 + per : a.B;
  const int n = b1 > b0 ? static_cast<int>((b1 - b0) * D / 16) : 0;       // k-steps of this slice
  f32x16 acc[4] = {};
  // A operand: fragment (mb, lane = hf*32 + r) of k-step s is dY[b][32 mb + r][d .. d + 7] with (b, d) the
  // sample and offset of column 16 s + 8 hf — 32 contiguous bytes of fp32; this thread stages row s_c of
  // every k-step (rows past C: row C - 1 again)
  const int s_mb = tid >> 6, s_lane = tid & 63;
  const int s_c = min(32 * s_mb + (s_lane & 31), a.C - 1);
  const float* pa = a.dY + (b0 * a.C + s_c) * D;                           // + sample * C * D + d
  const int64_t stride_a = static_cast<int64_t>(a.C) * D;
  // (sample relative to b0, offset inside the sample) of the 8 columns a lane half takes in k-step s, split
  // into a wave-uniform part that moves with s and a per-lane constant: D >= 16: sample 16 s / D, offset
  // 16 s % D + 8 half; D = 8: sample 2 s + half, offset 0
  auto sample_u = [](int s) { return D >= 16 ? (16 * s) / D : 2 * s; };
  auto offset_u = [](int s) { return D >= 16 ? (16 * s) % D : 0; };
  const uint32_t lane_a = D >= 16 ? 8u * (s_lane >> 5) : static_cast<uint32_t>((s_lane >> 5) * stride_a);
  // B operand: z[j] = hidden[b][h][8 hf + j] * x0[b][f][8 hf + j] of this lane's column; columns that do
  // not exist read (h, f) = (0, 0), the bias column reads ones twice with stride 0
  const float* hp0 = kbias ? kWgOnes : a.hidden + b0 * a.hidden_stride + (kvalid ? h : 0) * D;
  const float* xp0 = kbias ? kWgOnes : a.x0 + (b0 * a.F + (kvalid ? f : 0)) * D;
  const uint32_t hs = kbias ? 0u : static_cast<uint32_t>(a.hidden_stride), xs = kbias ? 0u : static_cast<uint32_t>(a.F * D);
  const uint32_t lane_h = D >= 16 ? 8u * hf : hf * hs, lane_x = D >= 16 ? 8u * hf : hf * xs;   // per-lane constants
  // LDS-staged B: this thread's row / quarter of the staged image, and this lane's two rows of it
  const int h_lo = (blockIdx.x * 4 * 32) / a.FP;
  const int sb_row = tid >> 2, sb_q = tid & 3;
  const bool sb_on = LB && sb_row < kOnes;
  const bool sb_x = sb_row < kXRows;
  const float* pb = sb_x ? a.x0 + (b0 * a.F + min(sb_row, a.F - 1)) * D + 4 * sb_q
                         : a.hidden + b0 * a.hidden_stride + min(h_lo + sb_row - kXRows, a.H - 1) * D + 4 * sb_q;
  const int64_t sb_stride = sb_x ? static_cast<int64_t>(a.F) * D : a.hidden_stride;
  const int lb_x = (kvalid ? f : (kbias ? kOnes : 0)) * kBRow + 8 * hf;
  const int lb_h = (kvalid ? kXRows + (h - h_lo) : (kbias ? kOnes : kXRows)) * kBRow + 8 * hf;
  if (LB) {
    for (int i = tid; i < 2 * kBRows * kBRow; i += 256) bbuf[i] = 1.f;     // the ones rows; the rest is overwritten
  }
  float4 rb[kWgDepth];
  auto load_bs = [&](int s, auto slot_tag) {      // staged B of sample s -> register slot
    constexpr int SL = decltype(slot_tag)::value;
    if (sb_on) rb[SL] = ld4g(pb + s * sb_stride);
  };
  auto store_bs = [&](float* base, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    if (sb_on) *reinterpret_cast<float4*>(base + sb_row * kBRow + 4 * sb_q) = rb[SL];
  };
  // Global loads run kWgDepth samples ahead of their use, in a ring of register slots (compile-time slot
  // numbers: the loop is unrolled by the ring size).  Measured with tools/microbench_cin_wgrad (H = 64
  // layer, kernel + slab reduce): depth 1 168 us, depth 2 174, depth 4 209, depth 6 198 (one wave per
  // SIMD) — more loads in flight only lengthen the TA queue.  Counters at depth 1 (rocprofv3 --pmc, three
  // workgroups per CU): MFMA busy 45 %, TA busy 57 %, LDS array 23 %, VALU issue 39 %, waves parked in
  // s_waitcnt / s_barrier 50 % of their cycles, no LDS bank conflicts: no unit is saturated, the step
  // (LDS reads -> convert + LDS write -> split the B product -> 12 MFMAs -> barrier) is a dependent chain
  // and two or three waves per SIMD do not cover it.  What would: a 64-column tile per wave (24 MFMAs per
  // barrier and per A fragment read) — not done.
  float4 rf0[kWgDepth], rf1[kWgDepth], hb0[kWgDepth], hb1[kWgDepth], xb0[kWgDepth], xb1[kWgDepth];
  auto load_a = [&](int s, auto slot_tag) {     // s: sample of the slice, clamped by the caller
    constexpr int SL = decltype(slot_tag)::value;
    const float* p = pa + (sample_u(s) * stride_a + offset_u(s)) + lane_a;
    rf0[SL] = ld4g(p);
    rf1[SL] = ld4g(p + 4);
  };
  auto load_b = [&](int s, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    const int bs = sample_u(s), od = offset_u(s);
    const float* hp = hp0 + (static_cast<uint64_t>(bs) * hs + od) + lane_h;
    const float* xp = xp0 + (static_cast<uint64_t>(bs) * xs + od) + lane_x;
    hb0[SL] = ld4g(hp); hb1[SL] = ld4g(hp + 4); xb0[SL] = ld4g(xp); xb1[SL] = ld4g(xp + 4);
  };
  auto split8 = [](const float4& v0, const float4& v1, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = static_cast<__bf16>(v[j]);
      if (SPLIT) lo[j] = static_cast<__bf16>(v[j] - static_cast<float>(hi[j]));
    }
  };
  auto store_a = [&](unsigned char* base, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    bf16x8 hi, lo;
    split8(rf0[SL], rf1[SL], hi, lo);
    reinterpret_cast<bf16x8*>(base)[s_mb * 64 + s_lane] = hi;
    if (SPLIT) reinterpret_cast<bf16x8*>(base + SLAB)[s_mb * 64 + s_lane] = lo;
  };
#ifdef DFM_CIN_STAMPS
  unsigned long long st_a = 0, st_b = 0, st_m = 0, st_bar = 0;
  const unsigned long long st_begin = wall_clock64();
#define WG_STAMP(var) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = wall_clock64(); var += t_ - st_t; st_t = t_; }
#else
#define WG_STAMP(var)
#endif
  // step s: slot s % kWgDepth holds B(s) and, from here on, A(s + kWgDepth) / B(s + kWgDepth);
  // slot (s + 1) % kWgDepth holds A(s + 1), which moves to the other LDS buffer now
  auto step = [&](int s, auto unroll_tag, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int U = decltype(unroll_tag)::value;        // position in the unrolled loop body
    constexpr int SL = U % kWgDepth, SN = (U + 1) % kWgDepth;
    const std::integral_constant<int, SL> slot_tag{};
    unsigned char* cur = wbuf + (U & 1) * 2 * SLAB;
    unsigned char* nxt = wbuf + ((U & 1) ^ 1) * 2 * SLAB;
#ifdef DFM_CIN_STAMPS
    unsigned long long st_t = wall_clock64();
#endif
    bf16x8 ah[4];
    if constexpr (FULL) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) ah[mb] = reinterpret_cast<const bf16x8*>(cur)[mb * 64 + lane];
    }
    float4 h0, h1, x0v, x1v;
    if constexpr (LB) {
      const float* bc = bbuf + (U & 1) * kBRows * kBRow;
      h0 = *reinterpret_cast<const float4*>(bc + lb_h); h1 = *reinterpret_cast<const float4*>(bc + lb_h + 4);
      x0v = *reinterpret_cast<const float4*>(bc + lb_x); x1v = *reinterpret_cast<const float4*>(bc + lb_x + 4);
    }
    if (!(CIN_BWD_ABLATE & 64)) {
      store_a(nxt, std::integral_constant<int, SN>{});     // sample s + 1 (the last step stores a copy nobody reads)
      if constexpr (LB) store_bs(bbuf + ((U & 1) ^ 1) * kBRows * kBRow, std::integral_constant<int, SN>{});
      load_a(min(s + 1 + kWgDepth, n - 1), std::integral_constant<int, SN>{});
      if constexpr (LB) load_bs(min(s + 1 + kWgDepth, n - 1), std::integral_constant<int, SN>{});
    }
#ifdef DFM_CIN_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WG_STAMP(st_a)
    bf16x8 bh, bl;
    if constexpr (LB) {
      const float4 z0 = {h0.x * x0v.x, h0.y * x0v.y, h0.z * x0v.z, h0.w * x0v.w};
      const float4 z1 = {h1.x * x1v.x, h1.y * x1v.y, h1.z * x1v.z, h1.w * x1v.w};
      split8(z0, z1, bh, bl);
    } else {
      const float4 z0 = {hb0[SL].x * xb0[SL].x, hb0[SL].y * xb0[SL].y, hb0[SL].z * xb0[SL].z, hb0[SL].w * xb0[SL].w};
      const float4 z1 = {hb1[SL].x * xb1[SL].x, hb1[SL].y * xb1[SL].y, hb1[SL].z * xb1[SL].z, hb1[SL].w * xb1[SL].w};
      split8(z0, z1, bh, bl);
      if (!(CIN_BWD_ABLATE & 32)) load_b(min(s + kWgDepth, n - 1), slot_tag);
    }
    WG_STAMP(st_b)
    if constexpr (FULL) {
      // hi fragments first, the lo fragments only once the hi ones are dead (16 live registers of A)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb], 0, 0, 0);
      if (SPLIT) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl, acc[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) ah[mb] = reinterpret_cast<const bf16x8*>(cur + SLAB)[mb * 64 + lane];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if (mb < a.MB) {
          const bf16x8 a_h = reinterpret_cast<const bf16x8*>(cur)[mb * 64 + lane];
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh, acc[mb], 0, 0, 0);
          if (SPLIT) {
            const bf16x8 a_l = reinterpret_cast<const bf16x8*>(cur + SLAB)[mb * 64 + lane];
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh, acc[mb], 0, 0, 0);
          }
        }
      }
    }
    WG_STAMP(st_m)
    __syncthreads();
    WG_STAMP(st_bar)
  };
  if (n > 0) {
    constexpr int kUnroll = kWgDepth % 2 ? 2 * kWgDepth : kWgDepth;   // slots and the two LDS buffers both cycle
    using S0 = std::integral_constant<int, 0>;
    load_a(0, S0{});
    if constexpr (LB) {
      static_assert(kWgDepth == 1, "the LDS-staged B operand is written for one step of look-ahead");
      load_bs(0, S0{});
      __syncthreads();                         // the ones fill above is complete before rows are written over it
      store_bs(bbuf, S0{});
      load_bs(min(1, n - 1), S0{});
    } else {
      static_for<0, kWgDepth>([&](auto i) { load_b(min(static_cast<int>(i), n - 1), i); });
    }
    store_a(wbuf, S0{});
    static_for<1, kWgDepth>([&](auto i) { load_a(min(static_cast<int>(i), n - 1), i); });
    load_a(min(kWgDepth, n - 1), S0{});
    __syncthreads();
    auto run = [&](auto full_tag) {
      int s = 0;
      for (; s + kUnroll - 1 < n; s += kUnroll) static_for<0, kUnroll>([&](auto i) { step(s + i, i, full_tag); });
      static_for<0, kUnroll - 1>([&](auto i) { if (s + i < n) step(s + i, i, full_tag); });
    };
    if (a.MB == 4) run(std::true_type{});
    else run(std::false_type{});
  }
#ifdef DFM_CIN_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    o[0] = st_begin; o[1] = wall_clock64(); o[2] = st_a; o[3] = st_b; o[4] = st_m; o[5] = st_bar; o[6] = n;
  }
#endif
  if (kt < a.KT) {
    // accumulator: col = k' column (lane&31), row = c = mb*32 + (r&3) + 8*(r>>2) + 4*hf
    float* out = a.slabs + (static_cast<int64_t>(blockIdx.y) * a.MB * 32 + 4 * hf) * (a.KT * 32) + kt * 32 + (lane & 31);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      if (mb < a.MB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = mb * 32 + (r & 3) + 8 * (r >> 2);
          out[static_cast<int64_t>(rr) * (a.KT * 32)] = acc[mb][r];
        }
      }
    }
  }
}

// dW[c][h*F+f] += sum_s slabs[s][c][h*FP+f]  (fixed order); with db: db[c] += sum_s slabs[s][c][F]
struct CinWgradReduceJob {
  const float* slabs;
  float* dW;
  float* db;
  int slices, rows_pad, cols_pad, C, H, F, FP;
};
__device__ __forceinline__ void cin_wgrad_reduce_body(int blk, const CinWgradReduceJob& j) {
  const int64_t o = static_cast<int64_t>(blk) * 256 + threadIdx.x;
  const int64_t K = static_cast<int64_t>(j.H) * j.F;
  const int64_t n = j.C * K;
  if (o >= n + (j.db ? j.C : 0)) return;
  const int c = o < n ? static_cast<int>(o / K) : static_cast<int>(o - n);
  const int k = static_cast<int>(o % K);
  const int kp = o < n ? (k / j.F) * j.FP + k % j.F : j.F;
  // 25 ... 39 slices: eight loads in flight at a time, added in slice order
  const float* p = j.slabs + static_cast<int64_t>(c) * j.cols_pad + kp;
  const int64_t stride = static_cast<int64_t>(j.rows_pad) * j.cols_pad;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= j.slices; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < j.slices; ++s) acc += p[s * stride];
  if (o < n) j.dW[o] += acc;
  else j.db[c] += acc;
}
__global__ __launch_bounds__(256) void cin_wgrad_reduce_mfma(CinWgradReduceJob j) {
  cin_wgrad_reduce_body(blockIdx.x, j);
}
// every layer's slabs in ONE launch (each layer has its own slab region): three 12 us launches -> one
constexpr int kCinReduceMaxJobs = 8;
struct CinWgradReduceJobs {
  CinWgradReduceJob job[kCinReduceMaxJobs];
  int first_block[kCinReduceMaxJobs + 1];
  int count;
};
__global__ __launch_bounds__(256) void cin_wgrad_reduce_all(CinWgradReduceJobs jobs) {
  int i = 0;
  while (i + 1 < jobs.count && static_cast<int>(blockIdx.x) >= jobs.first_block[i + 1]) ++i;
  cin_wgrad_reduce_body(blockIdx.x - jobs.first_block[i], jobs.job[i]);
}

// ---- host side ---------------------------------------------------------------------------
size_t cin_bwd_packed_wt_elems(int H, int F, int C) {
  const int HQ = (H + 3) / 4, FG = (F + 7) / 8, KS = (C + 15) / 16;
  return static_cast<size_t>(HQ) * FG * KS * 64 * 8;
}

int cin_bwd_pack_wt(const float* W, int C, int H, int F, __bf16* hi, __bf16* lo, hipStream_t st) {
  const int HQ = (H + 3) / 4, FG = (F + 7) / 8, KS = (C + 15) / 16;
  const size_t total = cin_bwd_packed_wt_elems(H, F, C);
  hipLaunchKernelGGL(cin_pack_wt, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, W, C, H, F,
                     HQ, FG, KS, hi, lo);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int cin_bwd_pack_wt_all(const float* const* W, const int* C, const int* H, int L, int F, __bf16* hi, __bf16* lo,
                        const size_t* offs, hipStream_t st) {
  DFM_REQUIRE(L > 0 && L <= kBwdMaxLayers, "1..%d layers", kBwdMaxLayers);
  CinPackWtJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  jobs.count = L; jobs.F = F; jobs.FG = (F + 7) / 8;
  int blocks = 0;
  for (int i = 0; i < L; ++i) {
    jobs.W[i] = W[i]; jobs.hi[i] = hi + offs[i]; jobs.lo[i] = lo + offs[i];
    jobs.C[i] = C[i]; jobs.H[i] = H[i]; jobs.HQ[i] = (H[i] + 3) / 4; jobs.KS[i] = (C[i] + 15) / 16;
    jobs.first_block[i] = blocks;
    blocks += static_cast<int>((cin_bwd_packed_wt_elems(H[i], F, C[i]) + 255) / 256);
  }
  jobs.first_block[L] = blocks;
  hipLaunchKernelGGL(cin_pack_wt_all, dim3(blocks), dim3(256), 0, st, jobs);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

template <int D, int FG>
static int launch_dgrad(const CinBwdArgs& args, bool split, hipStream_t st) {
  const int64_t ncols = args.B * D;
  const int64_t blocks = (ncols + kBwdWaves * kBwdCols - 1) / (kBwdWaves * kBwdCols);
  const size_t lds = 2 * 2 * (8 * 64 * 16) + sizeof(float) * kBwdWaves * args.dh_rows * kBwdCols;
  DFM_REQUIRE(lds <= 160 * 1024, "CIN dgrad kernel needs %zu bytes of LDS", lds);
  if (split) {
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dgrad_mfma<D, FG, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((cin_dgrad_mfma<D, FG, true>), dim3(static_cast<unsigned>(blocks)), dim3(kBwdWaves * 64),
                       lds, st, args);
  } else {
    DFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dgrad_mfma<D, FG, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((cin_dgrad_mfma<D, FG, false>), dim3(static_cast<unsigned>(blocks)), dim3(kBwdWaves * 64),
                       lds, st, args);
  }
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int cin_mfma_dgrad(const CinBwdArgs& args, int D, bool split, hipStream_t st) {
  const int FG = (args.F + 7) / 8;
#define DFM_CASE(GG)                                                  \
  if (D == 16 && FG == GG) return launch_dgrad<16, GG>(args, split, st); \
  if (D == 8 && FG == GG) return launch_dgrad<8, GG>(args, split, st);   \
  if (D == 32 && FG == GG) return launch_dgrad<32, GG>(args, split, st);
  DFM_CASE(1) DFM_CASE(2) DFM_CASE(3) DFM_CASE(4) DFM_CASE(5)
#undef DFM_CASE
  return fail(DFM_ERR_UNSUPPORTED, "no MFMA CIN dgrad kernel for D=%d, F=%d", D, args.F);
}

// Batch slices of the weight-gradient GEMM: the grid is (column groups) x (slices) workgroups of
// equal work, kWgOcc resident per CU, so the slice count is chosen to fill ONE round of kWgOcc x 256
// workgroup slots (20 column groups x 32 slices = 640 workgroups ran a second round at 25 %
// occupancy: 184 us where 118 us of work was needed).
constexpr int kWgradMaxSlices = 64;
constexpr int kWgSlots = kWgOcc * 256;
static int wgrad_slices(int64_t B, int KT) {
  const int cols = (KT + 3) / 4;
  int slices = kWgSlots / cols;
  slices = slices < 1 ? 1 : (slices > kWgradMaxSlices ? kWgradMaxSlices : slices);
  return B < slices ? static_cast<int>(B) : slices;
}

size_t cin_mfma_wgrad_workspace_bytes(int64_t B, int C, int H, int F) {
  const int MB = (C + 31) / 32, FP = ((F + 7) / 8) * 8;
  const int KT = (H * FP + 31) / 32;
  return sizeof(float) * static_cast<size_t>(wgrad_slices(B, KT)) * MB * 32 * KT * 32 + 512;
}

unsigned long long* g_wgrad_stamps = nullptr;     // set by tools/microbench_cin_wgrad (stamped build) only

// true: cin_mfma_wgrad also produces the bias gradient (a free padding column exists)
bool cin_mfma_wgrad_has_bias(int F) { return F % 8 != 0; }

// D == 8 pairs samples in a k-step: the batch must be even
bool cin_mfma_wgrad_supported(int64_t B, int D) { return D == 16 || D == 32 || (D == 8 && B % 2 == 0); }

static CinWgradReduceJob wgrad_reduce_job(void* workspace, float* dW, float* db, int64_t B, int F, int H, int C,
                                          int* blocks) {
  const int MB = (C + 31) / 32, FP = ((F + 7) / 8) * 8;
  const int KT = (H * FP + 31) / 32;
  const int64_t n = static_cast<int64_t>(C) * H * F + (db ? C : 0);
  *blocks = static_cast<int>((n + 255) / 256);
  return CinWgradReduceJob{static_cast<const float*>(workspace), dW, db, wgrad_slices(B, KT), MB * 32, KT * 32, C, H, F, FP};
}

// dW += dY^T (hidden (x) x0); db += sum_{b,d} dY when `db` is given (cin_mfma_wgrad_has_bias).
// reduce_now false: the batch slices' partial products stay in `workspace` — the caller gives every layer its own
// and adds them all with ONE cin_mfma_wgrad_reduce_layers launch.
int cin_mfma_wgrad(const float* dY, const float* x0, const float* hidden, int64_t hidden_stride, int64_t B,
                   int F, int H, int C, int D, float* dW, float* db, void* workspace, bool split, bool reduce_now,
                   hipStream_t st) {
  DFM_REQUIRE(!db || cin_mfma_wgrad_has_bias(F), "no padding column for the bias gradient (F = %d)", F);
  DFM_REQUIRE(cin_mfma_wgrad_supported(B, D), "no MFMA weight-gradient kernel for D = %d, B = %lld", D, (long long)B);
  const int MB = (C + 31) / 32, FP = ((F + 7) / 8) * 8;
  const int KT = (H * FP + 31) / 32;
  float* slabs = static_cast<float*>(workspace);
  const int slices = wgrad_slices(B, KT);
  CinWgradArgs a;
  a.dY = dY; a.C = C; a.x0 = x0; a.hidden = hidden; a.hidden_stride = hidden_stride; a.slabs = slabs;
  a.B = B; a.F = F; a.H = H; a.MB = MB; a.KT = KT; a.slices = slices; a.FP = FP;
  a.bias_col = db ? 1 : 0;
  a.per = (B + slices - 1) / slices;
  if (D == 8) a.per = (a.per + 1) & ~int64_t(1);
  a.stamps = g_wgrad_stamps;
  const dim3 grid((KT + 3) / 4, slices);
#define DFM_WG(DD)                                                                         \
  if (D == DD) {                                                                           \
    if (split) hipLaunchKernelGGL((cin_wgrad_mfma<true, DD>), grid, dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((cin_wgrad_mfma<false, DD>), grid, dim3(256), 0, st, a);       \
  }
  DFM_WG(8) DFM_WG(16) DFM_WG(32)
#undef DFM_WG
  DFM_LAUNCH_CHECK();
  if (!reduce_now) return DFM_OK;
  int blocks;
  const CinWgradReduceJob job = wgrad_reduce_job(workspace, dW, db, B, F, H, C, &blocks);
  hipLaunchKernelGGL(cin_wgrad_reduce_mfma, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, job);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

int cin_mfma_wgrad_max_layers_per_reduce() { return kCinReduceMaxJobs; }

// the reductions cin_mfma_wgrad(..., reduce_now = false) left open, layer i in workspaces[i]
int cin_mfma_wgrad_reduce_layers(int count, void* const* workspaces, float* const* dW, float* const* db, int64_t B,
                                 int F, const int* H, const int* C, hipStream_t st) {
  DFM_REQUIRE(count >= 0 && count <= kCinReduceMaxJobs, "at most %d layers per reduction launch", kCinReduceMaxJobs);
  if (count == 0) return DFM_OK;
  CinWgradReduceJobs js;
  js.count = count;
  js.first_block[0] = 0;
  for (int i = 0; i < count; ++i) {
    int blocks;
    js.job[i] = wgrad_reduce_job(workspaces[i], dW[i], db[i], B, F, H[i], C[i], &blocks);
    js.first_block[i + 1] = js.first_block[i] + blocks;
  }
  hipLaunchKernelGGL(cin_wgrad_reduce_all, dim3(static_cast<unsigned>(js.first_block[count])), dim3(256), 0, st, js);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

}  // namespace dfm
