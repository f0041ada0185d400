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

constexpr int kBwdMaxLayers = 8;
constexpr int kBwdWaves = 4;
constexpr int kBwdCols = 32;

struct CinBwdLayer {
  const __bf16* wt_hi;     // packed (HQ*FG, KS, 64, 8): W^T fragments
  const __bf16* wt_lo;
  const float* Y;          // (B, C, D) post-ReLU activations of this layer
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
__global__ __launch_bounds__(256) void cin_pack_wt(const float* __restrict__ W, int C, int H, int F, int HQ,
                                                   int FG, int KS, __bf16* __restrict__ hi,
                                                   __bf16* __restrict__ lo) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
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

template <int D, int FG, bool SPLIT>
__global__ __launch_bounds__(kBwdWaves * 64, 2) void cin_dgrad_mfma(CinBwdArgs args) {
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
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = ks * 16 + j;                     // compile-time; c = cc + 8*hf
          float g = 0.f;
          if (live && cc < c_lim && ks < KS) {
            if (cc < d_lim) g = gbase[cc];
            if (cc >= n_lo && cc < n_hi) g += hbase[cc * kBwdCols];
            g = ybase[cc * D] > 0.f ? g : 0.f;
            dybase[cc * D] = g;
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
    const int p0 = tid, p1 = 256 + tid;
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
#pragma unroll
        for (int fg = 0; fg < FG; ++fg, ++blk) {
          const int cur = blk & 1;
          const unsigned char* base = wbuf + cur * 2 * SLAB;
          f32x16 acc = {};
          if constexpr (FULL) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ah[q] = reinterpret_cast<const bf16x8*>(base)[q * 64 + lane];
              if (SPLIT) al[q] = reinterpret_cast<const bf16x8*>(base + SLAB)[q * 64 + lane];
            }
            if (blk + 1 < nblk) stage_store(cur ^ 1);          // block blk+1 (loaded one block ago)
            if (blk + 2 < nblk) stage_load(blk + 2);
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
              if (half == 0) {
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
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            dhq[r >> 2] = fmaf(x0q[fg * 4 + (r & 3)], acc[r], dhq[r >> 2]);
            dx0[fg * 4 + (r & 3)] = fmaf(hv[r >> 2], acc[r], dx0[fg * 4 + (r & 3)]);
          }
          __syncthreads();
        }
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
};

// grid (ceil(KT/4), slices); a wave owns one 32-column tile of k', all C rows.  One pipeline step
// = kWgStep samples; dY fragments (A) go global -> registers -> LDS one step ahead, the
// hidden / x0 pieces (B) global -> registers one step ahead.
constexpr int kWgStep = 2;

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void cin_wgrad_mfma(CinWgradArgs a) {
  constexpr int SLAB = kWgStep * 4 * 64 * 16;            // one hi (or lo) slab: samples x MB x 1 KiB
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * 2 * SLAB];   // [buf][hi,lo]
  const int lane = lane_id(), wave = wave_id_uniform(), tid = threadIdx.x;
  const int kt = blockIdx.x * 4 + wave;
  const int kcol = kt * 32 + (lane & 31), hf = lane >> 5;
  const int h = kcol / a.FP, f = kcol % a.FP;
  const bool kvalid = kt < a.KT && h < a.H && f < a.F;
  // bias gradient for free: the first padding column of hidden row 0 (k' = F, present when F is not a
  // multiple of 8) multiplies dY by ones, so its slab column is db[c] = sum_{b,d} dY[c,b,d]
  const bool kbias = a.bias_col && kt < a.KT && h == 0 && f == a.F;
  const int64_t per = (a.B + a.slices - 1) / a.slices;
  const int64_t b0 = blockIdx.y * per, b1 = b0 + per < a.B ? b0 + per : a.B;
  const int64_t nsteps = b1 > b0 ? (b1 - b0 + kWgStep - 1) / kWgStep : 0;
  f32x16 acc[4] = {};
  const int s_mb = tid >> 6, s_lane = tid & 63;
  // A operand: fragment (mb, lane = hf*32 + r) of a sample is dY[b][32 mb + r][8 hf .. 8 hf + 7] — 32
  // contiguous bytes of fp32, the same bytes a pre-packed bf16 hi + lo pair would take; the split into
  // bf16 hi / lo happens on the way into LDS (a separate cin_pack_dy pass cost 21 us per layer).
  float4 rf[kWgStep][2] = {};
  const int s_c = 32 * s_mb + (s_lane & 31);
  auto load_a = [&](int64_t step) {
#pragma unroll
    for (int u = 0; u < kWgStep; ++u) {
      const int64_t bb = b0 + step * kWgStep + u;
      rf[u][0] = rf[u][1] = float4{0.f, 0.f, 0.f, 0.f};
      if (s_c < a.C && bb < b1) {
        const float* p = a.dY + (bb * a.C + s_c) * 16 + 8 * (s_lane >> 5);
        rf[u][0] = ld4(p);
        rf[u][1] = ld4(p + 4);
      }
    }
  };
  auto store_a = [&](int buf) {
    if (s_mb < a.MB) {
      unsigned char* base = wbuf + buf * 2 * SLAB;
#pragma unroll
      for (int u = 0; u < kWgStep; ++u) {
        const float v[8] = {rf[u][0].x, rf[u][0].y, rf[u][0].z, rf[u][0].w, rf[u][1].x, rf[u][1].y, rf[u][1].z, rf[u][1].w};
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          h[j] = static_cast<__bf16>(v[j]);
          if (SPLIT) l[j] = static_cast<__bf16>(v[j] - static_cast<float>(h[j]));
        }
        reinterpret_cast<bf16x8*>(base)[(u * 4 + s_mb) * 64 + s_lane] = h;
        if (SPLIT) reinterpret_cast<bf16x8*>(base + SLAB)[(u * 4 + s_mb) * 64 + s_lane] = l;
      }
    }
  };
  float4 hb[kWgStep][2] = {}, xb[kWgStep][2] = {};
  auto load_b = [&](int64_t step) {
#pragma unroll
    for (int u = 0; u < kWgStep; ++u) {
      const int64_t bb = b0 + step * kWgStep + u;
      hb[u][0] = hb[u][1] = xb[u][0] = xb[u][1] = float4{0.f, 0.f, 0.f, 0.f};
      if (kvalid && bb < b1) {
        const float* hp = a.hidden + bb * a.hidden_stride + h * 16 + 8 * hf;
        const float* xp = a.x0 + (bb * a.F + f) * 16 + 8 * hf;
        hb[u][0] = ld4(hp); hb[u][1] = ld4(hp + 4); xb[u][0] = ld4(xp); xb[u][1] = ld4(xp + 4);
      } else if (kbias && bb < b1) {
        hb[u][0] = hb[u][1] = xb[u][0] = xb[u][1] = float4{1.f, 1.f, 1.f, 1.f};
      }
    }
  };
  if (nsteps > 0) {
    load_a(0);
    store_a(0);
    load_b(0);
    if (nsteps > 1) load_a(1);
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see cin_dgrad_mfma
  auto steps = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int64_t step = 0; step < nsteps; ++step) {
      const int cur = static_cast<int>(step & 1);
      const unsigned char* base = wbuf + cur * 2 * SLAB;
      bf16x8 ah[kWgStep][4], al[kWgStep][4];
      if constexpr (FULL) {
#pragma unroll
        for (int u = 0; u < kWgStep; ++u) {
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            ah[u][mb] = reinterpret_cast<const bf16x8*>(base)[(u * 4 + mb) * 64 + lane];
            if (SPLIT) al[u][mb] = reinterpret_cast<const bf16x8*>(base + SLAB)[(u * 4 + mb) * 64 + lane];
          }
        }
      }
      if (step + 1 < nsteps) store_a(cur ^ 1);
      if (step + 2 < nsteps) load_a(step + 2);
      // B operands of this step from the registers loaded one step ago
      bf16x8 bh[kWgStep], bl[kWgStep];
#pragma unroll
      for (int u = 0; u < kWgStep; ++u) {
        const float z[8] = {hb[u][0].x * xb[u][0].x, hb[u][0].y * xb[u][0].y, hb[u][0].z * xb[u][0].z,
                            hb[u][0].w * xb[u][0].w, hb[u][1].x * xb[u][1].x, hb[u][1].y * xb[u][1].y,
                            hb[u][1].z * xb[u][1].z, hb[u][1].w * xb[u][1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bh[u][j] = static_cast<__bf16>(z[j]);
          if (SPLIT) bl[u][j] = static_cast<__bf16>(z[j] - static_cast<float>(bh[u][j]));
        }
      }
      if (step + 1 < nsteps) load_b(step + 1);
#pragma unroll
      for (int u = 0; u < kWgStep; ++u) {
        if constexpr (FULL) {
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u][mb], bh[u], acc[mb], 0, 0, 0);
          if (SPLIT) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u][mb], bl[u], acc[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u][mb], bh[u], acc[mb], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            if (mb < a.MB) {
              const bf16x8 a_h = reinterpret_cast<const bf16x8*>(base)[(u * 4 + mb) * 64 + lane];
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh[u], acc[mb], 0, 0, 0);
              if (SPLIT) {
                const bf16x8 a_l = reinterpret_cast<const bf16x8*>(base + SLAB)[(u * 4 + mb) * 64 + lane];
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl[u], acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh[u], acc[mb], 0, 0, 0);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  };
  if (a.MB == 4) steps(std::true_type{});
  else steps(std::false_type{});
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
__global__ __launch_bounds__(256) void cin_wgrad_reduce_mfma(const float* __restrict__ slabs, int slices,
                                                             int rows_pad, int cols_pad, int C, int H, int F,
                                                             int FP, float* __restrict__ dW, float* __restrict__ db) {
  const int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t K = static_cast<int64_t>(H) * F;
  const int64_t n = C * K;
  if (o >= n + (db ? C : 0)) return;
  const int c = o < n ? static_cast<int>(o / K) : static_cast<int>(o - n);
  const int k = static_cast<int>(o % K);
  const int kp = o < n ? (k / F) * FP + k % F : F;
  float acc = 0.f;
  for (int s = 0; s < slices; ++s)
    acc += slabs[(static_cast<int64_t>(s) * rows_pad + c) * cols_pad + kp];
  if (o < n) dW[o] += acc;
  else db[c] += acc;
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
#define DFM_CASE(GG) \
  if (D == 16 && FG == GG) return launch_dgrad<16, GG>(args, split, st);
  DFM_CASE(1) DFM_CASE(2) DFM_CASE(3) DFM_CASE(4) DFM_CASE(5)
#undef DFM_CASE
  return fail(DFM_ERR_UNSUPPORTED, "no MFMA CIN dgrad kernel for D=%d, F=%d", D, args.F);
}

// Batch slices of the weight-gradient GEMM: the grid is (column groups) x (slices) workgroups of
// equal work, two resident per CU, so the slice count is chosen to fill ONE round of 2 x 256
// workgroup slots (20 column groups x 32 slices = 640 workgroups ran a second round at 25 %
// occupancy: 184 us where 118 us of work was needed).
constexpr int kWgradMaxSlices = 64;
constexpr int kWgSlots = 2 * 256;
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

// true: cin_mfma_wgrad also produces the bias gradient (a free padding column exists)
bool cin_mfma_wgrad_has_bias(int F) { return F % 8 != 0; }

// dW += dY^T (hidden (x) x0); db += sum_{b,d} dY when `db` is given (cin_mfma_wgrad_has_bias); D must be 16
int cin_mfma_wgrad(const float* dY, const float* x0, const float* hidden, int64_t hidden_stride, int64_t B,
                   int F, int H, int C, float* dW, float* db, void* workspace, bool split, hipStream_t st) {
  DFM_REQUIRE(!db || cin_mfma_wgrad_has_bias(F), "no padding column for the bias gradient (F = %d)", F);
  const int MB = (C + 31) / 32, FP = ((F + 7) / 8) * 8;
  const int KT = (H * FP + 31) / 32;
  float* slabs = static_cast<float*>(workspace);
  const int slices = wgrad_slices(B, KT);
  CinWgradArgs a;
  a.dY = dY; a.C = C; a.x0 = x0; a.hidden = hidden; a.hidden_stride = hidden_stride; a.slabs = slabs;
  a.B = B; a.F = F; a.H = H; a.MB = MB; a.KT = KT; a.slices = slices; a.FP = FP;
  a.bias_col = db ? 1 : 0;
  const dim3 grid((KT + 3) / 4, slices);
  if (split) hipLaunchKernelGGL(cin_wgrad_mfma<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(cin_wgrad_mfma<false>, grid, dim3(256), 0, st, a);
  DFM_LAUNCH_CHECK();
  const int64_t n = static_cast<int64_t>(C) * H * F + (db ? C : 0);
  hipLaunchKernelGGL(cin_wgrad_reduce_mfma, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, slabs,
                     slices, MB * 32, KT * 32, C, H, F, FP, dW, db);
  DFM_LAUNCH_CHECK();
  return DFM_OK;
}

}  // namespace dfm
