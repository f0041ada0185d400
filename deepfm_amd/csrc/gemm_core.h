// Device-side core of the exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32), shared by the plain
// GEMM entry point (gemm_f32.hip) and the fused DNN-tower kernels (tower.hip).
//
//   acc(m, n) = sum_{k in [kb, ke)} A(m, k) * B(n, k)      for one 64 x 64 output tile
// Each operand is "K-contiguous" (element (r,k) at base[r*ld + k]) or "K-strided" (element
// (r,k) at base[k*ld + r]).  A workgroup of 8 waves owns the tile: waves 0-3 and 4-7 hold the
// same four 32 x 32 MFMA tiles but opposite halves of every 32-deep k slice (two independent
// MFMA chains per SIMD); the halves are added through LDS in a fixed order (half 0 + half 1)
// and the result lives in the accumulators of waves 0-3.  64 x 32 slices of A and B are staged
// through LDS (coalesced 16-byte global loads in either layout, padded rows: conflict-free
// fragment reads), double buffered, and consumed from register fragments (see mainloop).
#pragma once

#include <type_traits>

#include "common.h"

namespace dfm {
namespace gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDS_STRIDE = BK + 1;               // padded row: bank = (row + k) % 32
constexpr int kThreads = 512;                    // 8 waves: 4 output tiles x 2 k-halves

struct Smem {
  float a[2][BM * LDS_STRIDE];
  float b[2][BN * LDS_STRIDE];
};

// Position of this thread's accumulator registers inside the workgroup's 64 x 64 tile.
struct TilePos {
  int tile, khalf;   // which 32x32 tile (0..3), which k-half (0: the wave that ends up with the sum)
  int wm, wn;        // tile origin inside the block tile
  int r, hf;         // column inside the tile (lane & 31), row-half selector (lane >> 5)
  __device__ __forceinline__ TilePos() {
    const int lane = lane_id(), wave = wave_id_uniform();
    tile = wave & 3; khalf = wave >> 2;
    wm = (tile >> 1) * 32; wn = (tile & 1) * 32;
    r = lane & 31; hf = lane >> 5;
  }
  // accumulator register `reg` holds row (reg&3) + 8*(reg>>2) + 4*hf of the 32x32 tile, column r
  __device__ __forceinline__ int row(int reg) const { return wm + (reg & 3) + 8 * (reg >> 2) + 4 * hf; }
  __device__ __forceinline__ int col() const { return wn + r; }
};

// Stage a (64 rows x 32 k) slice of an operand into LDS as [row][k] (stride 33).
//   KC:      global element (r, k) at base[r*ld + k]  -> float4 along k
//   strided: global element (r, k) at base[k*ld + r]  -> float4 along r
// FAST (chosen on the host): every 16-byte piece is either entirely inside the operand or
// entirely outside (leading dimension and base 16-byte aligned, the vectorised extent a
// multiple of 4), so the load is branch-free: clamp the address, load, select zero.  Otherwise
// the guarded element-wise path runs (ragged shapes; correctness only).
template <bool KC, bool FAST>
__device__ __forceinline__ void load_slice(const float* __restrict__ base, int64_t ld, int r0, int rows, int k0,
                                           int kend, float4& v, bool& okv) {
  const int p = threadIdx.x;                      // 512 float4 pieces per slice, one per thread
  int r, k;
  if (KC) { r = r0 + (p >> 3); k = k0 + (p & 7) * 4; }        // 8 pieces per row, along k
  else    { k = k0 + (p >> 4); r = r0 + (p & 15) * 4; }       // 16 pieces per k, along r
  if (FAST) {
    // Rows past the operand's end are CLAMPED, not zeroed: row i of A (B) only ever reaches
    // accumulator row (column) i, which the epilogues never store, so whatever a clamped row holds
    // is harmless — and the steady state needs no predicate at all.  Only the k tail must read as
    // zero (it feeds every output); the select is applied at the LDS store, so the load itself is
    // not consumed before the MFMAs of the current slice.
    const int rc = KC ? (r < rows ? r : rows - 1) : (r < rows ? r : rows - 4);
    okv = k < kend;
    const int kc = okv ? k : 0;
    const int64_t off = KC ? static_cast<int64_t>(rc) * ld + kc : static_cast<int64_t>(kc) * ld + rc;
    v = ld4(base + off);
  } else {
    const int64_t off = KC ? static_cast<int64_t>(r) * ld + k : static_cast<int64_t>(k) * ld + r;
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = KC ? (r < rows && k + j < kend) : (k < kend && r + j < rows);
      e[j] = ok ? base[off + j] : 0.f;
    }
    v = make_float4(e[0], e[1], e[2], e[3]);
    okv = true;
  }
}

// This thread's piece of the slice at k = 0 (FAST operands): the steady state of the k loop loads
// `ptr + k * kstep` with no predicate (all rows clamped, every slice it touches entirely inside K).
template <bool KC>
struct SteadyPtr {
  const float* ptr;
  int64_t kstep;       // elements per unit of k
  __device__ __forceinline__ SteadyPtr(const float* base, int64_t ld, int r0, int rows) {
    const int p = threadIdx.x;
    int r, k;
    if (KC) { r = r0 + (p >> 3); k = (p & 7) * 4; }
    else    { k = p >> 4; r = r0 + (p & 15) * 4; }
    const int rc = KC ? (r < rows ? r : rows - 1) : (r < rows ? r : rows - 4);
    ptr = base + (KC ? static_cast<int64_t>(rc) * ld + k : static_cast<int64_t>(k) * ld + rc);
    kstep = KC ? 1 : ld;
  }
  __device__ __forceinline__ float4 load(int k0) const { return ld4(ptr + static_cast<int64_t>(k0) * kstep); }
};

template <bool KC, bool MASK = true>
__device__ __forceinline__ void store_slice(float* __restrict__ lds, const float4& vin, bool okv) {
  const int p = threadIdx.x;
  const float4 v = (!MASK || okv) ? vin : make_float4(0.f, 0.f, 0.f, 0.f);
  if (KC) {
    const int row = p >> 3, kq = (p & 7) * 4;
    float* d = lds + row * LDS_STRIDE + kq;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  } else {
    const int kk = p >> 4, rq = (p & 15) * 4;
    float* d = lds + rq * LDS_STRIDE + kk;
    d[0] = v.x; d[LDS_STRIDE] = v.y; d[2 * LDS_STRIDE] = v.z; d[3 * LDS_STRIDE] = v.w;
  }
}

// One of the four floats of store_slice (J = 0..3): the interleaved form of the pipeline step spreads the eight
// LDS writes of a slice over the gaps of the MFMA chain.
template <bool KC, bool MASK, int J>
__device__ __forceinline__ void store_piece(float* __restrict__ lds, const float4& vin, bool okv) {
  const int p = threadIdx.x;
  const float e = J == 0 ? vin.x : (J == 1 ? vin.y : (J == 2 ? vin.z : vin.w));
  const float v = (!MASK || okv) ? e : 0.f;
  if (KC) lds[(p >> 3) * LDS_STRIDE + (p & 7) * 4 + J] = v;
  else lds[((p & 15) * 4 + J) * LDS_STRIDE + (p >> 4)] = v;
}

#ifndef GEMM_INTERLEAVE
#define GEMM_INTERLEAVE 1
#endif
#ifndef GEMM_INTERLEAVE_TAIL
#define GEMM_INTERLEAVE_TAIL 0   // 1: the checked steps (k tails, every step of a contraction shorter than 10 slices) use the gap schedule too — measured +-0 (d-input product of layer 1: 23.2 us either way; headline 0.2046 / 0.2045 ms)
#endif
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0    // timing-only builds (tools/build_variant.sh), steady state of the fp32 loop: 1 no MFMAs, 2 no global loads, 4 no LDS stores, 8 no fragment reads, 16 no barrier
#endif

// The whole k loop of one output tile.  On return `acc` is complete in the waves with
// pos.khalf == 0; every wave has passed the same barriers and `sm` is free for the epilogue.
//
// Software pipeline, three stages deep.  The fp32 MFMA chain (8 dependent v_mfma_f32_32x32x2 per
// wave and slice, 2 waves per SIMD: 1024 cycles per slice) is the floor; everything else has to
// hide under it, so while slice s is multiplied from REGISTER fragments,
//   - the fragments of slice s+1 are read from LDS buffer (s+1)%2 into a second register set,
//   - slice s+2 moves from its global-prefetch registers into LDS buffer s%2 (whose fragments were
//     consumed one barrier ago),
//   - slice s+6 is requested from memory into the register slot slice s+2 just left
// (kPrefetch = 4 slots, one float4 per operand per thread and slice).  One barrier per slice.
constexpr int kPrefetch = 4;
constexpr int kFrag = BK / 4;        // fragment values per operand per lane per slice (this wave's k-half)

__device__ __forceinline__ void read_frags(const Smem& sm, int buf, const TilePos& pos, float (&fa)[kFrag],
                                           float (&fb)[kFrag]) {
  const float* pa = sm.a[buf] + (pos.wm + pos.r) * LDS_STRIDE + pos.hf + pos.khalf * (BK / 2);
  const float* pb = sm.b[buf] + (pos.wn + pos.r) * LDS_STRIDE + pos.hf + pos.khalf * (BK / 2);
#pragma unroll
  for (int i = 0; i < kFrag; ++i) { fa[i] = pa[2 * i]; fb[i] = pb[2 * i]; }
}

// One slice of the pipeline (slice index s = J mod 4; kc = its k offset).  CHECK = false is the
// steady state (slices s+1, s+2 and s+6 all exist): no branches, so the compiler counts the
// outstanding loads exactly instead of waiting for all of them.
template <bool A_KC, bool B_KC, bool A_FAST, bool B_FAST, int J, bool CHECK>
__device__ __forceinline__ void pipe_step(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                          int64_t ldb, int M, int N, int m0, int n0, int kc, int ke, Smem& sm,
                                          const TilePos& pos, const SteadyPtr<A_KC>& spa, const SteadyPtr<B_KC>& spb,
                                          float4 (&va)[kPrefetch], float4 (&vb)[kPrefetch], bool (&oka)[kPrefetch],
                                          bool (&okb)[kPrefetch], float (&fa)[kFrag], float (&fb)[kFrag], f32x16& acc) {
  constexpr int b1 = (J + 1) & 1, b2 = J & 1, slot = (J + 2) % kPrefetch;
  // steady state on FAST operands: every slice touched is entirely inside K and rows are clamped,
  // so neither the loads nor the LDS stores carry a predicate (about 60 fewer VALU instructions per
  // wave and slice — address arithmetic and zero-selects — beside a 512-cycle MFMA chain).
  // (Also dropping the predicates in the CHECK steps of whole-slice K ranges measured SLOWER:
  // 0.289 vs 0.265 ms/step — four more inlined variants of the loop body per kernel.)
  constexpr bool kBareA = !CHECK && A_FAST, kBareB = !CHECK && B_FAST;
  if (CHECK && kc >= ke) return;
  float na[kFrag], nb[kFrag];
  if constexpr (GEMM_INTERLEAVE && (!CHECK || GEMM_INTERLEAVE_TAIL)) {
    // Steady state, interleaved (round 3): the 8 MFMAs of a slice are one dependent chain — the wave sits at each
    // of them for its 64-cycle pass — and everything else of the step is independent of it (the NEXT slice's
    // fragment reads, the slice after's LDS writes, a far slice's global loads).  Issued in front of the chain they
    // kept the matrix pipe idle for the whole LDS phase, in BOTH waves of a SIMD at once (the per-slice barrier
    // puts them in lock-step: ~1670 cycles per slice against 1024 of MFMA work); issued in the chain's gaps — two
    // fragment reads and one LDS write behind every MFMA, the global loads behind the last two — they cost nothing.
    const float* pa = sm.a[b1] + (pos.wm + pos.r) * LDS_STRIDE + pos.hf + pos.khalf * (BK / 2);
    const float* pb = sm.b[b1] + (pos.wn + pos.r) * LDS_STRIDE + pos.hf + pos.khalf * (BK / 2);
    static_assert(kFrag == 8, "the gap schedule below is written for 8 MFMAs per slice");
#define DFM_GAP(I, STORE)                                                                   \
    if (!(GEMM_ABLATE & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[I], fb[I], acc, 0, 0, 0); \
    else acc[I] += fa[I] + fb[I];                                                            \
    if (!(GEMM_ABLATE & 8)) { na[I] = pa[2 * (I)]; nb[I] = pb[2 * (I)]; } else { na[I] = fa[I]; nb[I] = fb[I]; } \
    if (!(GEMM_ABLATE & 4)) { STORE; }                                                      \
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    DFM_GAP(0, (store_piece<A_KC, !kBareA, 0>(sm.a[b2], va[slot], oka[slot])))
    DFM_GAP(1, (store_piece<A_KC, !kBareA, 1>(sm.a[b2], va[slot], oka[slot])))
    DFM_GAP(2, (store_piece<A_KC, !kBareA, 2>(sm.a[b2], va[slot], oka[slot])))
    DFM_GAP(3, (store_piece<A_KC, !kBareA, 3>(sm.a[b2], va[slot], oka[slot])))
    DFM_GAP(4, (store_piece<B_KC, !kBareB, 0>(sm.b[b2], vb[slot], okb[slot])))
    DFM_GAP(5, (store_piece<B_KC, !kBareB, 1>(sm.b[b2], vb[slot], okb[slot])))
    DFM_GAP(6, (store_piece<B_KC, !kBareB, 2>(sm.b[b2], vb[slot], okb[slot])))
    DFM_GAP(7, (store_piece<B_KC, !kBareB, 3>(sm.b[b2], vb[slot], okb[slot])))
#undef DFM_GAP
    // the slot is free now: slice s + 6 on its way
    if (!(GEMM_ABLATE & 2)) {
    if (kBareA) { va[slot] = spa.load(kc + (kPrefetch + 2) * BK); oka[slot] = true; }
    else load_slice<A_KC, A_FAST>(A, lda, m0, M, kc + (kPrefetch + 2) * BK, ke, va[slot], oka[slot]);
    if (kBareB) { vb[slot] = spb.load(kc + (kPrefetch + 2) * BK); okb[slot] = true; }
    else load_slice<B_KC, B_FAST>(B, ldb, n0, N, kc + (kPrefetch + 2) * BK, ke, vb[slot], okb[slot]);
    }
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
  for (int i = 0; i < kFrag; ++i) na[i] = nb[i] = 0.f;
  if (!CHECK || kc + BK < ke) read_frags(sm, b1, pos, na, nb);
  if (!CHECK || kc + 2 * BK < ke) {
    store_slice<A_KC, !kBareA>(sm.a[b2], va[slot], oka[slot]);
    store_slice<B_KC, !kBareB>(sm.b[b2], vb[slot], okb[slot]);
  }
  if (!CHECK || kc + (kPrefetch + 2) * BK < ke) {
    if (kBareA) { va[slot] = spa.load(kc + (kPrefetch + 2) * BK); oka[slot] = true; }
    else load_slice<A_KC, A_FAST>(A, lda, m0, M, kc + (kPrefetch + 2) * BK, ke, va[slot], oka[slot]);
    if (kBareB) { vb[slot] = spb.load(kc + (kPrefetch + 2) * BK); okb[slot] = true; }
    else load_slice<B_KC, B_FAST>(B, ldb, n0, N, kc + (kPrefetch + 2) * BK, ke, vb[slot], okb[slot]);
  }
  // keep the issue order [LDS reads, LDS writes, global loads] -> [MFMA chain]: left alone, the
  // scheduler sinks each fragment read next to its MFMA and the chain stalls on LDS latency again
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < kFrag; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  }
  if (!(GEMM_ABLATE & 16) || CHECK) __syncthreads();
#pragma unroll
  for (int i = 0; i < kFrag; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
}

template <bool A_KC, bool B_KC, bool A_FAST, bool B_FAST>
__device__ __forceinline__ void mainloop(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                         int64_t ldb, int M, int N, int m0, int n0, int kb, int ke, Smem& sm,
                                         const TilePos& pos, f32x16& acc) {
  static_assert(kPrefetch == 4, "the slot schedule below is written out for 4 slices in flight");
  const int lane = lane_id();
  float4 va[kPrefetch], vb[kPrefetch];
  bool oka[kPrefetch], okb[kPrefetch];
#pragma unroll
  for (int j = 0; j < kPrefetch; ++j) {
    oka[j] = okb[j] = false;
    va[j] = vb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kb + j * BK < ke) {
      load_slice<A_KC, A_FAST>(A, lda, m0, M, kb + j * BK, ke, va[j], oka[j]);
      load_slice<B_KC, B_FAST>(B, ldb, n0, N, kb + j * BK, ke, vb[j], okb[j]);
    }
  }
  // slices 0 and 1 -> LDS buffers 0 and 1; their slots take slices 4 and 5
  store_slice<A_KC>(sm.a[0], va[0], oka[0]);
  store_slice<B_KC>(sm.b[0], vb[0], okb[0]);
  store_slice<A_KC>(sm.a[1], va[1], oka[1]);
  store_slice<B_KC>(sm.b[1], vb[1], okb[1]);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (kb + (kPrefetch + j) * BK < ke) {
      load_slice<A_KC, A_FAST>(A, lda, m0, M, kb + (kPrefetch + j) * BK, ke, va[j], oka[j]);
      load_slice<B_KC, B_FAST>(B, ldb, n0, N, kb + (kPrefetch + j) * BK, ke, vb[j], okb[j]);
    }
  }
  __syncthreads();
  float fa[kFrag], fb[kFrag];
  read_frags(sm, 0, pos, fa, fb);
  __syncthreads();               // slice 0's fragments are in registers: buffer 0 may be overwritten
  const SteadyPtr<A_KC> spa(A, lda, m0, M);
  const SteadyPtr<B_KC> spb(B, ldb, n0, N);
#define DFM_PIPE(J, CHECK)                                                                                       \
  pipe_step<A_KC, B_KC, A_FAST, B_FAST, J, CHECK>(A, lda, B, ldb, M, N, m0, n0, k0 + (J) * BK, ke, sm, pos, spa, \
                                                  spb, va, vb, oka, okb, fa, fb, acc)
  // slice s: LDS buffer s % 2, global-prefetch slot s % 4
  int k0 = kb;
  // Tell the compiler that nothing older than the four prefetch slots is in flight on loop entry.
  // Without this wait the loop header inherits "possibly pending" loads from the branchy prologue
  // and the steady state waits for vmcnt(0) once per four slices (the prefetch ring drains);
  // with it every slice waits for vmcnt(7) / vmcnt(6), i.e. only for its own slot.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  for (; k0 + (kPrefetch + 6) * BK <= ke; k0 += kPrefetch * BK) {  // slices up to s+9 are FULL slices: no checks
    DFM_PIPE(0, false); DFM_PIPE(1, false); DFM_PIPE(2, false); DFM_PIPE(3, false);
  }
  for (; k0 < ke; k0 += kPrefetch * BK) {
    DFM_PIPE(0, true); DFM_PIPE(1, true); DFM_PIPE(2, true); DFM_PIPE(3, true);
  }
#undef DFM_PIPE
  // combine the two k-halves: waves 4-7 park their tile in LDS (the staging buffers are free after
  // the loop's last barrier), waves 0-3 add it (fixed order: half 0 + half 1)
  float* park = &sm.a[0][0];                   // 4 tiles x 16 regs x 64 lanes = 16 KiB <= sizeof(sm.a)
  if (pos.khalf == 1) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) park[(pos.tile * 16 + reg) * 64 + lane] = acc[reg];
  }
  __syncthreads();
  if (pos.khalf == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] += park[(pos.tile * 16 + reg) * 64 + lane];
  }
  __syncthreads();
}

// =====================================================================================
// The same 64 x 64 tile product on the bf16 matrix pipe, bf16 x 3 split (round 3; the tower's BACKWARD only):
//   a = ah + al, b = bh + bl (bf16 each, 2^-17 relative),  a b ~ ah bh + ah bl + al bh,  fp32 accumulate
// — the CIN's arithmetic.  v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, the split issues 3x: the
// matrix-pipe time of a tile falls 5.3x.  Why only the backward: its inputs (d z, activations, weights) are
// fixed by the exact-fp32 forward, so the ReLU / dropout masks are fp32's; bf16 x 3 in the FORWARD moves every
// pre-activation by ~1e-5 and flips ~27 ReLUs per step at B = 4096, which breaks the parity bar on most of the
// first layer's gradient elements (tools/emulate_tower_bf16x3.py: forward + backward 70-97 % outside, backward
// only 0 % outside, max 1.1e-5 of scale against fp32's 1.2e-5).
//
// Accumulator layout, tile decomposition (4 tiles x 2 k-halves) and the k-half combine are the fp32 loop's, so
// every epilogue is shared.  64-deep slices: a k-half takes two 16-deep k-steps per slice (6 MFMAs per wave),
// operands are split to bf16 hi / lo on their way into LDS, fragment-shaped:
//   plane[ks (4)][h (2)][row (64)][8 bf16]   (lane (row r, half h) of k-step ks reads 16 contiguous bytes)
// K-contiguous operands: a thread converts 8 consecutive k of one row -> one ds_write_b128 per plane;
// K-strided ones: a thread loads two consecutive k for 4 rows and writes 4 packed (k, k+1) pairs per plane.
// FAST operands only (16-byte pieces all-or-nothing); the k tail reads as zero, rows are clamped.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int BKX = 64;
constexpr int kPlane = 4 * 2 * 64 * 8;          // bf16 elements of one plane (hi or lo) of one operand: 8 KB
struct SmemX3 {
  __bf16 a[2][2][kPlane];                       // [buffer][hi, lo]
  __bf16 b[2][2][kPlane];
};
static_assert(sizeof(SmemX3) == 64 * 1024, "two operands x two buffers x (hi, lo) x 8 KB");

// this thread's two 16-byte pieces of a 64 x 64 operand slice
template <bool KC>
struct PieceX3 {
  const float* ptr;       // this thread's piece 0 at absolute k = 0 (row(s) clamped)
  int64_t kstep;          // elements per unit of k
  int k_off;              // k of piece 0 inside a slice (piece 1: KC: + 4, strided: + 1)
  int lds0;               // KC: element offset of the 8 bf16 this thread writes; strided: of its first packed pair
  __device__ __forceinline__ PieceX3(const float* base, int64_t ld, int r0, int rows) {
    const int p = threadIdx.x;
    if (KC) {             // row p >> 3, k = 8 (p & 7) .. + 7
      const int r = r0 + (p >> 3), k8 = p & 7;
      const int rc = r < rows ? r : rows - 1;
      ptr = base + static_cast<int64_t>(rc) * ld;
      kstep = 1; k_off = 8 * k8;
      lds0 = (((k8 >> 1) * 2 + (k8 & 1)) * 64 + (p >> 3)) * 8;
    } else {              // k pair 2 (p >> 4), 2 (p >> 4) + 1; rows 4 (p & 15) .. + 3
      const int kq = p >> 4, r = r0 + (p & 15) * 4;
      const int rc = r < rows ? r : rows - 4;
      ptr = base + rc;
      kstep = ld; k_off = 2 * kq;
      const int k = 2 * kq;                       // k-step k >> 4, half (k >> 3) & 1, element k & 7 (even)
      lds0 = ((((k >> 4) * 2 + ((k >> 3) & 1)) * 64 + (p & 15) * 4) * 8) + (k & 7);
    }
  }
  // the slice that starts at absolute k0; pieces at k >= ke read k = 0 (always inside the operand) and are zeroed
  __device__ __forceinline__ void load(int k0, int ke, float4& v0, float4& v1, bool& ok0, bool& ok1) const {
    const int ka = k0 + k_off, kb = ka + (KC ? 4 : 1);
    ok0 = ka < ke; ok1 = kb < ke;
    v0 = ld4(ptr + static_cast<int64_t>(ok0 ? ka : 0) * kstep);
    v1 = ld4(ptr + static_cast<int64_t>(ok1 ? kb : 0) * kstep);
  }
};

__device__ __forceinline__ void split2(float v, __bf16& hi, __bf16& lo) {
  hi = static_cast<__bf16>(v);
  lo = static_cast<__bf16>(v - static_cast<float>(hi));
}

template <bool KC>
__device__ __forceinline__ void store_x3(__bf16* __restrict__ hi, __bf16* __restrict__ lo, const PieceX3<KC>& pc,
                                         const float4& a0, const float4& a1, bool ok0, bool ok1) {
  const float z = 0.f;
  const float v[8] = {ok0 ? a0.x : z, ok0 ? a0.y : z, ok0 ? a0.z : z, ok0 ? a0.w : z,
                      ok1 ? a1.x : z, ok1 ? a1.y : z, ok1 ? a1.z : z, ok1 ? a1.w : z};
  if (KC) {
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 hh, ll; split2(v[j], hh, ll); h[j] = hh; l[j] = ll; }
    *reinterpret_cast<bf16x8*>(hi + pc.lds0) = h;
    *reinterpret_cast<bf16x8*>(lo + pc.lds0) = l;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {              // row r: (k, k + 1) = (v[r], v[4 + r])
      bf16x2 h, l;
      __bf16 hh, ll;
      split2(v[r], hh, ll); h[0] = hh; l[0] = ll;
      split2(v[4 + r], hh, ll); h[1] = hh; l[1] = ll;
      *reinterpret_cast<bf16x2*>(hi + pc.lds0 + r * 8) = h;
      *reinterpret_cast<bf16x2*>(lo + pc.lds0 + r * 8) = l;
    }
  }
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void mainloop_x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                            int64_t ldb, int M, int N, int m0, int n0, int kb, int ke, SmemX3& sm,
                                            const TilePos& pos, f32x16& acc) {
  const int lane = lane_id();
  const PieceX3<A_KC> pa(A, lda, m0, M);
  const PieceX3<B_KC> pb(B, ldb, n0, N);
  const int nsl = (ke - kb + BKX - 1) / BKX;
  // two register sets of prefetched slices, alternating: step s converts slice s + 1 (requested two steps ago)
  // into the other LDS buffer (its readers left it at the last barrier) and requests slice s + 3 into the registers
  // that just emptied
  // (compile-time set / buffer numbers — the loop is unrolled by two: a run-time index would put the sets in scratch)
  struct Set { float4 a0, a1, b0, b1; bool oa0, oa1, ob0, ob1; };
  Set st0, st1;
  auto request = [&](int sl, Set& t) {
    pa.load(kb + sl * BKX, ke, t.a0, t.a1, t.oa0, t.oa1);
    pb.load(kb + sl * BKX, ke, t.b0, t.b1, t.ob0, t.ob1);
  };
  auto convert = [&](int buf, const Set& t) {
    store_x3<A_KC>(sm.a[buf][0], sm.a[buf][1], pa, t.a0, t.a1, t.oa0, t.oa1);
    store_x3<B_KC>(sm.b[buf][0], sm.b[buf][1], pb, t.b0, t.b1, t.ob0, t.ob1);
  };
  st1 = Set{};
  request(0, st0);
  if (nsl > 1) request(1, st1);
  convert(0, st0);
  if (nsl > 2) request(2, st0);
  __syncthreads();
  // this wave's fragments: k-steps 2 khalf and 2 khalf + 1 of the slice, rows wm + r (A) / wn + r (B), half hf
  const int fa = (((2 * pos.khalf) * 2 + pos.hf) * 64 + pos.wm + pos.r) * 8;
  const int fb = (((2 * pos.khalf) * 2 + pos.hf) * 64 + pos.wn + pos.r) * 8;
  constexpr int kStepElems = 2 * 64 * 8;        // one k-step further
  // step s (parity P = s & 1): multiply LDS buffer P; the OTHER register set holds slice s + 1 -> buffer P ^ 1, then
  // takes slice s + 3
  auto step = [&](int s, auto parity, Set& other) {
    constexpr int P = decltype(parity)::value;
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ah[q] = *reinterpret_cast<const bf16x8*>(sm.a[P][0] + fa + q * kStepElems);
      al[q] = *reinterpret_cast<const bf16x8*>(sm.a[P][1] + fa + q * kStepElems);
      bh[q] = *reinterpret_cast<const bf16x8*>(sm.b[P][0] + fb + q * kStepElems);
      bl[q] = *reinterpret_cast<const bf16x8*>(sm.b[P][1] + fb + q * kStepElems);
    }
    if (s + 1 < nsl) convert(P ^ 1, other);
    if (s + 3 < nsl) request(s + 3, other);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], bh[q], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q], bl[q], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[q], bh[q], acc, 0, 0, 0);
    }
    __syncthreads();
  };
  int s = 0;
  for (; s + 1 < nsl; s += 2) {
    step(s, std::integral_constant<int, 0>{}, st1);
    step(s + 1, std::integral_constant<int, 1>{}, st0);
  }
  if (s < nsl) step(s, std::integral_constant<int, 0>{}, st1);
  // combine the two k-halves as the fp32 loop does (fixed order: half 0 + half 1)
  float* park = reinterpret_cast<float*>(&sm.a[0][0][0]);      // 16 KiB
  if (pos.khalf == 1) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) park[(pos.tile * 16 + reg) * 64 + lane] = acc[reg];
  }
  __syncthreads();
  if (pos.khalf == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] += park[(pos.tile * 16 + reg) * 64 + lane];
  }
  __syncthreads();
}

// Workgroups are dealt to the 8 XCDs round-robin in dispatch order, and each XCD has its own L2.
// Tiles that share an operand slab (the n-tiles of one m-tile, the tiles of one batch split) are
// adjacent in the LOGICAL order; this maps dispatch index -> logical index so that adjacent logical
// tiles run on the same XCD and the shared slab is fetched into one L2 instead of eight.
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_logical_index(int w, int total) {
  const int per = total / kXcds;
  if (w >= per * kXcds) return w;            // ragged tail: identity
  return (w % kXcds) * per + w / kXcds;
}

// The same for a group of workgroups that starts at dispatch index `first` (its members sit on XCD
// (first + w) % 8): one launch that carries two kinds of work (d weight tiles, then d input tiles) must
// spread EACH kind over all eight XCDs.  Mapping the whole grid at once put the first, long-running
// kind on the first XCDs only: at out 256 x in 2496 the fused launch took 326 us against 65 + 69 us
// for its two halves launched separately (tools/time_linear_bwd.py).
__device__ __forceinline__ int xcd_logical_index_from(int w, int total, int first) {
  const int per = total / kXcds;
  if (w >= per * kXcds) return w;
  return ((w + first) % kXcds) * per + w / kXcds;
}

// Host side: branch-free tile loads need all-or-nothing 16-byte pieces (see load_slice).
static inline bool operand_fast(const float* p, int64_t ld, bool kc, int rows, int kdim) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0 && (kc ? kdim % 4 == 0 : rows % 4 == 0);
}

}  // namespace gemm
}  // namespace dfm
