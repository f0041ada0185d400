// fp32-faithful GEMM tiles on the bf16 matrix pipe: the bf16 x 6 product (round 3; dfm_tower_set_mode(2)).
//
// An fp32 value splits EXACTLY into three bf16 values, x = h + m + l (8 + 8 + 8 significant bits, by truncation:
// h = x with its low 16 bits cleared, m the same of x - h, l = x - h - m, all three subtractions exact).  Of the nine
// partial products of a * b the six largest are kept,
//   a b ~ ah bh + (ah bm + am bh) + (am bm + ah bl + al bh),      dropped: am bl + al bm + al bl  <= 2^-23 |a b|,
// each an exact fp32 value accumulated in fp32 by v_mfma_f32_32x32x16_bf16 — the error of the product is that of
// ONE fp32 rounding, like the fp32 MFMA's, so the forward's ReLU / dropout masks stay fp32's (the bf16 x 3 split
// of mode 1 moves pre-activations by ~1e-5 and flips ~27 ReLUs per step; tools/emulate_tower_bf16x3.py).
// Six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16-deep k-step: 2.7 x less matrix-pipe time.
//
// What makes that pay: the split costs ~5 vector instructions per element, more than the MFMAs it feeds if every
// workgroup that stages an element splits it again (a 64 x 64 tile stages each element of an activation for 7 column
// tiles, each weight for 64 row tiles).  So operands are split ONCE, by the kernel that produces them (the
// BatchNorm apply kernels, a weight-split launch), into "planes":
//
//   planes of an operand with R rows and a contraction extent C:   bf16 [3 (h, m, l)][G][Rp][8]
//       G = 8 ceil(C / 64) groups of 8 consecutive contraction indices, Rp = 64 ceil(R / 64) rows,
//       element (row r, contraction c) of plane q at ((q G + c / 8) Rp + r) 8 + c % 8;  pads are zero (the buffer is
//       zero-filled once and producers write valid elements only), so the tile loop needs no predicate at all.
//
// A 64-row x 64-deep slice of a plane is 8 runs of 1 KB; a thread moves one 16-byte piece per plane (group p >> 6,
// row p & 63) and the LDS image has the SAME order, so the MFMA fragment of lane (row r, k-group hf) is one
// conflict-free ds_read_b128.  The same matrix is contracted over its columns by one product and over its rows by
// another (x in z = x W^T and in dW = dz^T x): producers write both ("role F": contraction = columns; "role S":
// contraction = rows).  The first layer's input (the embeddings, produced by the gather) stays fp32 and is split by
// the tile loop itself (OP_F32_KC / OP_F32_STRIDED) — one operand of two, under the MFMA time.
//
// Tile decomposition, accumulator layout and the k-half combine are gemm_core.h's (TilePos), so epilogues are shared.
#pragma once

#include "gemm_core.h"

namespace dfm {
namespace gemm {

#ifndef X6_ABLATE
#define X6_ABLATE 0      // timing-only builds (tools/build_variant.sh): 1 no MFMAs, 2 no loads, 4 no LDS writes, 8 no barrier in the loop
#endif
constexpr int X6_BK = 64;
constexpr int kX6PlaneElems = 8 * 64 * 8;                    // bf16 elements of one plane of one slice: 8 KB
constexpr int kX6OperandElems = 3 * kX6PlaneElems;           // 24 KB
constexpr int kX6SmemBytes = 2 * 2 * kX6OperandElems * 2;    // [buffer][operand][plane]: 96 KB (dynamic LDS)

__host__ __device__ inline int planes_rows(int64_t rows) { return static_cast<int>((rows + 63) / 64 * 64); }
__host__ __device__ inline int planes_groups(int64_t contraction) { return static_cast<int>((contraction + 63) / 64 * 8); }
__host__ __device__ inline int64_t planes_plane_elems(int64_t rows, int64_t contraction) {
  return static_cast<int64_t>(planes_groups(contraction)) * planes_rows(rows) * 8;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte piece (8 bf16)

struct Planes {            // device view of one plane set
  __bf16* base;            // plane h; m and l follow at +plane, +2 plane
  int64_t plane;           // elements per plane
  int rows;                // Rp
};
static inline Planes make_planes(void* p, int64_t rows, int64_t contraction) {
  Planes P;
  P.base = static_cast<__bf16*>(p);
  P.plane = planes_plane_elems(rows, contraction);
  P.rows = planes_rows(rows);
  return P;
}

// x = h + m + l exactly; returns the three as fp32 bit patterns whose HIGH halves are the bf16 values
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x);
  const float r1 = x - __uint_as_float(h & 0xffff0000u);
  m = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __float_as_uint(r2);             // <= 8 significant bits: its low half is zero
}
// high halves of (lo, hi) -> one dword (element 2d in the low half)
__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// 8 consecutive contraction values of one row -> the row's 16-byte piece of each plane
__device__ __forceinline__ void split_piece(const float (&v)[8], u32x4& ph, u32x4& pm, u32x4& pl) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
  ph = u32x4{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])};
  pm = u32x4{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])};
  pl = u32x4{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])};
}

// ---- producers: a workgroup holds a 32-row x 64-column fp32 tile in LDS (row stride kTileStride) and writes the
// pieces of both roles that lie inside the matrix (rows R % 8 == 0 for role S, columns C % 8 == 0 for role F).
constexpr int kTileRows = 32, kTileCols = 64, kTileStride = 68;     // 272-byte rows: conflict-free 16-byte row reads
__device__ __forceinline__ void emit_planes_from_tile(const float* __restrict__ tile, int r0, int c0, int R, int C,
                                                      const Planes& F, const Planes& S) {
  const int t = threadIdx.x;                    // 256 threads
  if (F.base) {                                 // role F: piece (row m, column group kg)
    const int m = t & 31, kg = t >> 5;
    if (r0 + m < R && c0 + kg * 8 < C) {
      const float4 a = *reinterpret_cast<const float4*>(tile + m * kTileStride + kg * 8);
      const float4 b = *reinterpret_cast<const float4*>(tile + m * kTileStride + kg * 8 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      u32x4 ph, pm, pl;
      split_piece(v, ph, pm, pl);
      u32x4* d = reinterpret_cast<u32x4*>(F.base) + static_cast<int64_t>((c0 >> 3) + kg) * F.rows + r0 + m;
      const int64_t pp = F.plane >> 3;
      d[0] = ph; d[pp] = pm; d[2 * pp] = pl;
    }
  }
  if (S.base) {                                 // role S: piece (column f, row group mg)
    const int f = t & 63, mg = t >> 6;
    if (c0 + f < C && r0 + mg * 8 < R) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(mg * 8 + j) * kTileStride + f];
      u32x4 ph, pm, pl;
      split_piece(v, ph, pm, pl);
      u32x4* d = reinterpret_cast<u32x4*>(S.base) + static_cast<int64_t>((r0 >> 3) + mg) * S.rows + c0 + f;
      const int64_t pp = S.plane >> 3;
      d[0] = ph; d[pp] = pm; d[2 * pp] = pl;
    }
  }
}

// ---- operands of the tile loop: request(k0, ke) asks memory for this thread's share of the slice that starts at
// absolute contraction index k0; commit(lds) puts it (split, if it was fp32) into the operand's LDS image.
enum { OP_PLANES = 0, OP_F32_KC = 1, OP_F32_STRIDED = 2 };
template <int KIND> struct X6Operand;

template <>
struct X6Operand<OP_PLANES> {
  const u32x4* ptr;      // this thread's piece of group 0, plane h
  int64_t gstep, pstep;  // 16-byte units per group, per plane
  struct Regs { u32x4 v0, v1, v2; };
  __device__ __forceinline__ X6Operand(const Planes& P, int r0) {
    const int p = threadIdx.x;
    ptr = reinterpret_cast<const u32x4*>(P.base) + static_cast<int64_t>(p >> 6) * P.rows + r0 + (p & 63);
    gstep = P.rows; pstep = P.plane >> 3;
  }
  __device__ __forceinline__ void request(int k0, int, Regs& r) const {
    const u32x4* q = ptr + static_cast<int64_t>(k0 >> 3) * gstep;
    r.v0 = q[0]; r.v1 = q[pstep]; r.v2 = q[2 * pstep];
  }
  __device__ __forceinline__ void pieces(const Regs& r, u32x4& p0, u32x4& p1, u32x4& p2) const { p0 = r.v0; p1 = r.v1; p2 = r.v2; }
  __device__ __forceinline__ int lds_index() const { return threadIdx.x; }
};

template <>
struct X6Operand<OP_F32_KC> {          // element (row r, contraction k) at base[r * ld + k]; K % 8 == 0, 16-byte aligned rows
  const float* ptr;
  int k_off, lds_off;
  struct Regs { float4 v0, v1; bool ok; };
  __device__ __forceinline__ X6Operand(const float* base, int64_t ld, int r0, int rows) {
    // a wave covers 8 rows x 64 k (256 B of each row) whichever lane takes which piece; 8 CONSECUTIVE lanes take the
    // same k group of 8 consecutive rows, so that their ds_write_b128s fill 128 contiguous bytes (with lane -> (row
    // p >> 3, group p & 7) the eight stores of a group hit the same four banks: 17.7 us against 12.4 for the first
    // layer's forward)
    const int p = threadIdx.x;
    const int row = (p & 7) + 8 * (p >> 6), grp = (p >> 3) & 7;
    const int r = r0 + row;
    ptr = base + static_cast<int64_t>(r < rows ? r : rows - 1) * ld;
    k_off = 8 * grp;
    lds_off = grp * 64 + row;
  }
  __device__ __forceinline__ void request(int k0, int ke, Regs& r) const {
    const int ka = k0 + k_off;
    r.ok = ka < ke;
    const float* q = ptr + (r.ok ? ka : 0);
    r.v0 = ld4(q); r.v1 = ld4(q + 4);
  }
  __device__ __forceinline__ void pieces(const Regs& r, u32x4& p0, u32x4& p1, u32x4& p2) const {
    const float z = 0.f;
    const float v[8] = {r.ok ? r.v0.x : z, r.ok ? r.v0.y : z, r.ok ? r.v0.z : z, r.ok ? r.v0.w : z,
                        r.ok ? r.v1.x : z, r.ok ? r.v1.y : z, r.ok ? r.v1.z : z, r.ok ? r.v1.w : z};
    split_piece(v, p0, p1, p2);
  }
  __device__ __forceinline__ int lds_index() const { return lds_off; }
};

template <>
struct X6Operand<OP_F32_STRIDED> {     // element (row f, contraction m) at base[m * ld + f]; extent of m % 8 == 0
  const float* ptr;
  int64_t ld;
  int m_off;
  struct Regs { float v[8]; bool ok; };
  __device__ __forceinline__ X6Operand(const float* base, int64_t ld_, int r0, int rows) {
    const int p = threadIdx.x, f = r0 + (p & 63);
    ptr = base + (f < rows ? f : rows - 1);
    ld = ld_;
    m_off = 8 * (p >> 6);
  }
  __device__ __forceinline__ void request(int k0, int ke, Regs& r) const {
    const int ma = k0 + m_off;
    r.ok = ma < ke;
    const float* q = ptr + static_cast<int64_t>(r.ok ? ma : 0) * ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = q[j * ld];
  }
  __device__ __forceinline__ void pieces(const Regs& r, u32x4& p0, u32x4& p1, u32x4& p2) const {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = r.ok ? r.v[j] : 0.f;
    split_piece(v, p0, p1, p2);
  }
  __device__ __forceinline__ int lds_index() const { return threadIdx.x; }
};

// The whole contraction [kb, ke) of one 64 x 64 output tile (planes operands: kb and ke - kb multiples of 64 or the
// pads beyond ke zero).  `smem` = kX6SmemBytes of dynamic LDS.  On return acc is complete in the waves with
// pos.khalf == 0 and smem is free.
template <int AK, int BK_>
__device__ __forceinline__ void mainloop_x6(const X6Operand<AK>& oa, const X6Operand<BK_>& ob, int kb, int ke,
                                            __bf16* __restrict__ smem, const TilePos& pos, f32x16& acc) {
  const int lane = lane_id();
  const int nsl = (ke - kb + X6_BK - 1) / X6_BK;
  struct Set { typename X6Operand<AK>::Regs a; typename X6Operand<BK_>::Regs b; };
  // Four register sets of requested slices: slice j travels in set j % 4.  Step s multiplies slice s out of LDS buffer
  // s & 1, moves slice s + 1 from its set into the other buffer (whose readers left at the last barrier) and asks
  // memory for slice s + 5 into the set that just emptied: four slices (3-4 us of memory latency at ~800 cycles a
  // step) are in flight.  With two sets the loop ran at the latency of one load per two steps (4500 cycles a slice).
  Set st0, st1, st2, st3;
  auto request = [&](int sl, Set& t) __attribute__((always_inline)) {
    oa.request(kb + sl * X6_BK, ke, t.a);
    ob.request(kb + sl * X6_BK, ke, t.b);
  };
  auto lds_a = [&](int buf) __attribute__((always_inline)) { return smem + buf * 2 * kX6OperandElems; };
  auto lds_b = [&](int buf) __attribute__((always_inline)) { return smem + buf * 2 * kX6OperandElems + kX6OperandElems; };
  const int ia = oa.lds_index(), ib = ob.lds_index();
  auto convert = [&](int buf, const Set& t) __attribute__((always_inline)) {     // prologue only
    u32x4 a0, a1, a2, b0, b1, b2;
    oa.pieces(t.a, a0, a1, a2);
    ob.pieces(t.b, b0, b1, b2);
    u32x4* da = reinterpret_cast<u32x4*>(lds_a(buf)) + ia;
    u32x4* db = reinterpret_cast<u32x4*>(lds_b(buf)) + ib;
    da[0] = a0; da[512] = a1; da[1024] = a2;
    db[0] = b0; db[512] = b1; db[1024] = b2;
  };
  // Every step issues the same loads and the same LDS writes whatever s is — a slice index past the end re-reads the
  // last slice and lands in the LDS buffer nobody reads any more.  With `if (s + 5 < nsl)` around the request the
  // compiler's wait-count pass lost track at the merge and waited for vmcnt(0) in every step (ALL loads, including
  // the ones just issued: one memory latency per slice, 4500 cycles; the ISA now shows vmcnt(18..23) in the loop).
  const int last = nsl - 1;
  auto clamp = [&](int sl) __attribute__((always_inline)) { return sl < last ? sl : last; };
  request(0, st0);
  request(clamp(1), st1);
  request(clamp(2), st2);
  request(clamp(3), st3);
  convert(0, st0);
  request(clamp(4), st0);
  __syncthreads();
  // this wave's fragments: groups 4 khalf + 2 q + hf (q = 0, 1), rows wm + r (A) / wn + r (B)
  const int fa = ((4 * pos.khalf + pos.hf) * 64 + pos.wm + pos.r) * 8;
  const int fb = ((4 * pos.khalf + pos.hf) * 64 + pos.wn + pos.r) * 8;
  constexpr int kStep = 2 * 64 * 8;             // one k-step (two groups) further
  // One step.  The 12 MFMAs of a wave are ONE dependent chain (same accumulator): the wave sits at each of them for its
  // 32-cycle pass, and whatever follows in program order waits behind.  Left to the scheduler, each fragment read was
  // sunk next to its MFMA (6 exposed LDS latencies per k-step: 3500 cycles a slice).  So the order is pinned: all twelve
  // fragment reads first, then the chain with the next slice's six LDS writes and the far slice's loads in its gaps.
  auto step = [&](int s, auto parity, Set& next, auto more) __attribute__((always_inline)) {
    constexpr int P = decltype(parity)::value;
    constexpr bool kRequest = decltype(more)::value;     // false in the tail: nothing left to ask for
    const __bf16* la = lds_a(P) + fa;
    const __bf16* lb = lds_b(P) + fb;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[q][pl] = *reinterpret_cast<const bf16x8*>(la + pl * kX6PlaneElems + q * kStep);
        b[q][pl] = *reinterpret_cast<const bf16x8*>(lb + pl * kX6PlaneElems + q * kStep);
      }
    u32x4 a0, a1, a2, b0, b1, b2;
    oa.pieces(next.a, a0, a1, a2);
    ob.pieces(next.b, b0, b1, b2);
    u32x4* da = reinterpret_cast<u32x4*>(lds_a(P ^ 1)) + ia;
    u32x4* db = reinterpret_cast<u32x4*>(lds_b(P ^ 1)) + ib;
    __builtin_amdgcn_sched_barrier(0);
#define DFM_X6_GAP(QA, PA, PB, FILL)                                                                \
    if (!(X6_ABLATE & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[QA][PA], b[QA][PB], acc, 0, 0, 0); \
    else acc[0] += static_cast<float>(a[QA][PA][0]) + static_cast<float>(b[QA][PB][0]);             \
    FILL;                                                                                           \
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool kW = !(X6_ABLATE & 4), kL = !(X6_ABLATE & 2);
    DFM_X6_GAP(0, 2, 0, if (kW) da[0] = a0)            // small terms first
    DFM_X6_GAP(0, 0, 2, if (kW) da[512] = a1)
    DFM_X6_GAP(0, 1, 1, if (kW) da[1024] = a2)
    DFM_X6_GAP(0, 1, 0, if (kW) db[0] = b0)
    DFM_X6_GAP(0, 0, 1, if (kW) db[512] = b1)
    DFM_X6_GAP(0, 0, 0, if (kW) db[1024] = b2)
    DFM_X6_GAP(1, 2, 0, if (kRequest && kL) oa.request(kb + clamp(s + 5) * X6_BK, ke, next.a))
    DFM_X6_GAP(1, 0, 2, if (kRequest && kL) ob.request(kb + clamp(s + 5) * X6_BK, ke, next.b))
    DFM_X6_GAP(1, 1, 1, (void)0)
    DFM_X6_GAP(1, 1, 0, (void)0)
    DFM_X6_GAP(1, 0, 1, (void)0)
    DFM_X6_GAP(1, 0, 0, (void)0)
#undef DFM_X6_GAP
    if (!(X6_ABLATE & 8)) __syncthreads();
  };
  const std::integral_constant<int, 0> even{};
  const std::integral_constant<int, 1> odd{};
  // whole groups of four steps in a loop WITHOUT exits in its body (a `break` between the steps is turned into a flag
  // and a common latch, where paths with different numbers of loads in flight merge — and the header waits for
  // vmcnt(0) again); the last 0-3 steps as nested straight-line code
  int s = 0;
  const std::true_type yes{};
  const std::false_type no{};
  for (; s + 4 <= nsl; s += 4) {
    step(s, even, st1, yes);
    step(s + 1, odd, st2, yes);
    step(s + 2, even, st3, yes);
    step(s + 3, odd, st0, yes);
  }
  if (s < nsl) {
    step(s, even, st1, no);
    if (s + 1 < nsl) {
      step(s + 1, odd, st2, no);
      if (s + 2 < nsl) step(s + 2, even, st3, no);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the redundant tail requests, before the registers are reused
  float* park = reinterpret_cast<float*>(smem);      // 16 KiB
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

}  // namespace gemm
}  // namespace dfm
