// Round-2 microbenchmark ladder for the fused embedding gather (Criteo shape, D = 16).
// Everything runs on the SAME packed 256-B row records the product uses (FeatureEmbedding.pack_tables_):
//   * the product kernel through the C ABI (dfm_embedding_forward),
//   * "sample-owner" candidates gather_owner<LPR>: a lane group of LPR lanes owns one sample across ALL
//     39 fields (no LDS, no barrier, reductions by lane shuffles),
//   * access-pattern probes (rows only / ids+rows) at several table sizes (TLB / cache reach),
//   * a stamped variant that records, per wave, when its ids and its rows arrived (s_memrealtime).
// Build: make -C tools microbench_gather2   Run on the GPU box:
//   rocprofv3 --kernel-trace --stats -d gpurun_out/mb2 -- tools/microbench_gather2 [B] [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/deepfm_hip.h"

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e = (x);                                                                  \
    if (e != hipSuccess) {                                                               \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

constexpr int S = 26, ND = 13, F = S + ND, D = 16, RS = 64;  // RS: floats per packed record

struct GArgs {
  const int64_t* ids[S];
  const float* tab[S];  // record base: w2 at +0, w1 at +D
  const float* x[ND];
  const float* dw2[ND];
  const float* db2[ND];
  const float* dw1[ND];
  const float* db1[ND];
  int64_t* ids_out[S];
  float* x_out[ND];
  const float* lab_src;
  float* lab_dst;
  int vocab;
};

template <int VW> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ float comp(const float2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float comp(const float& v, int) { return v; }
__device__ __forceinline__ void setc(float4& v, int i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else if (i == 2) v.z = f; else v.w = f; }
__device__ __forceinline__ void setc(float2& v, int i, float f) { if (i == 0) v.x = f; else v.y = f; }
__device__ __forceinline__ void setc(float& v, int, float f) { v = f; }

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(const float4& v, float* p) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); }
__device__ __forceinline__ void nt_store(const float2& v, float* p) { v2f t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(p)); }
__device__ __forceinline__ void nt_store(const float& v, float* p) { __builtin_nontemporal_store(v, p); }

// A lane group of LPR lanes owns sample b for every field.  Loads are issued in three rounds (all ids and
// dense values; all rows + first-order scalars + dense-field weights; then arithmetic and stores).
template <int LPR, bool STAGE, bool STAMP, bool NTST = false>
__global__ __launch_bounds__(64) void gather_owner(GArgs a, int B, float* __restrict__ fo_out, float* __restrict__ fe,
                                                   float* __restrict__ fm_out, float* __restrict__ fm_sum,
                                                   int* __restrict__ err, unsigned long long* __restrict__ stamps) {
  constexpr int VW = D / LPR, SPW = 64 / LPR;
  using V = typename Vec<VW>::T;
  const int lane = threadIdx.x;
  const int s = lane / LPR, q = lane % LPR;
  const int b = blockIdx.x * SPW + s;
  const bool live = b < B;
  const int bc = live ? b : B - 1;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, tk = 0;
  if (STAMP) {
    t0 = wall_clock64();
    // first touch of the kernel-argument segment: one pointer from it, waited for
    const int64_t* p0 = a.ids[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(p0) : "memory");
    tk = wall_clock64();
  }
  int64_t id[S];
#pragma unroll
  for (int f = 0; f < S; ++f) id[f] = a.ids[f][bc];
  float xv[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) xv[j] = a.x[j][bc];
  if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = wall_clock64(); }
  V row[S];
  float w1v[S];
  bool bad = false;
#pragma unroll
  for (int f = 0; f < S; ++f) {
    const bool oob = static_cast<uint64_t>(id[f]) >= static_cast<uint64_t>(a.vocab);
    bad |= oob;
    const int64_t i = oob ? 0 : id[f];
    const float* r = a.tab[f] + i * RS;
    row[f] = *reinterpret_cast<const V*>(r + q * VW);
    w1v[f] = r[D];
  }
  V dw[ND], db[ND];
  float dw1[ND], db1[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    dw[j] = *reinterpret_cast<const V*>(a.dw2[j] + q * VW);
    db[j] = *reinterpret_cast<const V*>(a.db2[j] + q * VW);
    dw1[j] = a.dw1[j][0];
    db1[j] = a.db1[j][0];
  }
  if (STAGE && live && q == 0) {
#pragma unroll
    for (int f = 0; f < S; ++f) a.ids_out[f][b] = id[f];
#pragma unroll
    for (int j = 0; j < ND; ++j) a.x_out[j][b] = xv[j];
    a.lab_dst[b] = a.lab_src[b];
  }
  if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t2 = wall_clock64(); }
  float Sv[VW], SQ[VW];
#pragma unroll
  for (int c = 0; c < VW; ++c) Sv[c] = SQ[c] = 0.f;
  float fo = 0.f;
  float* out = fe + static_cast<int64_t>(b) * F * D + q * VW;
#pragma unroll
  for (int f = 0; f < S; ++f) {
#pragma unroll
    for (int c = 0; c < VW; ++c) {
      const float e = comp(row[f], c);
      Sv[c] += e;
      SQ[c] = fmaf(e, e, SQ[c]);
    }
    fo += w1v[f];
    if (live) {
      if (NTST) nt_store(row[f], out + f * D);
      else *reinterpret_cast<V*>(out + f * D) = row[f];
    }
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    V e;
#pragma unroll
    for (int c = 0; c < VW; ++c) {
      const float v = fmaf(xv[j], comp(dw[j], c), comp(db[j], c));
      setc(e, c, v);
      Sv[c] += v;
      SQ[c] = fmaf(v, v, SQ[c]);
    }
    fo += fmaf(xv[j], dw1[j], db1[j]);
    if (live) {
      if (NTST) nt_store(e, out + (S + j) * D);
      else *reinterpret_cast<V*>(out + (S + j) * D) = e;
    }
  }
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < VW; ++c) t += Sv[c] * Sv[c] - SQ[c];
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, 64);
  if (live) {
    V sv;
#pragma unroll
    for (int c = 0; c < VW; ++c) setc(sv, c, Sv[c]);
    *reinterpret_cast<V*>(fm_sum + static_cast<int64_t>(b) * D + q * VW) = sv;
    if (q == 0) { fo_out[b] = fo; fm_out[b] = 0.5f * t; }
  }
  if (bad) atomicOr(err, 1);
  if (STAMP && lane == 0) {
    unsigned long long* p = stamps + static_cast<size_t>(blockIdx.x) * 8;
    p[0] = t0; p[1] = t1; p[2] = t2; p[3] = wall_clock64(); p[4] = tk;
  }
}


// owner<4> with explicit cache policies, all through compiler-visible builtins (an inline-asm load completes
// asynchronously behind the register allocator's back - the first version of this probe faulted that way):
// rows and first-order scalars as raw buffer loads with an aux cache policy (RPOL: 0 default, 1 sc0,
// 2 nt, 17 sc0 sc1), output stores optionally nt.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int RPOL, bool NTST, int TAG = 0>
__global__ __launch_bounds__(64) void gather_owner_pol(GArgs a, int B, float* __restrict__ fo_out, float* __restrict__ fe,
                                                       float* __restrict__ fm_out, float* __restrict__ fm_sum,
                                                       int* __restrict__ err) {
  const int lane = threadIdx.x;
  const int s = lane >> 2, q = lane & 3;
  const int b = blockIdx.x * 16 + s;
  const bool live = b < B;
  const int bc = live ? b : B - 1;
  int64_t id[S];
#pragma unroll
  for (int f = 0; f < S; ++f) id[f] = a.ids[f][bc];
  float xv[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) xv[j] = a.x[j][bc];
  v4f row[S];
  float w1v[S];
  bool bad = false;
#pragma unroll
  for (int f = 0; f < S; ++f) {
    const bool oob = static_cast<uint64_t>(id[f]) >= static_cast<uint64_t>(a.vocab);
    bad |= oob;
    const unsigned off = (oob ? 0u : static_cast<unsigned>(id[f])) * (RS * 4u);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tab[f]), 0, 0x7fffffff, 0x00020000);
    const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + q * 16u, 0, RPOL);
    row[f] = __builtin_bit_cast(v4f, r);
    w1v[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off + D * 4u, 0, RPOL));
  }
  float4 dw[ND], db[ND];
  float dw1[ND], db1[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    dw[j] = *reinterpret_cast<const float4*>(a.dw2[j] + q * 4);
    db[j] = *reinterpret_cast<const float4*>(a.db2[j] + q * 4);
    dw1[j] = a.dw1[j][0];
    db1[j] = a.db1[j][0];
  }
  float Sv[4] = {0.f, 0.f, 0.f, 0.f}, SQ[4] = {0.f, 0.f, 0.f, 0.f};
  float fo = 0.f;
  float* out = fe + static_cast<int64_t>(b) * F * D + q * 4;
#pragma unroll
  for (int f = 0; f < S; ++f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float e = row[f][c]; Sv[c] += e; SQ[c] = fmaf(e, e, SQ[c]); }
    fo += w1v[f];
    if (live) {
      if (NTST) __builtin_nontemporal_store(row[f], reinterpret_cast<v4f*>(out + f * D));
      else *reinterpret_cast<v4f*>(out + f * D) = row[f];
    }
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    v4f e;
    e[0] = fmaf(xv[j], dw[j].x, db[j].x); e[1] = fmaf(xv[j], dw[j].y, db[j].y);
    e[2] = fmaf(xv[j], dw[j].z, db[j].z); e[3] = fmaf(xv[j], dw[j].w, db[j].w);
#pragma unroll
    for (int c = 0; c < 4; ++c) { Sv[c] += e[c]; SQ[c] = fmaf(e[c], e[c], SQ[c]); }
    fo += fmaf(xv[j], dw1[j], db1[j]);
    if (live) {
      if (NTST) __builtin_nontemporal_store(e, reinterpret_cast<v4f*>(out + (S + j) * D));
      else *reinterpret_cast<v4f*>(out + (S + j) * D) = e;
    }
  }
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) t += Sv[c] * Sv[c] - SQ[c];
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  if (live) {
    v4f sv = {Sv[0], Sv[1], Sv[2], Sv[3]};
    *reinterpret_cast<v4f*>(fm_sum + static_cast<int64_t>(b) * D + q * 4) = sv;
    if (q == 0) { fo_out[b] = fo; fm_out[b] = 0.5f * t; }
  }
  if (bad) atomicOr(err, 1);
}

// What the tail of the previous training step would do for the next batch: copy its record (ids, dense
// values, labels) into the static input buffers.  Block j copies samples [16 j, 16 j + 16) of every field,
// i.e. exactly what gather workgroup j reads - and both run on XCD j % 8.  READ_ONLY: touch without copying.
template <bool READ_ONLY>
__global__ __launch_bounds__(64) void stage_batch(GArgs a, int B, float* __restrict__ sink) {
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * 16;
  float acc = 0.f;
  // ids: 26 fields x 16 ids = 416 int64 per block; lane l copies items l, l + 64, ...
  for (int it = lane; it < S * 16; it += 64) {
    const int f = it >> 4, b = b0 + (it & 15);
    if (b < B) {
      const int64_t v = a.ids[f][b];
      if (READ_ONLY) acc += static_cast<float>(v); else a.ids_out[f][b] = v;
    }
  }
  for (int it = lane; it < (ND + 1) * 16; it += 64) {
    const int j = it >> 4, b = b0 + (it & 15);
    if (b < B) {
      const float v = j < ND ? a.x[j][b] : a.lab_src[b];
      if (READ_ONLY) acc += v; else if (j < ND) a.x_out[j][b] = v; else a.lab_dst[b] = v;
    }
  }
  if (READ_ONLY && acc == 123.456f) sink[0] = acc;
}

// probes ------------------------------------------------------------------------------------------
struct Tabs { const float* tab[S]; const int64_t* ids[S]; };
// ids -> rows, 4 lanes per row, output in product layout; stride in floats
__global__ __launch_bounds__(256) void probe_rows(Tabs t, int B, int stride, float* __restrict__ out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid & 3, item = tid >> 2;
  const int f = (item / 16) % S, b = (item / (16 * S)) * 16 + (item % 16);
  if (b >= B) return;
  const int64_t id = t.ids[f][b];
  const float4 v = *reinterpret_cast<const float4*>(t.tab[f] + id * stride + q * 4);
  *reinterpret_cast<float4*>(out + (static_cast<int64_t>(b) * F + f) * D + q * 4) = v;
}
// same access with a cache policy on the row load: 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0
template <int POLICY>
__global__ __launch_bounds__(256) void probe_rows_policy(Tabs t, int B, int stride, float* __restrict__ out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid & 3, item = tid >> 2;
  const int f = (item / 16) % S, b = (item / (16 * S)) * 16 + (item % 16);
  if (b >= B) return;
  const int64_t id = t.ids[f][b];
  const float* p = t.tab[f] + id * stride + q * 4;
  float4 v;
  if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  *reinterpret_cast<float4*>(out + (static_cast<int64_t>(b) * F + f) * D + q * 4) = v;
}
// 128 B per row (8 lanes x 16 B): what a full-line fetch looks like when all of it is used
__global__ __launch_bounds__(256) void probe_rows128(Tabs t, int B, int stride, float* __restrict__ out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid & 7, item = tid >> 3;
  const int f = (item / 8) % S, b = (item / (8 * S)) * 8 + (item % 8);
  if (b >= B) return;
  const int64_t id = t.ids[f][b];
  const float4 v = *reinterpret_cast<const float4*>(t.tab[f] + id * stride + q * 4);
  if (q < 4) *reinterpret_cast<float4*>(out + (static_cast<int64_t>(b) * F + f) * D + q * 4) = v;
  else if (v.x == 123.456f) out[0] = v.y;
}
__global__ void empty_kernel() {}
__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ in, float4* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void stream_write(float4* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char** argv) {
  const int Bmax = 65536;
  const int B0 = argc > 1 ? atoi(argv[1]) : 4096;
  const int iters = argc > 2 ? atoi(argv[2]) : 30;
  const int V = 1000000;
  const int NB = 8;
  std::mt19937_64 rng(1);
  std::vector<float*> tab(S);
  const char* mt = getenv("MB_TABLE_MEM");   // "uncached" / "finegrained": memory type of the tables
  for (int s = 0; s < S; ++s) {
    if (mt && !strcmp(mt, "uncached"))
      CK(hipExtMallocWithFlags((void**)&tab[s], sizeof(float) * (size_t)V * RS, hipDeviceMallocUncached));
    else if (mt && !strcmp(mt, "finegrained"))
      CK(hipExtMallocWithFlags((void**)&tab[s], sizeof(float) * (size_t)V * RS, hipDeviceMallocFinegrained));
    else
      CK(hipMalloc(&tab[s], sizeof(float) * (size_t)V * RS));
    CK(hipMemset(tab[s], 0x3c, sizeof(float) * (size_t)V * RS));
  }
  printf("table memory: %s\n", mt ? mt : "default (coarse-grained)");
  // small dense "unpacked-size" tables for the reach probes: the first V/4 records of each table act as
  // a (V, 16) contiguous table (stride 16), and V2 = V/16 tables test cache-resident behaviour
  std::vector<float*> dw2(ND), db2(ND), dw1(ND), db1(ND);
  for (int i = 0; i < ND; ++i) {
    CK(hipMalloc(&dw2[i], 64)); CK(hipMalloc(&db2[i], 64)); CK(hipMalloc(&dw1[i], 4)); CK(hipMalloc(&db1[i], 4));
    std::vector<float> h(16);
    for (auto& v : h) v = (float)(rng() % 1000) / 1000.f;
    CK(hipMemcpy(dw2[i], h.data(), 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2[i], h.data(), 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw1[i], h.data(), 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1[i], h.data(), 4, hipMemcpyHostToDevice));
  }
  // random weights in the rows that will be touched are irrelevant for timing; fill a few for checks
  std::vector<int64_t> h_ids((size_t)NB * S * Bmax);
  for (auto& v : h_ids) v = 1 + rng() % (V - 1);
  for (size_t i = 0; i < h_ids.size(); i += 97) h_ids[i] = 0;   // ~1 % padding ids
  int64_t* d_ids;
  float *d_x, *d_lab;
  CK(hipMalloc(&d_ids, h_ids.size() * 8));
  CK(hipMemcpy(d_ids, h_ids.data(), h_ids.size() * 8, hipMemcpyHostToDevice));
  std::vector<float> h_x((size_t)NB * ND * Bmax);
  for (auto& v : h_x) v = (float)(rng() % 4096) / 4096.f;
  CK(hipMalloc(&d_x, h_x.size() * 4));
  CK(hipMemcpy(d_x, h_x.data(), h_x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_lab, (size_t)NB * Bmax * 4));
  CK(hipMemset(d_lab, 0, (size_t)NB * Bmax * 4));
  // scatter distinct values into the touched rows so that outputs can be compared
  {
    std::vector<float> rec(RS);
    for (int nb = 0; nb < 1; ++nb)
      for (int s = 0; s < S; ++s)
        for (int b = 0; b < 4096; ++b) {
          const int64_t id = h_ids[((size_t)nb * S + s) * Bmax + b];
          if (id == 0) continue;
          for (int j = 0; j < RS; ++j) rec[j] = (float)((id * 31 + j * 7 + s) % 1009) / 1009.f - 0.5f;
          CK(hipMemcpy(tab[s] + id * RS, rec.data(), RS * 4, hipMemcpyHostToDevice));
        }
    for (int s = 0; s < S; ++s) CK(hipMemset(tab[s], 0, RS * 4));
  }

  int64_t* st_ids; float *st_x, *st_lab;
  CK(hipMalloc(&st_ids, (size_t)S * Bmax * 8)); CK(hipMalloc(&st_x, (size_t)ND * Bmax * 4)); CK(hipMalloc(&st_lab, (size_t)Bmax * 4));
  float *fo, *fm, *fsum, *fe, *fo2, *fm2, *fsum2, *fe2;
  CK(hipMalloc(&fo, 4 * Bmax)); CK(hipMalloc(&fm, 4 * Bmax)); CK(hipMalloc(&fsum, 4 * Bmax * D));
  CK(hipMalloc(&fe, sizeof(float) * (size_t)Bmax * F * D));
  CK(hipMalloc(&fo2, 4 * Bmax)); CK(hipMalloc(&fm2, 4 * Bmax)); CK(hipMalloc(&fsum2, 4 * Bmax * D));
  CK(hipMalloc(&fe2, sizeof(float) * (size_t)Bmax * F * D));
  int32_t* err;
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 8 * 8 * 16384));
  float4 *cp_in, *cp_out;
  const int cp_n = 4096 * F * D / 4;
  CK(hipMalloc(&cp_in, 16 * cp_n)); CK(hipMalloc(&cp_out, 16 * cp_n));

  // product plan (packed strides)
  std::vector<dfm_field> fields(F);
  for (int f = 0; f < F; ++f) {
    dfm_field fd{};
    fd.dim = D;
    if (f < S) { fd.kind = DFM_SPARSE; fd.vocab = V; fd.w2 = tab[f]; fd.w1 = tab[f] + D; fd.stride2 = RS; fd.stride1 = RS; }
    else { int i = f - S; fd.kind = DFM_DENSE; fd.w2 = dw2[i]; fd.b2 = db2[i]; fd.w1 = dw1[i]; fd.b1 = db1[i]; }
    fields[f] = fd;
  }
  dfm_embedding_plan* plan;
  if (dfm_embedding_plan_create(fields.data(), F, D, &plan)) { fprintf(stderr, "%s\n", dfm_last_error()); return 1; }

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto gargs = [&](int nb, int B) {
    GArgs a{};
    for (int s = 0; s < S; ++s) {
      a.ids[s] = d_ids + ((size_t)nb * S + s) * Bmax;
      a.tab[s] = tab[s];
      a.ids_out[s] = st_ids + (size_t)s * Bmax;
    }
    for (int j = 0; j < ND; ++j) {
      a.x[j] = d_x + ((size_t)nb * ND + j) * Bmax;
      a.dw2[j] = dw2[j]; a.db2[j] = db2[j]; a.dw1[j] = dw1[j]; a.db1[j] = db1[j];
      a.x_out[j] = st_x + (size_t)j * Bmax;
    }
    a.lab_src = d_lab + (size_t)nb * Bmax;
    a.lab_dst = st_lab;
    a.vocab = V;
    return a;
  };
  auto time_loop = [&](const char* name, int B, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch(i % NB);
    CK(hipStreamSynchronize(st));
    float tot = 0, mn = 1e9;
    for (int i = 0; i < iters; ++i) {
      CK(hipEventRecord(e0, st));
      launch(i % NB);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms; mn = ms < mn ? ms : mn;
    }
    printf("%-40s B=%6d events avg %7.2f us  min %7.2f us  (%.0f GB/s algorithmic at avg)\n", name, B,
           tot / iters * 1e3, mn * 1e3, 4528.0 * B / (tot / iters * 1e-3) / 1e9);
  };
  std::vector<const void*> in(F);
  auto product = [&](int nb, int B, float* o_fo, float* o_fe, float* o_fm, float* o_fsum) {
    for (int s = 0; s < S; ++s) in[s] = d_ids + ((size_t)nb * S + s) * Bmax;
    for (int i = 0; i < ND; ++i) in[S + i] = d_x + ((size_t)nb * ND + i) * Bmax;
    if (dfm_embedding_forward(plan, in.data(), B, o_fo, o_fe, nullptr, o_fm, o_fsum, nullptr, err, st)) {
      fprintf(stderr, "%s\n", dfm_last_error()); exit(1);
    }
  };

  // ---- correctness of the candidates against the product kernel (batch 0, B0)
  {
    product(0, B0, fo, fe, fm, fsum);
    auto check = [&](const char* name) {
      CK(hipStreamSynchronize(st));
      std::vector<float> a((size_t)B0 * F * D), b((size_t)B0 * F * D), c(B0), d(B0);
      CK(hipMemcpy(a.data(), fe, a.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), fe2, b.size() * 4, hipMemcpyDeviceToHost));
      const bool same = memcmp(a.data(), b.data(), a.size() * 4) == 0;
      double dfo = 0, dfm = 0, mfm = 0;
      CK(hipMemcpy(c.data(), fo, B0 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), fo2, B0 * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < B0; ++i) dfo = std::max(dfo, (double)fabsf(c[i] - d[i]));
      CK(hipMemcpy(c.data(), fm, B0 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), fm2, B0 * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < B0; ++i) { dfm = std::max(dfm, (double)fabsf(c[i] - d[i])); mfm = std::max(mfm, (double)fabsf(c[i])); }
      printf("check %-28s fe bitwise %s, max|d fo| %.2e, max|d fm| %.2e (max |fm| %.2f)\n", name, same ? "EQUAL" : "DIFFERENT", dfo, dfm, mfm);
    };
#define RUN_OWNER(LPR, STAGE, STAMP, nb, B, FO, FE, FM, FS)                                                              \
  hipLaunchKernelGGL((gather_owner<LPR, STAGE, STAMP>), dim3(((B) + 64 / LPR - 1) / (64 / LPR)), dim3(64), 0, st,        \
                     gargs(nb, B), B, FO, FE, FM, FS, err, stamps)
    RUN_OWNER(4, false, false, 0, B0, fo2, fe2, fm2, fsum2); check("owner<4>");
    CK(hipMemsetAsync(fe2, 0, (size_t)B0 * F * D * 4, st));
    RUN_OWNER(8, false, false, 0, B0, fo2, fe2, fm2, fsum2); check("owner<8>");
    CK(hipMemsetAsync(fe2, 0, (size_t)B0 * F * D * 4, st));
    RUN_OWNER(16, true, false, 0, B0, fo2, fe2, fm2, fsum2); check("owner<16,stage>");
    CK(hipMemsetAsync(fe2, 0, (size_t)B0 * F * D * 4, st));
    hipLaunchKernelGGL((gather_owner_pol<1, true>), dim3((B0 + 15) / 16), dim3(64), 0, st, gargs(0, B0), B0, fo2, fe2, fm2, fsum2, err);
    check("pol<sc0,nt>");
    {
      GArgs a = gargs(0, B0);
      hipLaunchKernelGGL(stage_batch<false>, dim3((B0 + 15) / 16), dim3(64), 0, st, a, B0, fo);
      for (int s2 = 0; s2 < S; ++s2) a.ids[s2] = a.ids_out[s2];
      for (int j = 0; j < ND; ++j) a.x[j] = a.x_out[j];
      CK(hipMemsetAsync(fe2, 0, (size_t)B0 * F * D * 4, st));
      hipLaunchKernelGGL((gather_owner_pol<0, false>), dim3((B0 + 15) / 16), dim3(64), 0, st, a, B0, fo2, fe2, fm2, fsum2, err);
      check("stage_batch + pol<plain,plain>");
    }
  }

  // ---- ladder at B0
  time_loop("empty_kernel<<<1,64>>>", B0, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); });
  time_loop("empty_kernel<<<1024,64>>>", B0, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(64), 0, st); });
  time_loop("stream_write(10MB)", B0, [&](int) { hipLaunchKernelGGL(stream_write, dim3((cp_n + 255) / 256), dim3(256), 0, st, cp_out, cp_n); });
  time_loop("stream_copy(10MB)", B0, [&](int) { hipLaunchKernelGGL(stream_copy, dim3((cp_n + 255) / 256), dim3(256), 0, st, cp_in, cp_out, cp_n); });
  auto tabs_for = [&](int nb, int vocab_limit) {
    Tabs t;
    for (int s = 0; s < S; ++s) { t.tab[s] = tab[s]; t.ids[s] = d_ids + ((size_t)nb * S + s) * Bmax; }
    (void)vocab_limit;
    return t;
  };
  const int nrow = S * ((B0 + 15) / 16) * 16;
  time_loop("probe_rows packed (6.6 GB reach)", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("probe_rows stride16 (1.7 GB reach)", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, 16, fe2);
  });
  time_loop("probe_rows stride1 (104 MB reach)", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, 1, fe2);
  });
  time_loop("probe_rows nt", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows_policy<1>, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("probe_rows sc1", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows_policy<2>, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("probe_rows sc0 sc1", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows_policy<3>, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("probe_rows sc0", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows_policy<4>, dim3((nrow * 4 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("probe_rows128 (full line used)", B0, [&](int nb) {
    hipLaunchKernelGGL(probe_rows128, dim3((nrow * 8 + 255) / 256), dim3(256), 0, st, tabs_for(nb, V), B0, RS, fe2);
  });
  time_loop("product dfm_embedding_forward", B0, [&](int nb) { product(nb, B0, fo, fe, fm, fsum); });
  time_loop("owner<4>", B0, [&](int nb) { RUN_OWNER(4, false, false, nb, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<8>", B0, [&](int nb) { RUN_OWNER(8, false, false, nb, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<16>", B0, [&](int nb) { RUN_OWNER(16, false, false, nb, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<16> same batch (ids hot)", B0, [&](int) { RUN_OWNER(16, false, false, 0, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<16> nt stores", B0, [&](int nb) {
    hipLaunchKernelGGL((gather_owner<16, false, false, true>), dim3((B0 + 3) / 4), dim3(64), 0, st, gargs(nb, B0), B0, fo2, fe2, fm2, fsum2, err, stamps);
  });
  time_loop("owner<4> nt stores", B0, [&](int nb) {
    hipLaunchKernelGGL((gather_owner<4, false, false, true>), dim3((B0 + 15) / 16), dim3(64), 0, st, gargs(nb, B0), B0, fo2, fe2, fm2, fsum2, err, stamps);
  });
  // policies on the owner<4> layout
  {
    auto pol = [&](const char* name, auto kern, bool staged_ids, int pre) {
      time_loop(name, B0, [&, kern, staged_ids, pre](int nb) {
        GArgs a = gargs(nb, B0);
        if (pre == 1) hipLaunchKernelGGL(stage_batch<false>, dim3((B0 + 15) / 16), dim3(64), 0, st, a, B0, fo);
        if (pre == 2) hipLaunchKernelGGL(stage_batch<true>, dim3((B0 + 15) / 16), dim3(64), 0, st, a, B0, fo);
        if (staged_ids) {
          for (int s2 = 0; s2 < S; ++s2) a.ids[s2] = a.ids_out[s2];
          for (int j = 0; j < ND; ++j) a.x[j] = a.x_out[j];
        }
        hipLaunchKernelGGL(kern, dim3((B0 + 15) / 16), dim3(64), 0, st, a, B0, fo2, fe2, fm2, fsum2, err);
      });
    };
    pol("pol<plain,plain>", gather_owner_pol<0, false>, false, 0);
    pol("pol<sc0,plain>", gather_owner_pol<1, false>, false, 0);
    pol("pol<plain,nt>", gather_owner_pol<0, true>, false, 0);
    pol("pol<sc0,nt>", gather_owner_pol<1, true>, false, 0);
    pol("pol<nt,nt>", gather_owner_pol<2, true>, false, 0);
    pol("pol<sc0sc1,nt>", gather_owner_pol<17, true>, false, 0);
    pol("stage_batch + pol<sc0,nt> on staged ids", gather_owner_pol<1, true, 1>, true, 1);
    pol("stage_batch + pol<plain,plain> on staged ids", gather_owner_pol<0, false, 1>, true, 1);
    pol("prefetch(read) + pol<sc0,nt> on record ids", gather_owner_pol<1, true, 2>, false, 2);
  }
  time_loop("owner<4,stage>", B0, [&](int nb) { RUN_OWNER(4, true, false, nb, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<8,stage>", B0, [&](int nb) { RUN_OWNER(8, true, false, nb, B0, fo2, fe2, fm2, fsum2); });
  time_loop("owner<16,stage>", B0, [&](int nb) { RUN_OWNER(16, true, false, nb, B0, fo2, fe2, fm2, fsum2); });

  // ---- where the time goes inside a wave (owner<8>, stamped)
  for (int lpr : {4, 8, 16}) {
    const int nwg = (B0 + 64 / lpr - 1) / (64 / lpr);
    for (int rep = 0; rep < 3; ++rep) {
      if (lpr == 4) RUN_OWNER(4, false, true, rep + 1, B0, fo2, fe2, fm2, fsum2);
      if (lpr == 8) RUN_OWNER(8, false, true, rep + 1, B0, fo2, fe2, fm2, fsum2);
      if (lpr == 16) RUN_OWNER(16, false, true, rep + 1, B0, fo2, fe2, fm2, fsum2);
      CK(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> h((size_t)nwg * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> d_ids_t, d_rows_t, d_tail_t, starts, d_karg;
    for (int w = 0; w < nwg; ++w) {
      first = std::min(first, h[w * 8]); last = std::max(last, h[w * 8 + 3]);
    }
    for (int w = 0; w < nwg; ++w) {
      starts.push_back((h[w * 8] - first) * 10.0);
      d_karg.push_back((h[w * 8 + 4] - h[w * 8]) * 10.0);
      d_ids_t.push_back((h[w * 8 + 1] - h[w * 8 + 4]) * 10.0);
      d_rows_t.push_back((h[w * 8 + 2] - h[w * 8 + 1]) * 10.0);
      d_tail_t.push_back((h[w * 8 + 3] - h[w * 8 + 2]) * 10.0);
    }
    auto stat = [](std::vector<double>& v, const char* n) {
      std::sort(v.begin(), v.end());
      printf("    %-22s median %6.0f ns  p90 %6.0f ns  max %6.0f ns\n", n, v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    printf("stamps owner<%d> (%d waves): first wave start -> last wave end %.0f ns (s_memrealtime, 10 ns ticks)\n", lpr, nwg,
           (last - first) * 10.0);
    stat(starts, "wave start offset"); stat(d_karg, "kernarg first touch"); stat(d_ids_t, "ids + dense arrive"); stat(d_rows_t, "rows arrive"); stat(d_tail_t, "math + stores issued");
  }

  // ---- batch sweep: product kernel and the candidates
  if (argc > 3 && atoi(argv[3]) == 1)
  for (int B : {4096, 8192, 16384, 32768, 65536}) {
    time_loop("sweep product", B, [&](int nb) { product(nb, B, fo, fe, fm, fsum); });
    time_loop("sweep owner<4>", B, [&](int nb) { RUN_OWNER(4, false, false, nb, B, fo2, fe2, fm2, fsum2); });
    time_loop("sweep owner<8>", B, [&](int nb) { RUN_OWNER(8, false, false, nb, B, fo2, fe2, fm2, fsum2); });
    time_loop("sweep owner<16>", B, [&](int nb) { RUN_OWNER(16, false, false, nb, B, fo2, fe2, fm2, fsum2); });
  }
  CK(hipDeviceSynchronize());
  int herr = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("error flag %d\n", herr);
  return 0;
}
