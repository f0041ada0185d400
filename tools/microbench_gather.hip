// Microbenchmark for the embedding gather (Criteo shape): product kernel through the C ABI
// next to reference access patterns, all in one process on the same data.
// Build: make -C tools   Run (GPU box): tools/microbench_gather [V] [B] [iters]
// Read per-kernel times from `rocprofv3 --kernel-trace --stats -- tools/microbench_gather`.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../include/deepfm_hip.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int S = 26, ND = 13, F = S + ND, D = 16;

struct Tabs {
  const float* w2[S];
  const int64_t* ids[S];
};

// A: ceiling probe — 4 lanes per 64-B row, one row per lane group, flat (field, sample) order,
// output written in the product layout (B, F, D).
__global__ __launch_bounds__(256) void raw_gather_rows(Tabs t, int B, float* __restrict__ out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid & 3;
  const int item = tid >> 2;  // (b, f) with f fastest?  no: sample-major tiles of 16 per field
  const int f = (item / 16) % S;
  const int b = (item / (16 * S)) * 16 + (item % 16);
  if (b >= B) return;
  const int64_t id = t.ids[f][b];
  const float4 v = *reinterpret_cast<const float4*>(t.w2[f] + id * D + q * 4);
  *reinterpret_cast<float4*>(out + (static_cast<int64_t>(b) * F + f) * D + q * 4) = v;
}

// B: same but ids pre-resolved to absolute row pointers (no dependent id load): isolates the
// row fetch itself.
__global__ __launch_bounds__(256) void raw_gather_ptrs(const float* const* __restrict__ rows, int n,
                                                       float* __restrict__ out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int q = tid & 3, item = tid >> 2;
  if (item >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(rows[item] + q * 4);
  *reinterpret_cast<float4*>(out + static_cast<int64_t>(item) * D + q * 4) = v;
}

// C: pure streaming copy of the same number of bytes (launch + HBM floor for this size)
__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ in, float4* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

__global__ void empty_kernel() {}

int main(int argc, char** argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 1000000;
  const int B = argc > 2 ? atoi(argv[2]) : 4096;
  const int iters = argc > 3 ? atoi(argv[3]) : 40;
  const int NB = 8;  // distinct id batches (rows touched: NB * 106K * 64 B = 54 MB of a 1.7 GB table set)
  std::mt19937_64 rng(1);

  std::vector<float*> w2(S), w1(S);
  for (int s = 0; s < S; ++s) {
    CK(hipMalloc(&w2[s], sizeof(float) * V * D));
    CK(hipMalloc(&w1[s], sizeof(float) * V));
    CK(hipMemset(w2[s], 0x3c, sizeof(float) * V * D));
    CK(hipMemset(w1[s], 0x3c, sizeof(float) * V));
  }
  std::vector<float*> dw2(ND), db2(ND), dw1(ND), db1(ND);
  for (int i = 0; i < ND; ++i) {
    CK(hipMalloc(&dw2[i], 64)); CK(hipMalloc(&db2[i], 64)); CK(hipMalloc(&dw1[i], 4)); CK(hipMalloc(&db1[i], 4));
    CK(hipMemset(dw2[i], 0, 64)); CK(hipMemset(db2[i], 0, 64)); CK(hipMemset(dw1[i], 0, 4)); CK(hipMemset(db1[i], 0, 4));
  }
  // id batches (NB, S, B) and dense (NB, ND, B)
  std::vector<int64_t> h_ids(static_cast<size_t>(NB) * S * B);
  for (auto& v : h_ids) v = 1 + rng() % (V - 1);
  int64_t* d_ids;
  float* d_x;
  CK(hipMalloc(&d_ids, h_ids.size() * 8));
  CK(hipMemcpy(d_ids, h_ids.data(), h_ids.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_x, sizeof(float) * NB * ND * B));
  CK(hipMemset(d_x, 0, sizeof(float) * NB * ND * B));
  // pre-resolved row pointers for probe B
  std::vector<const float*> h_rows(static_cast<size_t>(NB) * S * B);
  for (int nb = 0; nb < NB; ++nb)
    for (int s = 0; s < S; ++s)
      for (int b = 0; b < B; ++b)
        h_rows[(static_cast<size_t>(nb) * B + b) * S + s] = w2[s] + h_ids[(static_cast<size_t>(nb) * S + s) * B + b] * D;
  const float** d_rows;
  CK(hipMalloc(&d_rows, h_rows.size() * 8));
  CK(hipMemcpy(d_rows, h_rows.data(), h_rows.size() * 8, hipMemcpyHostToDevice));

  float *fo, *fe, *fm;
  CK(hipMalloc(&fo, 4 * B)); CK(hipMalloc(&fm, 4 * B));
  CK(hipMalloc(&fe, sizeof(float) * B * F * D));
  float4 *cp_in, *cp_out;
  const int cp_n = B * F * D / 4;
  CK(hipMalloc(&cp_in, 16 * cp_n)); CK(hipMalloc(&cp_out, 16 * cp_n));
  int32_t* err;
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));

  // product plan
  std::vector<dfm_field> fields(F);
  for (int f = 0; f < F; ++f) {
    dfm_field fd{};
    fd.dim = D;
    if (f < S) { fd.kind = DFM_SPARSE; fd.vocab = V; fd.w2 = w2[f]; fd.w1 = w1[f]; }
    else { int i = f - S; fd.kind = DFM_DENSE; fd.w2 = dw2[i]; fd.b2 = db2[i]; fd.w1 = dw1[i]; fd.b1 = db1[i]; }
    fields[f] = fd;
  }
  dfm_embedding_plan* plan;
  if (dfm_embedding_plan_create(fields.data(), F, D, &plan)) { fprintf(stderr, "%s\n", dfm_last_error()); return 1; }

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double algo = 4528.0 * B;

  auto inputs_for = [&](int nb, std::vector<const void*>& in) {
    in.resize(F);
    for (int s = 0; s < S; ++s) in[s] = d_ids + (static_cast<size_t>(nb) * S + s) * B;
    for (int i = 0; i < ND; ++i) in[S + i] = d_x + (static_cast<size_t>(nb) * ND + i) * B;
  };
  auto time_loop = [&](const char* name, auto&& launch) {
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
    printf("%-28s events avg %7.2f us  min %7.2f us  (%.0f GB/s algorithmic at min)\n", name, tot / iters * 1e3,
           mn * 1e3, algo / (mn * 1e-3) / 1e9);
  };

  std::vector<const void*> in;
  time_loop("empty_kernel", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); });
  time_loop("stream_copy(10MB)", [&](int) {
    hipLaunchKernelGGL(stream_copy, dim3((cp_n + 255) / 256), dim3(256), 0, st, cp_in, cp_out, cp_n);
  });
  time_loop("raw_gather_ptrs", [&](int nb) {
    const int n = S * B;
    hipLaunchKernelGGL(raw_gather_ptrs, dim3((n * 4 + 255) / 256), dim3(256), 0, st,
                       d_rows + static_cast<size_t>(nb) * n, n, fe);
  });
  time_loop("raw_gather_rows", [&](int nb) {
    Tabs t;
    for (int s = 0; s < S; ++s) { t.w2[s] = w2[s]; t.ids[s] = d_ids + (static_cast<size_t>(nb) * S + s) * B; }
    const int n = S * ((B + 15) / 16) * 16;
    hipLaunchKernelGGL(raw_gather_rows, dim3((n * 4 + 255) / 256), dim3(256), 0, st, t, B, fe);
  });
  time_loop("product dfm_embedding_forward", [&](int nb) {
    inputs_for(nb, in);
    if (dfm_embedding_forward(plan, in.data(), B, fo, fe, nullptr, fm, nullptr, nullptr, err, st)) {
      fprintf(stderr, "%s\n", dfm_last_error()); exit(1);
    }
  });
  CK(hipDeviceSynchronize());
  return 0;
}
