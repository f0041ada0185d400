// Where does a tower GEMM workgroup spend its life?  Runs gemm_core.h's mainloop on the layer-1
// shapes with s_memtime stamps per workgroup: [start, after the k loop, end] and prints the
// distribution, plus per-dispatch kernel times (hipExtLaunchKernelGGL events).
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../deepfm_amd/csrc/gemm_core.h"

using namespace dfm;
using namespace dfm::gemm;

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(kThreads) void probe(const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                                                  int M, int N, int K, int tiles_n, int k_per_split,
                                                  unsigned long long* stamps) {
  __shared__ Smem sm;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const TilePos pos;
  const int lt = xcd_logical_index(blockIdx.x, gridDim.x);
  const int tiles = tiles_n * ((M + BM - 1) / BM);
  const int sp = lt / tiles, tl = lt % tiles;
  const int m0 = (tl / tiles_n) * BM, n0 = (tl % tiles_n) * BN;
  const int kb = sp * k_per_split, ke = kb + k_per_split < K ? kb + k_per_split : K;
  f32x16 acc = {};
  mainloop<A_KC, B_KC, true, true>(A, lda, B, ldb, M, N, m0, n0, kb, ke, sm, pos, acc);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (pos.khalf == 0) {
    const int n = n0 + pos.col();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + pos.row(reg);
      if (m < M && n < N) C[(static_cast<int64_t>(sp) * M + m) * N + n] = acc[reg];
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 3 + 0] = t0;
    stamps[blockIdx.x * 3 + 1] = t1;
    stamps[blockIdx.x * 3 + 2] = t2;
  }
}

template <bool A_KC, bool B_KC>
void run(const char* name, int M, int N, int K, int splits, float* dA, float* dB, float* dC, unsigned long long* dS) {
  const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
  const int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  const int sp = (K + kps - 1) / kps;
  const int blocks = tn * tm * sp;
  const int64_t lda = A_KC ? K : M, ldb = B_KC ? K : N;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum = 0;
  for (int it = 0; it < 12; ++it) {
    hipExtLaunchKernelGGL((probe<A_KC, B_KC>), dim3(blocks), dim3(kThreads), 0, 0, e0, e1, 0, dA, lda, dB, ldb, dC, M, N,
                          K, tn, kps, dS);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) { best = std::min(best, ms); sum += ms; }
  }
  std::vector<unsigned long long> h(blocks * 3);
  hipMemcpy(h.data(), dS, sizeof(unsigned long long) * blocks * 3, hipMemcpyDeviceToHost);
  unsigned long long first = ~0ull, last = 0;
  std::vector<double> loop, epi, start;
  for (int b = 0; b < blocks; ++b) { first = std::min(first, h[b * 3]); last = std::max(last, h[b * 3 + 2]); }
  for (int b = 0; b < blocks; ++b) {
    start.push_back((h[b * 3] - first) / 100.0);           // s_memtime: 100 MHz constant clock -> us
    loop.push_back((h[b * 3 + 1] - h[b * 3]) / 100.0);
    epi.push_back((h[b * 3 + 2] - h[b * 3 + 1]) / 100.0);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
  const int slices = kps / BK;
  printf("%-22s M%d N%d K%d splits%d: %d wgs x %d slices | kernel %.1f us (best %.1f) | wg start med %.1f max %.1f | "
         "k-loop med %.1f max %.1f (%.2f us/slice) | epilogue med %.1f max %.1f | span %.1f us\n",
         name, M, N, K, sp, blocks, slices, sum / 10 * 1e3, best * 1e3, med(start), mx(start), med(loop), mx(loop),
         med(loop) / slices, med(epi), mx(epi), (last - first) / 100.0);
}

int main() {
  const size_t n = 4096ull * 4096;
  float *dA, *dB, *dC; unsigned long long* dS;
  hipMalloc(&dA, n * 4); hipMalloc(&dB, n * 4); hipMalloc(&dC, n * 4 * 2); hipMalloc(&dS, 8 * 3 * 8192);
  hipMemset(dA, 0, n * 4); hipMemset(dB, 0, n * 4);
  run<true, true>("fwd L1 (x W^T)", 4096, 256, 624, 1, dA, dB, dC, dS);
  run<true, true>("fwd L2", 4096, 128, 256, 1, dA, dB, dC, dS);
  run<true, false>("dx L1 (dz W)", 4096, 624, 256, 1, dA, dB, dC, dS);
  run<false, false>("dW L1 (dz^T x)", 256, 624, 4096, 9, dA, dB, dC, dS);
  run<false, false>("dW L1 13 splits", 256, 624, 4096, 13, dA, dB, dC, dS);
  run<true, true>("fwd L1 K=640", 4096, 256, 640, 1, dA, dB, dC, dS);
  return 0;
}
