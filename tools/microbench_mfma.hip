// Effective fp32-MFMA rate on the box: dependent v_mfma_f32_32x32x2_f32 chains, 1 or 2 accumulators
// per wave, 2 or 4 waves per SIMD.  Prints ns per MFMA per wave and the implied TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACCS>
__global__ __launch_bounds__(512) void chain(float* out, int n, float a, float b) {
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i) acc[i] = f32x16{};
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < ACCS; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < ACCS; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int ACCS>
void run(int blocks, int threads, int n, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chain<ACCS>, dim3(blocks), dim3(threads), 0, 0, d, n, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<ACCS>, dim3(blocks), dim3(threads), 0, 0, d, n, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas_per_wave = double(n) * ACCS;
  const double waves = double(blocks) * threads / 64;
  const double flops = mfmas_per_wave * waves * 32 * 32 * 2 * 2;
  printf("accs=%d blocks=%d threads=%d n=%d: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", ACCS, blocks, threads, n, ms,
         ms * 1e6 / mfmas_per_wave, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(256, 256, 20000, d);    // 1 wave per SIMD
    run<1>(256, 512, 20000, d);    // 2 waves per SIMD
    run<2>(256, 256, 10000, d);
    run<1>(512, 512, 10000, d);    // 4 waves per SIMD (2 blocks per CU)
    run<1>(256, 512, 500, d);      // short kernel: launch + ramp effects
  }
  return 0;
}
