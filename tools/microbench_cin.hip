// CIN forward (Cfg3: F = 39, D = 16, [128,128,128], B = 4096) stand-alone, with per-wave phase stamps:
// how much of a k-step is spent issuing its work and how much waiting in the workgroup barrier.
// Build: make -C tools microbench_cin   (compiles csrc/cin_mfma.hip with -DDFM_CIN_STAMPS)
#define DFM_CIN_STAMPS 1
#include "../deepfm_amd/csrc/cin_mfma.hip"
#include "../deepfm_amd/csrc/runtime.hip"

#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, F = 39, D = 16, L = 3;
  const int Cs[3] = {128, 128, 128}, Hs[3] = {39, 64, 64}, direct[3] = {64, 64, 128}, next_off[3] = {64, 64, 0};
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-0.5f, 0.5f);
  std::vector<float> hx((size_t)B * F * D);
  for (auto& v : hx) v = u(rng);
  float *x0, *out;
  CK(hipMalloc(&x0, hx.size() * 4)); CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)B * 256 * 4));
  dfm::CinMfmaArgs args;
  memset(&args, 0, sizeof(args));
  args.x0 = x0; args.out = out; args.B = B; args.F = F; args.L = L; args.out_dim = 256;
  int hid_rows = 40, col = 0;
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int i = 0; i < L; ++i) {
    const size_t n = (size_t)Cs[i] * Hs[i] * F;
    std::vector<float> hw(n), hb(Cs[i]);
    for (auto& v : hw) v = u(rng) * 0.1f;
    for (auto& v : hb) v = u(rng) * 0.1f;
    float *w, *b; CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&b, Cs[i] * 4));
    CK(hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), Cs[i] * 4, hipMemcpyHostToDevice));
    const size_t pe = dfm::cin_mfma_packed_elems(Hs[i], F, Cs[i]);
    __bf16 *hi, *lo; CK(hipMalloc(&hi, pe * 2)); CK(hipMalloc(&lo, pe * 2));
    if (dfm::cin_mfma_pack(w, Cs[i], Hs[i], F, hi, lo, st)) { fprintf(stderr, "%s\n", dfm::last_error_buf()); return 1; }
    dfm::CinMfmaLayer& ly = args.layer[i];
    ly.w_hi = hi; ly.w_lo = lo; ly.bias = b; ly.Y = nullptr;
    ly.C = Cs[i]; ly.H = Hs[i]; ly.HP = (Hs[i] + 1) / 2; ly.MB = (Cs[i] + 31) / 32;
    ly.direct = direct[i]; ly.next_off = next_off[i]; ly.next_count = i < L - 1 ? Hs[i + 1] : 0; ly.out_col = col;
    col += direct[i];
    hid_rows = std::max(hid_rows, 2 * ly.HP);
  }
  args.hid_rows = hid_rows;
  const int waves = B * D / 32;
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)waves * 8 * 8));
  args.stamps = stamps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) if (dfm::cin_mfma_forward(args, D, true, st)) { fprintf(stderr, "%s\n", dfm::last_error_buf()); return 1; }
  CK(hipStreamSynchronize(st));
  const int iters = 10;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) dfm::cin_mfma_forward(args, D, true, st);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("cin forward (split, stamped build) B=%d: %.1f us per launch\n", B, ms / iters * 1e3);
  std::vector<unsigned long long> h((size_t)waves * 8);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> life, work, barr, pre, epi;
  unsigned long long first = ~0ull, last = 0;
  for (int w = 0; w < waves; ++w) {
    life.push_back((h[w * 8 + 1] - h[w * 8]) * 10.0); work.push_back(h[w * 8 + 2] * 10.0); barr.push_back(h[w * 8 + 3] * 10.0);
    pre.push_back(h[w * 8 + 4] * 10.0); epi.push_back(h[w * 8 + 5] * 10.0);
    first = std::min(first, h[w * 8]); last = std::max(last, h[w * 8 + 1]);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("waves %d: first start -> last end %.1f us; per wave (median): lifetime %.1f us, k-step work issue %.1f us, "
         "barrier wait %.1f us (420 k-steps: %.0f + %.0f ns per k-step)\n", waves, (last - first) * 10.0 / 1e3,
         med(life) / 1e3, med(work) / 1e3, med(barr) / 1e3, med(work) / 420, med(barr) / 420);
  printf("  per wave (median, 3 layers): before the k loops %.1f us, epilogues %.1f us\n", med(pre) / 1e3, med(epi) / 1e3);
  return 0;
}
