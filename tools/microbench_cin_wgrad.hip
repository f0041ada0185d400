// CIN weight gradient (Cfg3 layers: C = 128, H = 39 | 64, F = 39, D = 16, B = 4096) stand-alone: the
// kernel of csrc/cin_mfma_bwd.hip on random operands, timed with HIP events, plus a checksum of dW so
// that two builds can be compared.  Build: make -C tools microbench_cin_wgrad
#include "../deepfm_amd/csrc/cin_mfma_bwd.hip"
#include "../deepfm_amd/csrc/runtime.hip"

#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, F = 39, D = 16, C = 128;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-0.5f, 0.5f);
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int H : {39, 64}) {
    std::vector<float> hx((size_t)B * F * D), hh((size_t)B * H * D), hd((size_t)B * C * D);
    for (auto& v : hx) v = u(rng);
    for (auto& v : hh) v = u(rng);
    for (auto& v : hd) v = u(rng) * 0.01f;
    float *x0, *hid, *dY, *dW, *db; void* ws;
    CK(hipMalloc(&x0, hx.size() * 4)); CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&hid, hh.size() * 4)); CK(hipMemcpy(hid, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dY, hd.size() * 4)); CK(hipMemcpy(dY, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    const size_t nw = (size_t)C * H * F;
    CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&db, C * 4));
    CK(hipMalloc(&ws, dfm::cin_mfma_wgrad_workspace_bytes(B, C, H, F)));
    auto run = [&]() {
      if (dfm::cin_mfma_wgrad(dY, x0, hid, (int64_t)H * D, B, F, H, C, D, dW, db, ws, true, true, st)) { fprintf(stderr, "%s\n", dfm::last_error_buf()); exit(1); }
    };
    CK(hipMemsetAsync(dW, 0, nw * 4, st)); CK(hipMemsetAsync(db, 0, C * 4, st));
    run();
    CK(hipStreamSynchronize(st));
    std::vector<float> w(nw), bb(C);
    CK(hipMemcpy(w.data(), dW, nw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bb.data(), db, C * 4, hipMemcpyDeviceToHost));
    // spot check against a double-precision sum for a few (c, h, f)
    double worst = 0;
    for (int t = 0; t < 6; ++t) {
      const int c = (17 * t + 3) % C, h = (11 * t + 1) % H, f = (7 * t + 2) % F;
      double ref = 0;
      for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
          ref += (double)hd[((size_t)b * C + c) * D + d] * hh[((size_t)b * H + h) * D + d] * hx[((size_t)b * F + f) * D + d];
      worst = std::max(worst, std::abs(ref - w[((size_t)c * H + h) * F + f]) / (std::abs(ref) + 1e-6));
    }
    double bref = 0; for (int b = 0; b < B; ++b) for (int d = 0; d < D; ++d) bref += hd[((size_t)b * C + 5) * D + d];
    double sum = 0; for (float v : w) sum += v;
#ifdef DFM_CIN_STAMPS
    {
      const size_t waves = 4096 * 4;
      unsigned long long* stamps; CK(hipMalloc(&stamps, waves * 8 * 8)); CK(hipMemset(stamps, 0, waves * 8 * 8));
      dfm::g_wgrad_stamps = stamps;
      run(); CK(hipStreamSynchronize(st));
      dfm::g_wgrad_stamps = nullptr;
      std::vector<unsigned long long> hs(waves * 8);
      CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> life, pa, pb, pm, pbar, ns;
      for (size_t w = 0; w < waves; ++w) {
        if (hs[w * 8 + 6] == 0) continue;
        life.push_back((hs[w * 8 + 1] - hs[w * 8]) * 10.0); pa.push_back(hs[w * 8 + 2] * 10.0); pb.push_back(hs[w * 8 + 3] * 10.0);
        pm.push_back(hs[w * 8 + 4] * 10.0); pbar.push_back(hs[w * 8 + 5] * 10.0); ns.push_back((double)hs[w * 8 + 6]);
      }
      auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      const double steps = med(ns);
      printf("  stamps (median wave, %zu waves, %.0f steps): lifetime %.1f us; per step: A phase (LDS reads issued, A(s+1) -> LDS incl. its load wait) %.0f ns, "
             "B phase (B(s) wait + split + next loads issued) %.0f ns, MFMA issue %.0f ns, barrier %.0f ns\n", life.size(), steps,
             med(life) / 1e3, med(pa) / steps, med(pb) / steps, med(pm) / steps, med(pbar) / steps);
    }
#endif
    for (int i = 0; i < 3; ++i) run();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) run();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * B * D * C * (double)H * F * 3;
    printf("wgrad H=%d: %.1f us per call (kernel + slab reduce), %.0f TFLOP/s bf16 incl. the 3x split; checksum %.9g, "
           "worst spot rel err %.2e, db[5] %.6g vs %.6g\n", H, ms / iters * 1e3, flop / (ms / iters * 1e-3) / 1e12, sum, worst,
           bb[5], bref);
  }
  return 0;
}
