// C ABI of the CIN stack (reference deepfm/models/layers/cin.py:26-105): layer bookkeeping,
// saved-activation layout and dispatch between the general fp32 kernels (cin_simple.hip)
// and the MFMA kernels (cin_mfma.hip) for the shapes those support.
#include "common.h"

#include <cstdlib>
#include <vector>

using namespace dfm;

namespace dfm {
int cin_simple_forward_layer(const float* x0, const float* hidden, int64_t hidden_stride,
                             const float* W, const float* bias, int64_t B, int F, int H, int C, int D,
                             int direct, float* Y, float* out, int out_stride, int out_col,
                             hipStream_t st);
int cin_simple_backward_layer(const float* x0, const float* hidden, int64_t hidden_stride,
                              const float* W, const float* Y, int64_t B, int F, int H, int C, int D,
                              int direct, int next_off, int next_count, const float* g_out,
                              int out_stride, int out_col, const float* d_next, float* dY,
                              float* d_hidden_out, float* d_x0, int layer0, float* dW, float* db,
                              float* partial, hipStream_t st);
typedef __bf16 bf16_t;
constexpr int kCinMaxLayers = 8;
struct CinMfmaLayer {
  const bf16_t* w_hi;
  const bf16_t* w_lo;
  const float* bias;
  float* Y;
  uint32_t* mask;
  int y_from;
  int C, H, HP, MB, direct, next_off, next_count, out_col;
};
struct CinMfmaArgs {
  const float* x0;
  float* out;
  int64_t B;
  int F, L, out_dim, hid_rows;
  int pad_;
  unsigned long long* stamps;
  CinMfmaLayer layer[kCinMaxLayers];
};
struct CinBwdLayer {
  const bf16_t* wt_hi;
  const bf16_t* wt_lo;
  const float* Y;
  const uint32_t* mask;
  const float* hidden;
  int64_t hidden_stride;
  float* dY;
  int C, H, HQ, KS, direct, next_off, next_count, out_col;
};
struct CinBwdArgs {
  const float* x0;
  const float* g_out;
  float* g_x0;
  int64_t B;
  int F, L, out_dim, dh_rows;
  CinBwdLayer layer[kCinMaxLayers];
};
size_t cin_bwd_packed_wt_elems(int H, int F, int C);
int cin_bwd_pack_wt(const float* W, int C, int H, int F, bf16_t* hi, bf16_t* lo, hipStream_t st);
int cin_bwd_pack_wt_all(const float* const* W, const int* C, const int* H, int L, int F, bf16_t* hi, bf16_t* lo,
                        const size_t* offs, hipStream_t st);
int cin_mfma_dgrad(const CinBwdArgs& args, int D, bool split, hipStream_t st);
size_t cin_mfma_wgrad_workspace_bytes(int64_t B, int C, int H, int F);
bool cin_mfma_wgrad_has_bias(int F);
bool cin_mfma_wgrad_supported(int64_t B, int D);
int cin_mfma_wgrad(const float* dY, const float* x0, const float* hidden, int64_t hidden_stride, int64_t B,
                   int F, int H, int C, int D, float* dW, float* db, void* workspace, bool split, bool reduce_now,
                   hipStream_t st);
int cin_mfma_wgrad_max_layers_per_reduce();
int cin_mfma_wgrad_reduce_layers(int count, void* const* workspaces, float* const* dW, float* const* db, int64_t B,
                                 int F, const int* H, const int* C, hipStream_t st);
size_t cin_bias_grad_workspace_bytes(int C);
int cin_bias_grad_launch(const float* dY, int64_t B, int C, int D, float* db, float* partial, hipStream_t st);
bool cin_mfma_supported(int F, int D, const int* C, const int* H, int L);
size_t cin_mfma_packed_elems(int H, int F, int C);
int cin_mfma_pack(const float* W, int C, int H, int F, bf16_t* hi, bf16_t* lo, hipStream_t st);
int cin_mfma_pack_all(const float* const* W, const int* C, const int* H, int L, int F, bf16_t* hi, bf16_t* lo,
                      const size_t* offs, hipStream_t st);
int cin_mfma_forward(const CinMfmaArgs& args, int D, bool split, hipStream_t st);
}  // namespace dfm

namespace {
// arithmetic of the MFMA path, set by dfm_cin_set_mode: 0 = bf16 x 3 split (default, parity grade:
// 1e-4), 1 = plain bf16 MFMA (throughput mode with its own, looser tolerance), 2 = general exact-fp32
// kernels only.  An explicit API call, not an environment variable: nothing outside the caller's
// code can change what a training run computes.
int g_cin_mode = 0;
int cin_mode() { return g_cin_mode; }

struct Layout {
  int L = 0, F = 0, D = 0, out_dim = 0;
  std::vector<int> C, H, direct, next_off, out_col;
  std::vector<int64_t> y_off;  // float offset of Y_i inside the saved buffer (per batch of B)
  std::vector<int64_t> mask_off;  // float offset of layer i's ReLU masks (B*D columns x 4 words), behind every Y
  int64_t y_floats = 0;        // the Y_i alone (= the dY workspace of the matrix-core backward)
  int64_t saved_floats = 0;
  int max_C = 0, max_H = 0;
  int64_t max_CK = 0;
};

// direct / next bookkeeping of CIN.__init__ (cin.py:41-64)
int make_layout(const int32_t* sizes, int L, int split_half, int F, int D, int64_t B, Layout* lo) {
  DFM_REQUIRE(sizes && L > 0 && L <= 16, "bad layer list");
  DFM_REQUIRE(F > 0 && D > 0, "bad shape");
  lo->L = L; lo->F = F; lo->D = D;
  int prev = F, col = 0;
  int64_t off = 0;
  for (int i = 0; i < L; ++i) {
    const int c = sizes[i];
    DFM_REQUIRE(c > 0, "layer size must be positive");
    const bool split = split_half && i < L - 1;
    const int direct = split ? c / 2 : c;
    const int next = split ? c - direct : c;
    DFM_REQUIRE(direct >= 0 && next > 0, "layer %d too small to split", i);
    lo->C.push_back(c);
    lo->H.push_back(prev);
    lo->direct.push_back(direct);
    lo->next_off.push_back(split ? direct : 0);
    lo->out_col.push_back(col);
    lo->y_off.push_back(off);
    off += B * c * D;
    col += direct;
    lo->max_C = c > lo->max_C ? c : lo->max_C;
    lo->max_H = prev > lo->max_H ? prev : lo->max_H;
    const int64_t ck = static_cast<int64_t>(c) * prev * F;
    lo->max_CK = ck > lo->max_CK ? ck : lo->max_CK;
    prev = next;
  }
  lo->out_dim = col;
  lo->y_floats = off;
  // ReLU masks of the matrix-core path (round 3): 128 bits per (b, d) column and layer.  When the backward will run
  // on the matrix cores too, cin_fwd_mfma stores these and only the "next" half of every Y (the hidden input of the
  // following layer); cin_dgrad_mfma reads a 16-byte mask per column where it read 64 scalars of Y.
  off = (off + 3) / 4 * 4;                       // 16-byte aligned
  for (int i = 0; i < L; ++i) {
    lo->mask_off.push_back(off);
    off += B * D * 4;
  }
  lo->saved_floats = off;
  return DFM_OK;
}
constexpr int kWgradSlices = 8;
}  // namespace

extern "C" int dfm_cin_set_mode(int mode) {
  DFM_REQUIRE(mode >= 0 && mode <= 2, "CIN mode %d outside [0, 2]", mode);
  g_cin_mode = mode;
  return DFM_OK;
}
extern "C" int dfm_cin_get_mode(void) { return g_cin_mode; }

extern "C" int dfm_cin_output_dim(const int32_t* layer_sizes, int num_layers, int split_half) {
  Layout lo;
  if (make_layout(layer_sizes, num_layers, split_half, 1, 1, 1, &lo)) return -1;
  return lo.out_dim;
}

extern "C" size_t dfm_cin_saved_bytes(const int32_t* layer_sizes, int num_layers, int split_half,
                                      int64_t batch, int num_fields, int dim) {
  Layout lo;
  if (make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return 0;
  return sizeof(float) * static_cast<size_t>(lo.saved_floats);
}

static size_t packed_total_elems(const Layout& lo) {
  size_t total = 0;
  for (int i = 0; i < lo.L; ++i) total += cin_mfma_packed_elems(lo.H[i], lo.F, lo.C[i]);
  return total;
}

extern "C" size_t dfm_cin_forward_workspace_bytes(const int32_t* layer_sizes, int num_layers, int split_half,
                                                  int num_fields, int dim) {
  Layout lo;
  if (make_layout(layer_sizes, num_layers, split_half, num_fields, dim, 1, &lo)) return 0;
  return 2 * sizeof(bf16_t) * packed_total_elems(lo) + 256;   // hi + lo fragment images
}

static size_t packed_wt_total(const Layout& lo) {
  size_t total = 0;
  for (int i = 0; i < lo.L; ++i) total += ((cin_bwd_packed_wt_elems(lo.H[i], lo.F, lo.C[i]) + 63) / 64) * 64;
  return total;
}
static size_t wgrad_region_bytes(int64_t batch, int C, int H, int F) {
  return (cin_mfma_wgrad_workspace_bytes(batch, C, H, F) + 255) / 256 * 256;
}
static bool mfma_bwd_ok(const Layout& lo, int64_t batch) {
  return cin_mode() != 2 && cin_mfma_wgrad_supported(batch, lo.D) &&
         cin_mfma_supported(lo.F, lo.D, lo.C.data(), lo.H.data(), lo.L);
}

extern "C" size_t dfm_cin_backward_workspace_bytes(const int32_t* layer_sizes, int num_layers,
                                                   int split_half, int64_t batch, int num_fields,
                                                   int dim) {
  Layout lo;
  if (make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return 0;
  const int64_t dy = batch * lo.max_C * dim;           // dY of the current layer
  const int64_t dh = batch * lo.max_H * dim;           // d hidden, two buffers (ping-pong)
  int64_t part = kWgradSlices * lo.max_CK;             // weight-gradient partials (reused for bias partials)
  if (part < 256 * static_cast<int64_t>(lo.max_C)) part = 256 * static_cast<int64_t>(lo.max_C);
  size_t simple = sizeof(float) * static_cast<size_t>(dy + 2 * dh + part);
  // matrix-core path: dY of every layer + W^T fragments (hi, lo) + every layer's wgrad workspace (the partial
  // products of all layers are added by one launch at the end)
  size_t wg = 0;
  for (int i = 0; i < lo.L; ++i) wg += wgrad_region_bytes(batch, lo.C[i], lo.H[i], num_fields);
  size_t mfma = sizeof(float) * static_cast<size_t>(lo.y_floats) + 2 * sizeof(bf16_t) * packed_wt_total(lo) + wg + 1024;
  return simple > mfma ? simple : mfma;
}

extern "C" int dfm_cin_forward(const float* d_x0, int64_t batch, int num_fields, int dim,
                               const float* const* weights, const float* const* biases,
                               const int32_t* layer_sizes, int num_layers, int split_half,
                               float* d_out, float* d_saved, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_x0 && weights && biases && d_out, "null argument");
  Layout lo;
  if (int rc = make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return rc;
  if (batch == 0) return DFM_OK;
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < lo.L; ++i) DFM_REQUIRE(weights[i] && biases[i], "layer %d: null parameter", i);

  if (cin_mode() != 2 && d_workspace && cin_mfma_supported(num_fields, dim, lo.C.data(), lo.H.data(), lo.L)) {
    // matrix-core path: repack the weights (they change every step), then one fused launch
    CinMfmaArgs args;
    memset(&args, 0, sizeof(args));
    args.x0 = d_x0; args.out = d_out; args.B = batch; args.F = num_fields; args.L = lo.L;
    args.out_dim = lo.out_dim;
    bf16_t* hi = static_cast<bf16_t*>(d_workspace);
    bf16_t* lop = hi + ((packed_total_elems(lo) + 63) / 64) * 64;
    size_t off = 0;
    int hid_rows = 2;
    size_t pack_offs[kCinMaxLayers] = {};
    for (int i = 0; i < lo.L; ++i) {
      CinMfmaLayer& ly = args.layer[i];
      ly.w_hi = hi + off; ly.w_lo = lop + off; ly.bias = biases[i];
      ly.Y = d_saved ? d_saved + lo.y_off[i] : nullptr;
      // masks + the next half only, when the backward of this (mode, batch, shape) takes the matrix-core kernels
      // (dfm_cin_set_mode must not change between a forward and its backward)
      const bool masks = d_saved && mfma_bwd_ok(lo, batch);
      ly.mask = masks ? reinterpret_cast<uint32_t*>(d_saved + lo.mask_off[i]) : nullptr;
      ly.y_from = masks ? (i < lo.L - 1 ? lo.next_off[i] : lo.C[i]) : 0;
      ly.C = lo.C[i]; ly.H = lo.H[i]; ly.HP = (lo.H[i] + 1) / 2; ly.MB = (lo.C[i] + 31) / 32;
      ly.direct = lo.direct[i]; ly.next_off = lo.next_off[i];
      ly.next_count = i < lo.L - 1 ? lo.H[i + 1] : 0;
      ly.out_col = lo.out_col[i];
      pack_offs[i] = off;
      off += cin_mfma_packed_elems(lo.H[i], num_fields, lo.C[i]);
      hid_rows = 2 * ly.HP > hid_rows ? 2 * ly.HP : hid_rows;
    }
    if (int rc = cin_mfma_pack_all(weights, lo.C.data(), lo.H.data(), lo.L, num_fields, hi, lop, pack_offs, st)) return rc;
    const int fg8 = ((num_fields + 7) / 8) * 8;
    args.hid_rows = hid_rows > fg8 ? hid_rows : fg8;
    return cin_mfma_forward(args, dim, cin_mode() == 0, st);
  }

  DFM_REQUIRE(d_saved, "the general CIN kernels need the d_saved buffer");
  const float* hidden = d_x0;
  int64_t hstride = static_cast<int64_t>(num_fields) * dim;
  for (int i = 0; i < lo.L; ++i) {
    float* Y = d_saved + lo.y_off[i];
    if (int rc = cin_simple_forward_layer(d_x0, hidden, hstride, weights[i], biases[i], batch, num_fields,
                                          lo.H[i], lo.C[i], dim, lo.direct[i], Y, d_out, lo.out_dim,
                                          lo.out_col[i], st))
      return rc;
    hidden = Y + static_cast<int64_t>(lo.next_off[i]) * dim;
    hstride = static_cast<int64_t>(lo.C[i]) * dim;
  }
  return DFM_OK;
}

extern "C" int dfm_cin_backward(const float* d_x0, int64_t batch, int num_fields, int dim,
                                const float* const* weights, const int32_t* layer_sizes,
                                int num_layers, int split_half, const float* d_saved,
                                const float* d_g_out, float* d_g_x0, float* const* g_weights,
                                float* const* g_biases, void* d_workspace, dfm_stream_t stream) {
  DFM_REQUIRE(d_x0 && weights && d_saved && d_g_out && d_g_x0 && g_weights && g_biases && d_workspace,
              "null argument");
  Layout lo;
  if (int rc = make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return rc;
  if (batch == 0) return DFM_OK;
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < lo.L; ++i) DFM_REQUIRE(weights[i] && g_weights[i] && g_biases[i], "layer %d: null parameter", i);
  if (mfma_bwd_ok(lo, batch)) {
    const bool split = cin_mode() == 0;
    float* dY_all = static_cast<float*>(d_workspace);
    bf16_t* hi = reinterpret_cast<bf16_t*>(dY_all + lo.y_floats);
    bf16_t* lop = hi + packed_wt_total(lo);
    unsigned char* wg_ws = reinterpret_cast<unsigned char*>(lop + packed_wt_total(lo));
    wg_ws += (256 - (reinterpret_cast<uintptr_t>(wg_ws) & 255)) & 255;
    CinBwdArgs args;
    memset(&args, 0, sizeof(args));
    args.x0 = d_x0; args.g_out = d_g_out; args.g_x0 = d_g_x0; args.B = batch; args.F = num_fields;
    args.L = lo.L; args.out_dim = lo.out_dim;
    size_t off = 0;
    int dh_rows = 4;
    size_t pack_offs[kCinMaxLayers] = {};
    for (int i = 0; i < lo.L; ++i) {
      CinBwdLayer& ly = args.layer[i];
      ly.wt_hi = hi + off; ly.wt_lo = lop + off;
      ly.Y = d_saved + lo.y_off[i];
      ly.mask = reinterpret_cast<const uint32_t*>(d_saved + lo.mask_off[i]);
      ly.hidden = i == 0 ? d_x0 : d_saved + lo.y_off[i - 1] + static_cast<int64_t>(lo.next_off[i - 1]) * dim;
      ly.hidden_stride = i == 0 ? static_cast<int64_t>(num_fields) * dim : static_cast<int64_t>(lo.C[i - 1]) * dim;
      ly.dY = dY_all + lo.y_off[i];
      ly.C = lo.C[i]; ly.H = lo.H[i]; ly.HQ = (lo.H[i] + 3) / 4; ly.KS = (lo.C[i] + 15) / 16;
      ly.direct = lo.direct[i]; ly.next_off = lo.next_off[i];
      ly.next_count = i < lo.L - 1 ? lo.H[i + 1] : 0;
      ly.out_col = lo.out_col[i];
      pack_offs[i] = off;
      off += ((cin_bwd_packed_wt_elems(lo.H[i], num_fields, lo.C[i]) + 63) / 64) * 64;
      dh_rows = 4 * ly.HQ > dh_rows ? 4 * ly.HQ : dh_rows;
    }
    if (int rc = cin_bwd_pack_wt_all(weights, lo.C.data(), lo.H.data(), lo.L, num_fields, hi, lop, pack_offs, st)) return rc;
    const int fg8 = ((num_fields + 7) / 8) * 8;
    args.dh_rows = dh_rows > fg8 ? dh_rows : fg8;
    if (int rc = cin_mfma_dgrad(args, dim, split, st)) return rc;
    // every layer's batch-sliced partial products into its own region, then ONE launch adds them all
    const bool fused_bias = cin_mfma_wgrad_has_bias(num_fields);   // the bias gradient rides in a padding column
    const bool together = fused_bias && lo.L <= cin_mfma_wgrad_max_layers_per_reduce();
    void* regions[kCinMaxLayers];
    int Hs[kCinMaxLayers], Cs[kCinMaxLayers];
    for (int i = 0; i < lo.L; ++i) {
      const CinBwdLayer& ly = args.layer[i];
      regions[i] = wg_ws;
      Hs[i] = lo.H[i]; Cs[i] = lo.C[i];
      if (int rc = cin_mfma_wgrad(ly.dY, d_x0, ly.hidden, ly.hidden_stride, batch, num_fields, lo.H[i], lo.C[i], dim,
                                  g_weights[i], fused_bias ? g_biases[i] : nullptr, wg_ws, split, !together, st))
        return rc;
      // else: wgrad's slabs are consumed (stream order), its workspace doubles as the bias partials
      if (!fused_bias)
        if (int rc = cin_bias_grad_launch(ly.dY, batch, lo.C[i], dim, g_biases[i], reinterpret_cast<float*>(wg_ws), st)) return rc;
      if (together) wg_ws += wgrad_region_bytes(batch, lo.C[i], lo.H[i], num_fields);
    }
    if (together)
      if (int rc = cin_mfma_wgrad_reduce_layers(lo.L, regions, g_weights, g_biases, batch, num_fields, Hs, Cs, st)) return rc;
    return DFM_OK;
  }
  float* ws = static_cast<float*>(d_workspace);
  float* dY = ws;
  float* dh[2] = {dY + batch * lo.max_C * dim, dY + batch * lo.max_C * dim + batch * lo.max_H * dim};
  float* partial = dh[1] + batch * lo.max_H * dim;   // >= max(slices*C*K, 256*C) floats
  DFM_HIP_TRY(hipMemsetAsync(d_g_x0, 0, sizeof(float) * batch * num_fields * dim, st));
  const float* d_next = nullptr;
  for (int i = lo.L - 1; i >= 0; --i) {
    DFM_REQUIRE(weights[i] && g_weights[i] && g_biases[i], "layer %d: null parameter", i);
    const float* hidden = i == 0 ? d_x0 : d_saved + lo.y_off[i - 1] + static_cast<int64_t>(lo.next_off[i - 1]) * dim;
    const int64_t hstride = i == 0 ? static_cast<int64_t>(num_fields) * dim : static_cast<int64_t>(lo.C[i - 1]) * dim;
    const int next_count = i < lo.L - 1 ? lo.H[i + 1] : 0;
    float* d_hidden_out = dh[i & 1];
    if (int rc = cin_simple_backward_layer(d_x0, hidden, hstride, weights[i], d_saved + lo.y_off[i], batch,
                                           num_fields, lo.H[i], lo.C[i], dim, lo.direct[i], lo.next_off[i],
                                           next_count, d_g_out, lo.out_dim, lo.out_col[i], d_next, dY,
                                           d_hidden_out, d_g_x0, i == 0 ? 1 : 0, g_weights[i], g_biases[i],
                                           partial, st))
      return rc;
    d_next = d_hidden_out;
  }
  return DFM_OK;
}
