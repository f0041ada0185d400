// C ABI of the CIN stack (reference deepfm/models/layers/cin.py:26-105): layer bookkeeping,
// saved-activation layout and dispatch between the general fp32 kernels (cin_simple.hip)
// and the MFMA kernels (cin_mfma.hip) for the shapes those support.
#include "common.h"

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
}  // namespace dfm

namespace {
struct Layout {
  int L = 0, F = 0, D = 0, out_dim = 0;
  std::vector<int> C, H, direct, next_off, out_col;
  std::vector<int64_t> y_off;  // float offset of Y_i inside the saved buffer (per batch of B)
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
  lo->saved_floats = off;
  return DFM_OK;
}
constexpr int kWgradSlices = 8;
}  // namespace

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

extern "C" size_t dfm_cin_backward_workspace_bytes(const int32_t* layer_sizes, int num_layers,
                                                   int split_half, int64_t batch, int num_fields,
                                                   int dim) {
  Layout lo;
  if (make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return 0;
  const int64_t dy = batch * lo.max_C * dim;           // dY of the current layer
  const int64_t dh = batch * lo.max_H * dim;           // d hidden, two buffers (ping-pong)
  const int64_t part = kWgradSlices * lo.max_CK;       // weight-gradient partials
  return sizeof(float) * static_cast<size_t>(dy + 2 * dh + part);
}

extern "C" int dfm_cin_forward(const float* d_x0, int64_t batch, int num_fields, int dim,
                               const float* const* weights, const float* const* biases,
                               const int32_t* layer_sizes, int num_layers, int split_half,
                               float* d_out, float* d_saved, dfm_stream_t stream) {
  DFM_REQUIRE(d_x0 && weights && biases && d_out && d_saved, "null argument");
  Layout lo;
  if (int rc = make_layout(layer_sizes, num_layers, split_half, num_fields, dim, batch, &lo)) return rc;
  if (batch == 0) return DFM_OK;
  hipStream_t st = as_stream(stream);
  const float* hidden = d_x0;
  int64_t hstride = static_cast<int64_t>(num_fields) * dim;
  for (int i = 0; i < lo.L; ++i) {
    DFM_REQUIRE(weights[i] && biases[i], "layer %d: null parameter", i);
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
  float* ws = static_cast<float*>(d_workspace);
  float* dY = ws;
  float* dh[2] = {dY + batch * lo.max_C * dim, dY + batch * lo.max_C * dim + batch * lo.max_H * dim};
  float* partial = dh[1] + batch * lo.max_H * dim;
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
