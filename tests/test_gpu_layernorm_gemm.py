"""GPU checks of the attention block's helper kernels through the C ABI: the residual LayerNorm
(attention.py:117-118; scalar and 16-byte variants) and the many-rows GEMMs of the projections
(attention.py:95-97, :115 and their autograd; csrc/gemm_skinny.hip) against torch in fp64."""
import numpy as np
import pytest
import torch

from deepfm_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D", [10, 12, 32, 64])          # 10: scalar lanes; the others: 16-byte lanes
@pytest.mark.parametrize("rows", [1, 777])
def test_layernorm_forward_backward(D, rows):
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(D * 1000 + rows)
    y = torch.randn(rows, D, device="cuda", generator=g)
    res = torch.randn(rows, D, device="cuda", generator=g)
    gamma = torch.randn(D, device="cuda", generator=g)
    beta = torch.randn(D, device="cuda", generator=g)
    up = torch.randn(rows, D, device="cuda", generator=g)
    out = torch.empty_like(y)
    stats = torch.empty(rows, 2, device="cuda")
    _lib.check(lib.dfm_layernorm_forward(y.data_ptr(), res.data_ptr(), rows, D, gamma.data_ptr(), beta.data_ptr(),
                                         1e-5, out.data_ptr(), stats.data_ptr(), 0, 0, _lib.stream_handle()))
    s = (y + res).double().requires_grad_()
    ga, be = gamma.double().requires_grad_(), beta.double().requires_grad_()
    want = torch.nn.functional.layer_norm(s, (D,), ga, be, 1e-5)
    torch.testing.assert_close(out.double(), want.detach(), rtol=1e-4, atol=1e-5)
    (want * up.double()).sum().backward()
    g_s = torch.empty_like(y)
    d_gamma = torch.zeros(D, device="cuda")
    d_beta = torch.zeros(D, device="cuda")
    ws = torch.empty(max(lib.dfm_layernorm_workspace_bytes(rows, D) // 4, 1), device="cuda")
    _lib.check(lib.dfm_layernorm_backward(up.data_ptr(), y.data_ptr(), res.data_ptr(), stats.data_ptr(), rows, D,
                                          gamma.data_ptr(), g_s.data_ptr(), d_gamma.data_ptr(), d_beta.data_ptr(),
                                          ws.data_ptr(), 0, 0, _lib.stream_handle()))
    torch.testing.assert_close(g_s.double(), s.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(d_gamma.double(), ga.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(d_beta.double(), be.grad, rtol=1e-4, atol=1e-4)


def test_layernorm_grouped_rows_equal_contiguous():
    """Output / incoming gradient addressed as (sample, field) rows inside a wider per-sample layout (the
    attention half of AttentionDeepFM's concatenated DNN input): same bits as the contiguous call."""
    lib = _lib.load()
    B, F, D = 37, 39, 32
    rows, wide = B * F, 2 * F * D
    g = torch.Generator(device="cuda").manual_seed(3)
    y, res, up = (torch.randn(rows, D, device="cuda", generator=g) for _ in range(3))
    gamma, beta = torch.randn(D, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
    outs, grads = [], []
    for grouped in (False, True):
        out = torch.full((B, wide), 7.0, device="cuda") if grouped else torch.empty(rows, D, device="cuda")
        stats = torch.empty(rows, 2, device="cuda")
        _lib.check(lib.dfm_layernorm_forward(y.data_ptr(), res.data_ptr(), rows, D, gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                             out.data_ptr(), stats.data_ptr(), F if grouped else 0, wide if grouped else 0,
                                             _lib.stream_handle()))
        gin = torch.zeros(B, wide, device="cuda")
        gin[:, :F * D] = up.view(B, F * D)
        g_s, dg, db = torch.empty_like(y), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        ws = torch.empty(max(lib.dfm_layernorm_workspace_bytes(rows, D) // 4, 1), device="cuda")
        _lib.check(lib.dfm_layernorm_backward((gin if grouped else up).data_ptr(), y.data_ptr(), res.data_ptr(), stats.data_ptr(),
                                              rows, D, gamma.data_ptr(), g_s.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                              ws.data_ptr(), F if grouped else 0, wide if grouped else 0, _lib.stream_handle()))
        if grouped:
            assert bool((out[:, F * D:] == 7.0).all())               # the other half is left alone
            out = out[:, :F * D].reshape(rows, D)
        outs.append(out.clone()); grads.append((g_s, dg, db))
    assert torch.equal(outs[0], outs[1])
    for a, b in zip(grads[0], grads[1]):
        assert torch.equal(a, b)


def _gemm(a, lda, a_kc, b, ldb, b_kc, c, m, n, k, bias=None, accumulate=False):
    from deepfm_amd.models.layers.dnn import _gemm as gemm
    gemm(a, lda, a_kc, b, ldb, b_kc, c, m, n, k, bias=bias, accumulate=accumulate)


@pytest.mark.parametrize("N,K,w_kc", [(192, 32, True), (32, 64, True), (32, 192, False), (64, 32, False),
                                       (96, 16, True), (64, 64, False)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_many_rows_gemm(N, K, w_kc, accumulate):
    """C (+)= A W^T + bias with 20 011 rows (ragged last tile): the rows kernel of gemm_skinny.hip."""
    M = 20011
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g)
    c = c0.clone()
    W = w if w_kc else w.t().contiguous()
    _gemm(a, K, True, W, K if w_kc else N, w_kc, c, M, N, K, bias=bias, accumulate=accumulate)
    want = a.double() @ w.double().t() + bias.double() + (c0.double() if accumulate else 0.0)
    torch.testing.assert_close(c.double(), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("N1,N2", [(192, 32), (32, 64), (64, 32), (32, 32), (96, 32)])
def test_weight_and_bias_gradient(N1, N2):
    """dW = g^T x, db = column sums of g over 20 011 rows (dfm_weight_grad_f32), twice: bitwise equal."""
    lib = _lib.load()
    M = 20011
    gen = torch.Generator(device="cuda").manual_seed(N1 + N2)
    g = torch.randn(M, N1, device="cuda", generator=gen)
    x = torch.randn(M, N2, device="cuda", generator=gen)
    ws_bytes = lib.dfm_weight_grad_workspace_bytes(M, N1, N2)
    assert ws_bytes > 0
    outs = []
    for _ in range(2):
        ws = torch.empty(ws_bytes // 4, device="cuda")
        dw = torch.empty(N1, N2, device="cuda")
        db = torch.empty(N1, device="cuda")
        _lib.check(lib.dfm_weight_grad_f32(g.data_ptr(), N1, x.data_ptr(), N2, M, N1, N2, dw.data_ptr(), N2,
                                           db.data_ptr(), 0, ws.data_ptr(), _lib.stream_handle()))
        outs.append((dw.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    torch.testing.assert_close(outs[0][0].double(), g.double().t() @ x.double(), rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(outs[0][1].double(), g.double().sum(0), rtol=1e-4, atol=2e-3)
    # the same product through dfm_gemm_f32's dispatch (both operands K-strided), accumulating
    c = torch.ones(N1, N2, device="cuda")
    _gemm(g, N1, False, x, N2, False, c, N1, N2, M, accumulate=True)
    torch.testing.assert_close(c.double(), 1.0 + g.double().t() @ x.double(), rtol=1e-4, atol=2e-3)


def test_weight_grad_unsupported_shape_fails_loudly():
    lib = _lib.load()
    assert lib.dfm_weight_grad_workspace_bytes(100, 32, 32) == 0          # too few rows
    assert lib.dfm_weight_grad_workspace_bytes(20000, 48, 32) == 0        # not a multiple of 32
    t = torch.zeros(64, device="cuda")
    rc = lib.dfm_weight_grad_f32(t.data_ptr(), 32, t.data_ptr(), 32, 100, 32, 32, t.data_ptr(), 32, None, 0,
                                 t.data_ptr(), _lib.stream_handle())
    assert rc != 0
