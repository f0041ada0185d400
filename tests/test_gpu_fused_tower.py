"""GPU: the fused DNN-tower kernels (csrc/tower.hip) against torch fp64 restatements of the
reference modules (dnn.py:45-55 Linear/BatchNorm1d/ReLU/Dropout, deepfm.py:30-42 head,
trainer.py:59 BCEWithLogitsLoss, fm.py:18-23 backward), and the fused DeepFM step against the
oracle's training step and against the autograd step."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, npy
from tests.test_gpu_models_step import _oracle_state, _pool, _small_deepfm

pytestmark = pytest.mark.gpu


def _ws(nbytes):
    return torch.zeros(max((nbytes + 3) // 4, 1), dtype=torch.int32, device="cuda")


def _bn_ctx(z, stats, gamma, beta, dy, g_gamma, g_beta, ws, p=0.0, seed=None, salt=0):
    from deepfm_amd import _lib
    c = _lib.BnBwd()
    c.z, c.mean_rstd, c.gamma, c.beta = z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    c.dy, c.g_gamma, c.g_beta = dy.data_ptr(), g_gamma.data_ptr(), g_beta.data_ptr()
    c.seed = seed.data_ptr() if seed is not None else None
    c.workspace, c.p_drop, c.salt = ws.data_ptr(), p, salt
    return c


@pytest.mark.parametrize("shape", [(4096, 256, 624), (4099, 40, 52), (37, 72, 12), (2, 4, 8), (700, 128, 256)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_linear_bn_forward_and_apply_match_torch(shape, p):
    from deepfm_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g) * 1.5 + 0.3
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g) * 30.0        # column means far from zero (|mean| >> std)
    gamma = torch.rand(N, device="cuda", generator=g) + 0.5
    beta = torch.randn(N, device="cuda", generator=g) * 0.3
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    nb = torch.zeros(1, dtype=torch.int64, device="cuda")
    z, out = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    stats = torch.empty(2, N, device="cuda")
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    ws = _ws(lib.dfm_linear_bn_workspace_bytes(M, N))
    for _ in range(2):
        _lib.check(lib.dfm_linear_bn_forward(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), M, N, K, z.data_ptr(),
                                             ws.data_ptr(), _lib.stream_handle()))
        _lib.check(lib.dfm_bn_relu_dropout_apply(z.data_ptr(), M, N, ws.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 stats.data_ptr(), rm.data_ptr(), rv.data_ptr(), nb.data_ptr(), 0.1, 1e-5,
                                                 p, seed.data_ptr(), 1, out.data_ptr(), _lib.stream_handle()))
    zd = x.double() @ w.double().t() + b.double()
    assert_close(npy(z), npy(zd), rtol=1e-5, what="z")
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    assert_close(npy(stats[0]), npy(mean), rtol=1e-5, what="mean")
    # |mean|/std ~ 30: fp32 z itself carries ~30 * 6e-8 relative noise per element
    assert_close(npy(stats[1]), npy((var + 1e-5).rsqrt()), rtol=1e-4, what="rstd")
    assert int(nb) == 2
    unb = var * M / max(M - 1, 1)
    assert_close(npy(rm), npy(0.9 * (0.1 * mean) + 0.1 * mean), rtol=1e-5, what="running_mean")
    assert_close(npy(rv), npy(0.9 * (0.9 + 0.1 * unb) + 0.1 * unb), rtol=1e-4, what="running_var")
    # the normalised activations against torch's own BatchNorm on the same z (fp32)
    want = torch.relu(torch.nn.functional.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5))
    if p == 0.0:
        assert_close(npy(out), npy(want), rtol=1e-4, atol_scale=2e-5, what="relu(bn(z))")
    else:
        kept = out != 0
        on = want > 1e-3
        assert abs(float(kept[on].float().mean()) - (1 - p)) < 0.03 + 2.0 / (on.sum().item() ** 0.5 + 1)
        assert_close(npy(out[kept]), npy(want[kept] / (1 - p)), rtol=1e-4, atol_scale=2e-5, what="kept values / (1-p)")


def _torch_tower_ref(x, lins, bns, head_w, head_b, fo, fm, labels, e=None, fm_from_e=False):
    """fp64 autograd restatement: [Linear, BatchNorm1d(train), ReLU] * n -> head -> BCE."""
    h = x
    zs = []
    for (w, b), (ga, be) in zip(lins, bns):
        z = h @ w.t() + b
        mu, var = z.mean(0), z.var(0, unbiased=False)
        h = torch.relu(ga * (z - mu) * (var + 1e-5).rsqrt() + be)
        zs.append(z)
    logits = (fo + fm) + (h @ head_w.t() + head_b).view(-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels)
    return logits, loss


@pytest.mark.parametrize("M,K", [(4096, 64), (777, 32), (33, 256)])
def test_head_bce_matches_torch(M, K):
    from deepfm_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    z = torch.randn(M, K, device="cuda", generator=g)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g) * 0.3
    w = torch.randn(1, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(1, device="cuda", generator=g)
    fo, fm = torch.randn(M, device="cuda", generator=g), torch.randn(M, device="cuda", generator=g)
    y = (torch.rand(M, device="cuda", generator=g) < 0.3).float()
    # reference in fp64 with autograd
    zd = z.double().requires_grad_()
    gd, bd, wd, b_d = (t.double().requires_grad_() for t in (gamma, beta, w, b))
    mu, var = zd.mean(0), zd.var(0, unbiased=False)
    stats = torch.stack([mu.detach(), (var.detach() + 1e-5).rsqrt()]).float().contiguous()
    xhat = (zd - mu.detach()) * (var.detach() + 1e-5).rsqrt()          # statistics held fixed: the kernel's half
    yv = gd * xhat + bd
    a = torch.relu(yv)
    logits = (fo.double() + fm.double()) + (a @ wd.t() + b_d).view(-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y.double())
    dy_ref, = torch.autograd.grad(loss, yv, retain_graph=True)
    loss.backward()
    # kernel: head (mask + partial sums) then the BatchNorm backward apply that finishes it
    a32 = a.detach().float().contiguous()
    out = {k: torch.zeros(sh, device="cuda") for k, sh in dict(logits=M, loss=1, dl=M, gw=K, gb=1, dy=(M, K), dz=(M, K),
                                                                 gg=K, gbeta=K).items()}
    wsb = _ws(lib.dfm_bn_bwd_workspace_bytes(M, K))
    ctx = _bn_ctx(z, stats, gamma, beta, out["dy"], out["gg"], out["gbeta"], wsb)
    tail = _lib.HeadTail()
    tail.g_w, tail.g_b, tail.loss = out["gw"].data_ptr(), out["gb"].data_ptr(), out["loss"].data_ptr()
    for rep in range(2):
        for k in ("gw", "gb", "gg", "gbeta"):
            out[k].zero_()
        _lib.check(lib.dfm_head_bce(a32.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), fo.data_ptr(), fm.data_ptr(),
                                    y.data_ptr(), out["logits"].data_ptr(), out["dl"].data_ptr(), C.byref(ctx),
                                    _lib.stream_handle()))
        dy_got = out["dy"].clone()
        _lib.check(lib.dfm_bn_backward_apply(C.byref(ctx), M, K, C.byref(tail), out["dz"].data_ptr(), _lib.stream_handle()))
    assert_close(npy(out["logits"]), npy(logits), rtol=1e-5, what="logits")
    assert abs(float(out["loss"]) - float(loss)) < 1e-6
    dl_ref = (torch.sigmoid(logits) - y.double()) / M
    assert_close(npy(out["dl"]), npy(dl_ref), rtol=1e-5, what="d logits")
    assert_close(npy(out["gw"]), npy(wd.grad.view(-1)), rtol=1e-4, what="d head weight")
    assert_close(npy(out["gb"]), npy(b_d.grad), rtol=1e-4, what="d head bias")
    assert_close(npy(dy_got), npy(dy_ref), rtol=1e-5, what="dy")
    assert_close(npy(out["gbeta"]), npy(bd.grad), rtol=1e-4, what="d beta")
    assert_close(npy(out["gg"]), npy(gd.grad), rtol=1e-4, what="d gamma")
    xh = xhat.detach()
    want_dz = gd.detach() * stats[1].double() * (dy_ref - dy_ref.mean(0) - xh * (dy_ref * xh).mean(0))
    assert_close(npy(out["dz"]), npy(want_dz), rtol=1e-4, what="dz")


@pytest.mark.parametrize("shape", [(4096, 128, 256), (4096, 256, 624), (1000, 36, 44), (65, 8, 12), (50, 6, 10)])
@pytest.mark.parametrize("epi", ["plain", "bn", "fm"])
def test_linear_backward_matches_torch(shape, epi):
    from deepfm_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    if epi == "fm" and K % 4:
        pytest.skip("fm epilogue needs K = fields * dim")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dz = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    gw = torch.ones(N, K, device="cuda")
    ws = _ws(lib.dfm_linear_backward_workspace_bytes(M, N, K))
    gx_ref = dz.double() @ w.double()
    gw_ref = dz.double().t() @ x.double() + 1.0
    gx = torch.zeros(M, K, device="cuda")
    if epi == "plain":
        _lib.check(lib.dfm_linear_backward(dz.data_ptr(), M, N, x.data_ptr(), K, w.data_ptr(), gx.data_ptr(),
                                           None, None, 2, ws.data_ptr(), _lib.stream_handle()))    # d input, then d weight: two launches
        _lib.check(lib.dfm_linear_backward(dz.data_ptr(), M, N, x.data_ptr(), K, w.data_ptr(), gx.data_ptr(),
                                           None, None, 1, ws.data_ptr(), _lib.stream_handle()))
        assert_close(npy(gx), npy(gx_ref), rtol=1e-5, atol_scale=2e-6, what="dx")
    elif epi == "fm":
        D = 4
        gfm = torch.randn(M, device="cuda", generator=g)
        S = torch.randn(M, D, device="cuda", generator=g)
        fmb = _lib.FmBwd()
        fmb.g_fm, fmb.fm_sum, fmb.e, fmb.dim = gfm.data_ptr(), S.data_ptr(), x.data_ptr(), D
        _lib.check(lib.dfm_linear_backward(dz.data_ptr(), M, N, x.data_ptr(), K, w.data_ptr(), gx.data_ptr(),
                                           None, C.byref(fmb), 3, ws.data_ptr(), _lib.stream_handle()))
        want = gx_ref + gfm.double()[:, None] * (S.double().repeat(1, K // D) - x.double())
        assert_close(npy(gx), npy(want), rtol=1e-5, atol_scale=2e-6, what="dx + fm backward")
    else:
        z = torch.randn(M, K, device="cuda", generator=g)
        gamma = torch.rand(K, device="cuda", generator=g) + 0.5
        beta = torch.randn(K, device="cuda", generator=g) * 0.3
        mu, var = z.double().mean(0), z.double().var(0, unbiased=False)
        stats = torch.stack([mu, (var + 1e-5).rsqrt()]).float().contiguous()
        xhat = (z.double() - mu) * (var + 1e-5).rsqrt()
        mask = (gamma.double() * xhat + beta.double()) > 0
        dy_ref = gx_ref * mask
        dy = torch.zeros(M, K, device="cuda")
        gg, gb = torch.ones(K, device="cuda"), torch.ones(K, device="cuda")
        wsb = _ws(lib.dfm_bn_bwd_workspace_bytes(M, K))
        ctx = _bn_ctx(z, stats, gamma, beta, dy, gg, gb, wsb)
        _lib.check(lib.dfm_linear_backward(dz.data_ptr(), M, N, x.data_ptr(), K, w.data_ptr(), None,
                                           C.byref(ctx), None, 3, ws.data_ptr(), _lib.stream_handle()))
        # elements whose y sits within rounding of the ReLU kink may fall on either side
        near = (gamma.double() * xhat + beta.double()).abs() < 1e-5
        got = npy(dy).copy()
        got[npy(near)] = npy(dy_ref)[npy(near)]
        assert_close(got, npy(dy_ref), rtol=1e-5, atol_scale=2e-6, what="dy")
        # second half of the BatchNorm backward (merges the dx epilogue's per-tile sums)
        if K % 4 == 0:
            dzl = torch.empty(M, K, device="cuda")
            _lib.check(lib.dfm_bn_backward_apply(C.byref(ctx), M, K, None, dzl.data_ptr(), _lib.stream_handle()))
            dyd = dy.double()
            want = gamma.double() * stats[1].double() * (dyd - dyd.mean(0) - xhat * (dyd * xhat).mean(0))
            assert_close(npy(dzl), npy(want), rtol=1e-4, atol_scale=2e-5, what="dz")
            assert_close(npy(gb), npy(dyd.sum(0) + 1), rtol=1e-4, what="d beta")
            assert_close(npy(gg), npy((dyd * xhat).sum(0) + 1), rtol=1e-4, what="d gamma")
    if (N * K) % 4 == 0:
        ref = _lib.SlabRef()
        ref.workspace, ref.g_w, ref.batch, ref.out_features, ref.in_features = ws.data_ptr(), gw.data_ptr(), M, N, K
        _lib.check(lib.dfm_linear_backward_finish(C.byref(ref), 1, _lib.stream_handle()))
        assert_close(npy(gw), npy(gw_ref), rtol=1e-5, atol_scale=2e-6, what="dW (accumulated)")


def test_apply_and_mask_share_the_dropout_pattern():
    """dfm_bn_relu_dropout_apply and the backward mask (dfm_head_bce / dfm_linear_backward) must
    rebuild the same keep pattern from (seed, salt, element index)."""
    from deepfm_amd import _lib
    lib = _lib.load()
    M, K, p = 2048, 64, 0.3
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, 16, device="cuda", generator=g)
    w = torch.randn(K, 16, device="cuda", generator=g)
    gamma, beta = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    z, a, stats = torch.empty(M, K, device="cuda"), torch.empty(M, K, device="cuda"), torch.empty(2, K, device="cuda")
    seed = torch.tensor([1234567], dtype=torch.int64, device="cuda")
    wsf = _ws(lib.dfm_linear_bn_workspace_bytes(M, K))
    _lib.check(lib.dfm_linear_bn_forward(x.data_ptr(), 16, w.data_ptr(), None, M, K, 16, z.data_ptr(), wsf.data_ptr(),
                                         _lib.stream_handle()))
    _lib.check(lib.dfm_bn_relu_dropout_apply(z.data_ptr(), M, K, wsf.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             stats.data_ptr(), None, None, None, 0.1, 1e-5, p, seed.data_ptr(), 3,
                                             a.data_ptr(), _lib.stream_handle()))
    xhat = (z - stats[0]) * stats[1]
    kept = (a != 0)
    assert abs(float(kept[xhat > 0].float().mean()) - (1 - p)) < 0.02
    assert_close(npy(a[kept]), npy(xhat[kept] / (1 - p)), rtol=1e-5, what="kept values scaled by 1/(1-p)")
    hw = torch.ones(1, K, device="cuda")
    y = torch.zeros(M, device="cuda")
    logits, dl = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
    dy, gg, gb = torch.zeros(M, K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    wsb = _ws(lib.dfm_bn_bwd_workspace_bytes(M, K))
    ctx = _bn_ctx(z, stats, gamma, beta, dy, gg, gb, wsb, p=p, seed=seed, salt=3)
    _lib.check(lib.dfm_head_bce(a.data_ptr(), M, K, hw.data_ptr(), None, None, None, y.data_ptr(), logits.data_ptr(),
                                dl.data_ptr(), C.byref(ctx), _lib.stream_handle()))
    torch.cuda.synchronize()
    assert torch.equal(dy != 0, kept)        # d logits > 0 everywhere (labels 0), so dy != 0 exactly where kept


@pytest.mark.parametrize("M,K,p", [(2048, 64, 0.3), (4096, 64, 0.0), (1000, 32, 0.2), (777, 128, 0.5), (4096, 256, 0.1)])
def test_head_with_the_last_batchnorm_inside_is_bitwise_the_two_launches(M, K, p):
    """dfm_head_bn_bce = dfm_bn_relu_dropout_apply (last block) + dfm_head_bce: same statistics (mean, rstd,
    running mean / variance, batch counter), logits, d logits, masked gradient and workgroup partials, bit for
    bit — every workgroup of the fused launch merges the tile statistics in the order the apply kernel uses."""
    from deepfm_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, 48, device="cuda", generator=g)
    w = torch.randn(K, 48, device="cuda", generator=g)
    bias = torch.randn(K, device="cuda", generator=g)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g) * 0.1
    hw, hb = torch.randn(1, K, device="cuda", generator=g) * 0.1, torch.randn(1, device="cuda", generator=g)
    fo, fm = torch.randn(M, device="cuda", generator=g), torch.randn(M, device="cuda", generator=g)
    y = (torch.rand(M, device="cuda", generator=g) < 0.3).float()
    seed = torch.tensor([987654321], dtype=torch.int64, device="cuda")
    z = torch.empty(M, K, device="cuda")
    wsf = _ws(lib.dfm_linear_bn_workspace_bytes(M, K))
    _lib.check(lib.dfm_linear_bn_forward(x.data_ptr(), 48, w.data_ptr(), bias.data_ptr(), M, K, 48, z.data_ptr(),
                                         wsf.data_ptr(), _lib.stream_handle()))
    nws = lib.dfm_bn_bwd_workspace_bytes(M, K) // 4

    def run(fused):
        stats = torch.zeros(2, K, device="cuda")
        rm, rv = torch.full((K,), 0.25, device="cuda"), torch.full((K,), 2.0, device="cuda")
        nb = torch.tensor([7], dtype=torch.int64, device="cuda")
        logits, dl = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
        dy, gg, gb = torch.zeros(M, K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
        wsb = torch.zeros(nws, device="cuda")
        ctx = _bn_ctx(z, stats, gamma, beta, dy, gg, gb, wsb, p=p, seed=seed, salt=2)
        if fused:
            _lib.check(lib.dfm_head_bn_bce(wsf.data_ptr(), stats.data_ptr(), rm.data_ptr(), rv.data_ptr(), nb.data_ptr(),
                                           0.1, 1e-5, M, K, hw.data_ptr(), hb.data_ptr(), fo.data_ptr(), fm.data_ptr(),
                                           y.data_ptr(), logits.data_ptr(), dl.data_ptr(), C.byref(ctx),
                                           _lib.stream_handle()))
        else:
            a = torch.empty(M, K, device="cuda")
            _lib.check(lib.dfm_bn_relu_dropout_apply(z.data_ptr(), M, K, wsf.data_ptr(), gamma.data_ptr(),
                                                     beta.data_ptr(), stats.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                     nb.data_ptr(), 0.1, 1e-5, p, seed.data_ptr(), 2, a.data_ptr(),
                                                     _lib.stream_handle()))
            _lib.check(lib.dfm_head_bce(a.data_ptr(), M, K, hw.data_ptr(), hb.data_ptr(), fo.data_ptr(), fm.data_ptr(),
                                        y.data_ptr(), logits.data_ptr(), dl.data_ptr(), C.byref(ctx),
                                        _lib.stream_handle()))
        torch.cuda.synchronize()
        return dict(stats=stats, rm=rm, rv=rv, nb=nb, logits=logits, dl=dl, dy=dy, partials=wsb)

    two, one = run(False), run(True)
    assert int(one["nb"]) == 8
    for k in two:
        assert torch.equal(two[k], one[k]), k


@pytest.mark.parametrize("M,K", [(4096, 256), (1000, 64), (77, 200), (4099, 1024), (5, 4)])
def test_one_output_linear_kernels_match_torch(M, K):
    """dfm_linear1_forward / _backward (xDeepFM's cin_linear, xdeepfm.py:41-47) against torch in float64; the weight
    gradient arrives as slabs and is added by dfm_linear_backward_finish."""
    from deepfm_amd import _lib
    lib = _lib.load()
    assert lib.dfm_linear1_supported(K) == 1 and lib.dfm_linear1_supported(K + 1) == 0 and lib.dfm_linear1_supported(2048) == 0
    g = torch.Generator(device="cuda").manual_seed(M * 7 + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(1, K, device="cuda", generator=g)
    b = torch.randn(1, device="cuda", generator=g)
    go = torch.randn(M, device="cuda", generator=g)
    out = torch.empty(M, device="cuda")
    _lib.check(lib.dfm_linear1_forward(x.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), out.data_ptr(), _lib.stream_handle()))
    want = (x.double() @ w.double().t()).view(-1) + b.double()
    assert_close(npy(out), npy(want), rtol=1e-5, atol_scale=2e-6, what="forward")
    _lib.check(lib.dfm_linear1_forward(x.data_ptr(), M, K, w.data_ptr(), None, out.data_ptr(), _lib.stream_handle()))
    assert_close(npy(out), npy(want - b.double()), rtol=1e-5, atol_scale=2e-6, what="forward without bias")
    splits = lib.dfm_linear1_backward_splits(M)
    ws = torch.full((splits, K), float("nan"), device="cuda")
    gx = torch.empty(M, K, device="cuda")
    _lib.check(lib.dfm_linear1_backward(go.data_ptr(), x.data_ptr(), M, K, w.data_ptr(), gx.data_ptr(), ws.data_ptr(),
                                        _lib.stream_handle()))
    assert torch.equal(gx, go.view(-1, 1) * w)
    gw = torch.ones(1, K, device="cuda")
    ref = _lib.SlabRef()
    ref.workspace, ref.g_w, ref.batch, ref.out_features, ref.in_features, ref.splits = ws.data_ptr(), gw.data_ptr(), 1, 1, K, splits
    _lib.check(lib.dfm_linear_backward_finish(C.byref(ref), 1, _lib.stream_handle()))
    assert_close(npy(gw), npy(go.double().view(1, -1) @ x.double() + 1), rtol=1e-5, atol_scale=2e-6, what="dW (accumulated)")


def _fused_pair(B, seed=0):
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    fields, cfg, model = _small_deepfm(seed=seed)
    hp = dict(lr=1e-3, l2=1e-5, max_grad_norm=1.0)
    opt = RowSparseAdam(model, lr=hp["lr"], l2=hp["l2"], max_grad_norm=hp["max_grad_norm"])
    return fields, cfg, model, hp, opt, FusedDeepFMStep


@pytest.mark.parametrize("B", [512, 6000])
def test_fused_deepfm_steps_vs_oracle(B):
    fields, cfg, model, hp, opt, Step = _fused_pair(B)
    params, state = _oracle_state(model)
    step = Step(model, opt, B, use_graph=False)
    rng = np.random.default_rng(3)
    ids, dense, labels = _pool(fields, 3, B, rng)
    ocfg = dict(fm_dim=16, hidden_units=cfg.dnn.hidden_units)
    for i in range(3):
        step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
        step.run()
        batch = {f["name"]: ids[i, j] for j, f in enumerate(fields[:26])}
        batch.update({f["name"]: dense[i, j] for j, f in enumerate(fields[26:])})
        oloss = O.deepfm_train_step_rowsparse(fields, params, state, batch, labels[i], ocfg, hp, i + 1, exact_order=(B <= 512))
        assert abs(float(step.loss) - float(oloss)) < 2e-5 + 1e-4 * abs(float(oloss)), (i, float(step.loss), float(oloss))
    got = {k: npy(v) for k, v in model.state_dict().items()}
    for k, want in params.items():
        if "running_" in k:
            continue
        if k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0:
            continue        # zero-gradient parameter (Linear bias in front of BatchNorm)
        assert_close(got[k], want, rtol=1e-4, atol_scale=0.0, floor=1e-4, what=k)


def test_fused_step_graph_replay_is_bitwise_equal_to_eager():
    from tools_shared import criteo_fields
    B = 1024
    results = []
    rng = np.random.default_rng(9)
    fields = criteo_fields(300, 16)
    ids, dense, labels = _pool(fields, 4, B, rng)
    for use_graph in (False, True, True):
        _, _, model, hp, opt, Step = _fused_pair(B, seed=4)
        step = Step(model, opt, B, use_graph=use_graph)
        start = copy.deepcopy(model.state_dict())
        step.load_batch(torch.from_numpy(ids[0]).cuda(), torch.from_numpy(dense[0]).cuda(), torch.from_numpy(labels[0]).cuda())
        step.capture()
        for k, v in model.state_dict().items():
            assert torch.equal(v, start[k]), f"capture() changed {k}"
        assert int(opt.step_count) == 0 and not opt.flat_m.any()
        for i in range(4):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
        torch.cuda.synchronize()
        results.append({k: npy(v).copy() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert np.array_equal(results[1][k], results[2][k]), f"graph replay not deterministic: {k}"
        assert np.array_equal(results[0][k], results[1][k]), f"graph != eager: {k}"


def test_graph_node_updates_and_eager_gather_variant_are_bitwise_equal_to_eager():
    """The step as ONE graph whose gather node is re-pointed at a new record before every launch
    (two alternating execs), mixed with the timed variant (gather launched eagerly in front of the
    gather-less copy of the graph), from device records at changing addresses: bit-identical to the
    eager step.  7 steps: both execs are updated several times and the host runs ahead of the device."""
    from tools_shared import criteo_fields
    B, n = 778, 7                                            # odd batch: ragged tail workgroup
    rng = np.random.default_rng(19)
    fields = criteo_fields(300, 16)
    ids, dense, labels = _pool(fields, n, B, rng)
    results = []
    for mode in ("eager", "graph", "mixed", "groups", "handoff"):
        _, _, model, hp, opt, Step = _fused_pair(B, seed=8)
        step = Step(model, opt, B, use_graph=mode != "eager")
        recs = step.pack_batches(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        step.capture(timed_variant=True, steps_per_graph=3 if mode in ("groups", "handoff") else 1)
        losses = []
        if mode == "handoff":
            # round 3: a launch's LAST optimizer kernel builds the row plan of the record the next launch starts with
            # (run_group(next_record=...)): the next launch — a group in its plan-less "continuation" flavour, or a single
            # step whose eager part is the gather alone — must give the same bits; a wrong announcement (a launch that
            # starts with another record than the one planned for) falls back to the stand-alone plan
            assert step.cont_slots and len(step._plan_sets) == 3
            step.run_group([recs[0], recs[1], recs[2]], next_record=recs[3])
            losses.append(step.loss.clone())
            step.run_from(recs[3], eager_gather=True)          # starts from the hand-off plan
            losses.append(step.loss.clone())
            # the same launch in its two halves (bench.py prepares the timed region's first launch ahead of the opening
            # synchronisation): host half, an unrelated synchronisation, then the launch
            step.prepare_group([recs[4], recs[5], recs[6]], next_record=recs[4])
            with pytest.raises(RuntimeError):
                step.prepare_group([recs[4], recs[5], recs[6]])          # one pending launch at a time
            with pytest.raises(RuntimeError):
                step.run_group([recs[4], recs[5], recs[6]])
            torch.cuda.synchronize()
            step.launch_prepared()
            with pytest.raises(RuntimeError):
                step.launch_prepared()
            losses.append(step.loss.clone())
            snap7 = {k: npy(v).copy() for k, v in model.state_dict().items()}
            # continuation flavour twice in a row, then an announcement that is not honoured
            step.run_group([recs[4], recs[5], recs[6]], next_record=recs[0])
            step.run_group([recs[0], recs[1], recs[2]], next_record=recs[5])
            step.run_group([recs[4], recs[5], recs[6]])
            torch.cuda.synchronize()
        if mode == "groups":
            # three steps per graph launch (run_group), a single eager-gather step in between
            assert step.steps_per_graph == 3
            with pytest.raises(RuntimeError):
                step.run_from(recs[0])
            step.run_group([recs[0], recs[1], recs[2]])
            losses.append(step.loss.clone())
            step.run_from(recs[3], eager_gather=True)
            losses.append(step.loss.clone())
            step.run_group([recs[4], recs[5], recs[6]])
            losses.append(step.loss.clone())
        if mode == "groups":       # the same extra launches as the hand-off run, every one with its own stand-alone plan
            snap7 = {k: npy(v).copy() for k, v in model.state_dict().items()}
            for grp in ([4, 5, 6], [0, 1, 2], [4, 5, 6]):
                step.run_group([recs[j] for j in grp])
        for i in range(n if mode not in ("groups", "handoff") else 0):
            if mode == "mixed" and i % 3 == 1:
                step.run_from(recs[i], eager_gather=True)
            else:
                step.run_from(recs[i])
            losses.append(step.loss.clone())
        torch.cuda.synchronize()
        assert torch.equal(step.ids, torch.from_numpy(ids[n - 1]).cuda())      # static inputs: the last batch
        results.append(({k: npy(v).copy() for k, v in model.state_dict().items()}, [float(l) for l in losses],
                        snap7 if mode in ("groups", "handoff") else None))
    ref_losses = results[0][1]
    for mode, other in zip(("graph", "mixed", "groups", "handoff"), results[1:]):
        want = ref_losses if mode in ("graph", "mixed") else [ref_losses[2], ref_losses[3], ref_losses[6]]
        assert other[1] == want, f"{mode}: losses differ"
        after7 = other[0] if mode in ("graph", "mixed") else other[2]             # groups / handoff go on for three more launches
        for k in results[0][0]:
            assert np.array_equal(after7[k], results[0][0][k]), (mode, k)
    for k in results[3][0]:                                                        # ... and still agree with each other
        assert np.array_equal(results[4][0][k], results[3][0][k]), ("handoff vs groups", k)


def test_fused_step_rejects_ineligible_models():
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from tests.helpers import schema_from_fields
    from tools_shared import criteo_fields
    cfg = ExperimentConfig()
    cfg.dnn.use_batch_norm = False
    model = create_model("deepfm", schema_from_fields(criteo_fields(50, 16)), cfg).cuda().train()
    model.embedding.set_grad_mode("rowsparse")
    assert not FusedDeepFMStep.eligible(model)


# ------------------------------------------------------------------ xDeepFM on the fused tower
def _small_xdeepfm(seed=0, V=300):
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from tests.helpers import schema_from_fields
    from tools_shared import criteo_fields
    fields = criteo_fields(V, 16)
    cfg = ExperimentConfig()
    cfg.dnn.hidden_units, cfg.dnn.dropout = [64, 32], 0.0
    cfg.cin.layer_sizes, cfg.cin.split_half = [32, 24, 16], True
    torch.manual_seed(seed)
    model = create_model("xdeepfm", schema_from_fields(fields), cfg).cuda().train()
    model.embedding.set_grad_mode("rowsparse")
    return fields, cfg, model


@pytest.mark.parametrize("B", [512, 1000])
def test_fused_xdeepfm_step_matches_the_autograd_step(B):
    """FusedXDeepFMStep (CIN + heads called directly, CIN's d embeddings added in the first Linear's
    epilogue) against RowSparseTrainStep over autograd — whose forward and gradients are pinned by the
    reference golden ``model_xdeepfm``.  After ONE step from equal state the Adam moments (linear in the
    gradients: 0.1 * clip * g) of every dense parameter and of the tables must agree to 1e-4; then the
    losses of three more steps must track."""
    from deepfm_amd.training.fused_step import FusedXDeepFMStep, fused_step_class
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    rng = np.random.default_rng(41)
    fields, _, _ = _small_xdeepfm()
    ids, dense, labels = _pool(fields, 4, B, rng)
    runs = []
    for fused in (False, True):
        _, _, model = _small_xdeepfm(seed=5)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        if fused:
            assert fused_step_class(model) is FusedXDeepFMStep
        step = (FusedXDeepFMStep if fused else RowSparseTrainStep)(model, opt, B, use_graph=False)
        losses, snap = [], None
        for i in range(4):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
            losses.append(float(step.loss))
            if i == 0:
                snap = dict(sq=float(opt.sq_norm), m=npy(opt.flat_m).copy(), tab=[npy(t).copy() for t in opt.exp_avg[:6]])
        runs.append((losses, snap))
    (l0, s0), (l1, s1) = runs
    assert abs(s0["sq"] - s1["sq"]) < 1e-4 * s0["sq"]
    assert_close(s1["m"], s0["m"], rtol=1e-4, atol_scale=2e-5, what="dense exp_avg after step 1")
    for a, b in zip(s1["tab"], s0["tab"]):
        assert_close(a, b, rtol=1e-4, atol_scale=2e-5, what="table exp_avg after step 1")
    for a, b in zip(l1, l0):
        assert abs(a - b) < 1e-4 * abs(b), (l1, l0)


def test_fused_xdeepfm_graph_replay_is_bitwise_equal_to_eager():
    from deepfm_amd.training.fused_step import FusedXDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    B = 640
    rng = np.random.default_rng(43)
    fields, _, _ = _small_xdeepfm()
    ids, dense, labels = _pool(fields, 4, B, rng)
    results = []
    for use_graph in (False, True):
        _, _, model = _small_xdeepfm(seed=6)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = FusedXDeepFMStep(model, opt, B, use_graph=use_graph)
        start = copy.deepcopy(model.state_dict())
        step.capture()
        for k, v in model.state_dict().items():
            assert torch.equal(v, start[k]), f"capture() changed {k}"
        recs = step.pack_batches(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        for i in range(4):
            step.run_from(recs[i])
        torch.cuda.synchronize()
        results.append({k: npy(v).copy() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), f"graph != eager: {k}"


# ------------------------------------------------------------------ AttentionDeepFM on the fused tower
def _small_attention_deepfm(seed=0, V=300, layers=1):
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from tests.helpers import schema_from_fields
    from tools_shared import criteo_fields
    fields = criteo_fields(V, 16)
    cfg = ExperimentConfig()
    cfg.dnn.hidden_units, cfg.dnn.dropout = [64, 32], 0.0
    cfg.attention.num_heads, cfg.attention.attention_dim, cfg.attention.num_layers = 4, 32, layers
    torch.manual_seed(seed)
    model = create_model("attention_deepfm", schema_from_fields(fields), cfg).cuda().train()
    model.embedding.set_grad_mode("rowsparse")
    return fields, cfg, model


@pytest.mark.parametrize("layers", [1, 2])
def test_fused_attention_deepfm_step_matches_the_autograd_step(layers):
    """FusedAttentionDeepFMStep against RowSparseTrainStep over autograd (whose forward and gradients are
    pinned by the reference golden ``model_attention_deepfm``): Adam moments after one step, then losses."""
    from deepfm_amd.training.fused_step import FusedAttentionDeepFMStep, fused_step_class
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    B = 600
    rng = np.random.default_rng(51)
    fields, _, _ = _small_attention_deepfm()
    ids, dense, labels = _pool(fields, 4, B, rng)
    runs = []
    for fused in (False, True):
        _, _, model = _small_attention_deepfm(seed=5, layers=layers)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        if fused:
            assert fused_step_class(model) is FusedAttentionDeepFMStep
        # the optimizer lays W_q | W_k | W_v (and their gradients) out back to back: the stacked projection
        # weight is a view, not a torch.cat per step (attention.py stacked_view)
        from deepfm_amd.models.layers.attention import stacked_view
        for blk in model.attention.layers:
            for group in blk.adjacent_parameters():
                assert stacked_view(group) is not None and stacked_view([p.grad for p in group]) is not None
        step = (FusedAttentionDeepFMStep if fused else RowSparseTrainStep)(model, opt, B, use_graph=False)
        losses, snap = [], None
        for i in range(4):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
            losses.append(float(step.loss))
            if i == 0:
                snap = dict(sq=float(opt.sq_norm), m=npy(opt.flat_m).copy(), tab=[npy(t).copy() for t in opt.exp_avg[:6]])
        runs.append((losses, snap))
    (l0, s0), (l1, s1) = runs
    assert abs(s0["sq"] - s1["sq"]) < 1e-4 * s0["sq"]
    assert_close(s1["m"], s0["m"], rtol=1e-4, atol_scale=2e-5, what="dense exp_avg after step 1")
    for a, b in zip(s1["tab"], s0["tab"]):
        assert_close(a, b, rtol=1e-4, atol_scale=2e-5, what="table exp_avg after step 1")
    for a, b in zip(l1, l0):
        assert abs(a - b) < 1e-4 * abs(b), (l1, l0)


def test_fused_attention_deepfm_graph_replay_is_bitwise_equal_to_eager():
    from deepfm_amd.training.fused_step import FusedAttentionDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    B = 640
    rng = np.random.default_rng(53)
    fields, _, _ = _small_attention_deepfm()
    ids, dense, labels = _pool(fields, 4, B, rng)
    results = []
    for use_graph in (False, True):
        _, _, model = _small_attention_deepfm(seed=6)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = FusedAttentionDeepFMStep(model, opt, B, use_graph=use_graph)
        start = copy.deepcopy(model.state_dict())
        step.capture()
        for k, v in model.state_dict().items():
            assert torch.equal(v, start[k]), f"capture() changed {k}"
        recs = step.pack_batches(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        for i in range(4):
            step.run_from(recs[i])
        torch.cuda.synchronize()
        results.append({k: npy(v).copy() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), f"graph != eager: {k}"
