"""GPU: the fused BatchNorm -> ReLU -> Dropout kernels of the DNN tower against torch.nn
(the reference's own modules, dnn.py:45-55) and the oracle."""
import copy

import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, npy

pytestmark = pytest.mark.gpu


def _pair(in_dim, hidden, dropout):
    from deepfm_amd.models.layers.dnn import DNN
    torch.manual_seed(0)
    fused = DNN(in_dim, hidden, "relu", dropout, True).cuda().train()
    plain = copy.deepcopy(fused)
    plain.fused = False
    with torch.no_grad():
        for m in fused.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
        plain.load_state_dict(fused.state_dict())
    return fused, plain


def _relu_masks_agree(fused, plain, x):
    """True when both paths sit on the same side of every ReLU kink for this input.  The two
    paths sum the Linear layers in different orders, so a pre-activation within a few ulp of 0
    can land on different sides; the derivative is discontinuous there, and through BatchNorm
    one flipped element moves the gradient of every row — gradient parity is only defined
    away from the kinks."""
    from deepfm_amd.models.layers.dnn import _LinearBnReluDropoutFn
    fa, pb = copy.deepcopy(fused), copy.deepcopy(plain)      # keep the running statistics untouched
    ha, hb = x, x
    with torch.no_grad():
        for i in range(len(fa.mlp) // 4):
            lin, bn, act, drop = (fa.mlp[4 * i + j] for j in range(4))
            ha = _LinearBnReluDropoutFn.apply(ha, lin.weight, lin.bias, bn.weight, bn.bias, bn, 0.0, None, i, False)
            hb = pb.mlp[4 * i + 2](pb.mlp[4 * i + 1](pb.mlp[4 * i](hb)))
            if not torch.equal(ha > 0, hb > 0):
                return False
    return True


@pytest.mark.parametrize("shape", [(4096, 624, [256, 128, 64]), (37, 10, [7, 5]), (2, 3, [4])])
def test_fused_matches_torch_modules(shape):
    B, in_dim, hidden = shape
    fused, plain = _pair(in_dim, hidden, 0.0)
    for attempt in range(20):
        g0 = torch.Generator(device="cuda").manual_seed(100 + attempt)
        x = torch.randn(B, in_dim, device="cuda", generator=g0) * 2 + 0.5
        if _relu_masks_agree(fused, plain, x):
            break
    else:
        pytest.fail("no input without a ReLU-kink disagreement in 20 draws")
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = fused(xa), plain(xb)
    rtol = 1e-2 if B < 8 else 1e-4        # a batch of 2 makes BatchNorm itself ill-conditioned
    assert_close(npy(ya), npy(yb), rtol=rtol, what="output")
    g = torch.randn_like(ya)
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    assert_close(npy(xa.grad), npy(xb.grad), rtol=rtol, what="d_x")
    for (k, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        zero_grad = k.endswith(".bias") and int(k.split(".")[1]) % 4 == 0     # Linear bias before BN
        assert pa.grad is not None, k
        # (identically-zero gradient: both sides hold rounding noise that grows with the batch)
        # a batch of 2 normalises to exactly +-1: every gradient through BatchNorm cancels to ~1e-6
        floor = 1e-3 if zero_grad else (1e-6 if B < 8 else 0.0)
        assert_close(npy(pa.grad), npy(pb.grad), rtol=rtol, what=k, floor=floor)
    for (k, ba), (_, bb) in zip(fused.named_buffers(), plain.named_buffers()):
        if k.endswith("num_batches_tracked"):
            assert int(ba) == int(bb) == 1
        else:
            assert_close(npy(ba), npy(bb), rtol=rtol, what=k)        # running_mean / running_var


def test_fused_large_column_mean():
    """Column means far from zero (|mean| >> std): the shifted-sum variance must not cancel.
    Compared with torch's BatchNorm on the SAME rocBLAS pre-activations."""
    fused, plain = _pair(16, [8], 0.0)
    x = torch.randn(512, 16, device="cuda") * 0.1 + 50.0
    # (z - mean) itself cancels here (|mean|/std ~ 500): both sides carry ~1e-4 absolute noise
    assert_close(npy(fused(x)), npy(plain(x)), rtol=5e-3, atol_scale=2e-4, what="large mean")
    want = O.dnn_forward(npy(x), {k: npy(v) for k, v in plain.state_dict().items()}, 1, training=True)
    assert_close(npy(fused(x)), want, rtol=5e-2, atol_scale=2e-3, what="oracle (ill-conditioned: loose)")


def test_dropout_mask_consistent_and_fresh():
    fused, _ = _pair(64, [128], 0.25)
    x = torch.randn(2048, 64, device="cuda").requires_grad_()
    y = fused(x)
    keep = (npy(y) != 0)
    relu_on = keep.mean()
    assert 0.2 < relu_on < 0.55                      # ~half pass ReLU, 75 % of those are kept
    g = torch.ones_like(y)
    y.backward(g)
    y2 = fused(x.detach())
    assert (npy(y2) != 0).mean() > 0 and not np.array_equal(npy(y2) != 0, keep)   # new seed, new mask
    fused.eval()
    assert torch.isfinite(fused(x.detach())).all()   # eval falls back to nn.Sequential


def test_direct_accumulation_into_existing_grad_buffers():
    """With pre-existing .grad buffers (the row-sparse optimizer's flat views) the layer adds
    its parameter gradients in place: two backward passes give exactly twice one pass."""
    fused, plain = _pair(32, [16, 8], 0.0)
    x = torch.randn(256, 32, device="cuda")
    g = torch.randn(256, 8, device="cuda")
    (plain(x) * g).sum().backward()
    fused.direct_grads = True           # what RowSparseTrainStep sets on the model's DNN and heads
    for p in fused.parameters():
        p.grad = torch.zeros_like(p)
    for _ in range(2):
        (fused(x) * g).sum().backward()
    for (k, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        zero_grad = k.endswith(".bias") and int(k.split(".")[1]) % 4 == 0
        assert_close(npy(pa.grad), 2 * npy(pb.grad), rtol=1e-4, what=k, floor=1e-3 if zero_grad else 0.0)


def test_default_backward_returns_gradients_to_autograd():
    """Without ``direct_grads`` the layer is an ordinary autograd citizen: torch.autograd.grad() returns
    the parameter gradients and existing .grad buffers are left alone."""
    fused, plain = _pair(32, [16, 8], 0.0)
    x = torch.randn(256, 32, device="cuda")
    g = torch.randn(256, 8, device="cuda")
    (plain(x) * g).sum().backward()
    for p in fused.parameters():
        p.grad = torch.full_like(p, 7.0)
    params = list(fused.parameters())
    grads = torch.autograd.grad((fused(x) * g).sum(), params)
    for (k, pa), ga, (_, pb) in zip(fused.named_parameters(), grads, plain.named_parameters()):
        zero_grad = k.endswith(".bias") and int(k.split(".")[1]) % 4 == 0
        assert ga is not None, k
        assert_close(npy(ga), npy(pb.grad), rtol=1e-4, what=k, floor=1e-3 if zero_grad else 0.0)
        assert float((pa.grad - 7.0).abs().max()) == 0.0, f".grad of {k} was written"


def test_two_forwards_then_backward_keeps_the_first_mask():
    """Backward regenerates the dropout mask from the forward's seed: a second train-mode forward
    between a forward and its backward must not change the gradients (bitwise)."""
    fused, _ = _pair(64, [128, 32], 0.3)
    x = torch.randn(512, 64, device="cuda")
    g = torch.randn(512, 32, device="cuda")
    fused(x)                                            # creates the device seed
    seed0 = fused._seed.clone()
    results = []
    for extra_forward in (False, True):
        fused._seed.copy_(seed0)
        fused.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        y = fused(xi)
        if extra_forward:
            y2 = fused(x)
            assert not torch.equal(y2 != 0, y.detach() != 0)        # a different mask
        (y * g).sum().backward()
        results.append([npy(xi.grad).copy()] + [npy(p.grad).copy() for p in fused.parameters()])
    for a, b in zip(*results):
        assert np.array_equal(a, b)


def test_fixed_input_bounded_outliers():
    """One FIXED input at the headline tower shape, no re-drawing: a pre-activation within an ulp of 0 may
    take the other side of the ReLU kink than torch's modules (different fp32 summation order), so a
    bounded fraction of gradient elements may differ — but a kernel that flips masks systematically
    would blow the bound.  Outputs are held to the normal bar."""
    from tests.helpers import assert_close_mostly
    B, in_dim, hidden = 4096, 624, [256, 128, 64]
    fused, plain = _pair(in_dim, hidden, 0.0)
    x = torch.randn(B, in_dim, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4242)) * 2 + 0.5
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = fused(xa), plain(xb)
    # |y| of a flipped element is itself ~1e-7: the relative bar cannot see it, the outputs all pass
    assert_close(npy(ya), npy(yb), rtol=1e-4, what="output")
    g = torch.randn(B, hidden[-1], device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    assert_close_mostly(npy(xa.grad), npy(xb.grad), 2e-3, what="d_x")
    for (k, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        if k.endswith(".bias") and int(k.split(".")[1]) % 4 == 0:
            continue
        assert_close_mostly(npy(pa.grad), npy(pb.grad), 2e-3, what=k)


def test_fused_bce_matches_torch_and_oracle():
    from deepfm_amd.training.losses import bce_with_logits_mean
    rng = np.random.default_rng(5)
    z = (rng.standard_normal(4099) * 6).astype(np.float32)
    z[:4] = [60.0, -60.0, 0.0, 1e-8]
    y = (rng.random(4099) < 0.3).astype(np.float32)
    zt = torch.from_numpy(z).cuda().requires_grad_()
    loss = bce_with_logits_mean(zt.view(-1, 1), torch.from_numpy(y).cuda())
    (loss * 3.0).backward()
    zr = torch.from_numpy(z).cuda().requires_grad_()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(zr, torch.from_numpy(y).cuda())
    (ref * 3.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    assert_close(npy(zt.grad), npy(zr.grad), rtol=1e-5, what="dz")
    oloss, odz = O.bce_with_logits(z, y)
    assert abs(float(loss) - float(oloss)) < 1e-5
    assert_close(npy(zt.grad), 3.0 * odz, rtol=1e-4, what="dz vs oracle")


@pytest.mark.parametrize("shape", [(4096, 256, 624), (4096, 64, 128), (100, 33, 70), (5, 3, 7), (256, 624, 4096),
                                   (64, 128, 4096)])
def test_gemm_f32_all_layouts(shape):
    """csrc/gemm_f32.hip (exact fp32 MFMA) against torch fp64 for the four operand layouts,
    ragged tiles and the split-reduction shapes of the weight gradient."""
    from deepfm_amd.models.layers.dnn import _gemm
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    b = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    want = (a.double() @ b.double().t()).float()
    at, bt = a.t().contiguous(), b.t().contiguous()          # (K, M), (K, N): K-strided operands
    for a_kc, b_kc in ((True, True), (True, False), (False, True), (False, False)):
        c = torch.full((M, N), 7.0, device="cuda")
        _gemm(a if a_kc else at, K if a_kc else M, a_kc, b if b_kc else bt, K if b_kc else N, b_kc, c, M, N, K,
              bias=bias)
        assert_close(npy(c), npy(want + bias), rtol=1e-5, atol_scale=2e-6, what=f"layout {a_kc},{b_kc}")
    c = torch.ones(M, N, device="cuda")
    _gemm(at, M, False, bt, N, False, c, M, N, K, accumulate=True)
    assert_close(npy(c), npy(want + 1.0), rtol=1e-5, atol_scale=2e-6, what="accumulate")


@pytest.mark.parametrize("B,N,K,epi", [(4096, 256, 624, "fm"), (4096, 128, 256, "plain"), (4096, 64, 128, "plain"),
                                       (4096, 256, 2496, "plain"), (778, 64, 40, "plain"), (130, 32, 24, "plain")])
def test_linear_backward_bf16x3_vs_fp64(B, N, K, epi):
    """dfm_tower_set_mode(1): d W = d z^T x and d x = d z W of the tower's backward on the bf16 pipe with the bf16 x 3
    split (gemm_core.h::mainloop_x3) — at the tower's own shapes (configs 2 and 4), a ragged batch and a ragged k tail —
    against float64, held to the SAME bar as the exact-fp32 kernel (1e-4 relative + 1e-5 of the scale), which is run
    beside it on the same inputs."""
    import ctypes as C
    from deepfm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B + N + K)
    dz = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32) * 1e-3).cuda()
    x = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).cuda()
    want_dw = (dz.double().T @ x.double()).cpu().numpy()
    want_dx = (dz.double() @ w.double()).cpu().numpy()
    ws = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, N, K) // 4, 1), dtype=torch.float32, device="cuda")
    old = lib.dfm_tower_get_mode()
    try:
        for mode in (0, 1):
            _lib.check(lib.dfm_tower_set_mode(mode))
            g_w = torch.zeros(N, K, device="cuda")
            g_x = torch.empty(B, K, device="cuda")
            ref = (_lib.SlabRef * 1)()
            ref[0].workspace, ref[0].g_w = ws.data_ptr(), g_w.data_ptr()
            ref[0].batch, ref[0].out_features, ref[0].in_features = B, N, K
            _lib.check(lib.dfm_linear_backward(dz.data_ptr(), B, N, x.data_ptr(), K, w.data_ptr(), g_x.data_ptr(), None,
                                               None, 3, ws.data_ptr(), _lib.stream_handle()))
            _lib.check(lib.dfm_linear_backward_finish(ref, 1, _lib.stream_handle()))
            torch.cuda.synchronize()
            assert_close(npy(g_w), want_dw, what=f"dW mode {mode}")
            assert_close(npy(g_x), want_dx, what=f"dx mode {mode}")
    finally:
        _lib.check(lib.dfm_tower_set_mode(old))
