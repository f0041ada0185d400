"""bf16 x 6 tower (csrc/gemm_x6.h, dfm_tower_set_mode(2)): the exact three-way split, the plane layout of both roles,
and the forward / backward GEMMs on planes against float64 — held to the bar of the exact-fp32 kernels, which run
beside them on the same inputs, and to 'no worse than twice fp32's own error'.  Reference: deepfm/models/layers/dnn.py:45-59
(Linear -> BatchNorm1d -> ReLU -> Dropout)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, npy

pytestmark = pytest.mark.gpu


def _lib():
    from deepfm_amd import _lib
    return _lib, _lib.load()


def planes_buffer(lib, rows, contraction):
    return torch.zeros(lib.dfm_planes_bytes(rows, contraction) // 2, dtype=torch.bfloat16, device="cuda")


def planes_to_matrix(buf, rows, contraction):
    """(3, rows, contraction) float64 from a plane set [3][G][Rp][8]; also checks that the pads are zero."""
    G, Rp = (contraction + 63) // 64 * 8, (rows + 63) // 64 * 64
    a = buf[:3 * G * Rp * 8].view(3, G, Rp, 8).permute(0, 2, 1, 3).reshape(3, Rp, G * 8).double().cpu().numpy()
    assert not a[:, rows:, :].any() and not a[:, :, contraction:].any(), "pads must stay zero"
    return a[:, :rows, :contraction]


def split_both(lib, _l, x):
    R, Cc = x.shape
    pf, ps = planes_buffer(lib, R, Cc), planes_buffer(lib, Cc, R)
    job = (_l.SplitJob * 1)()
    job[0].src, job[0].rows, job[0].cols = x.data_ptr(), R, Cc
    job[0].planes_f, job[0].planes_s = pf.data_ptr(), ps.data_ptr()
    _l.check(lib.dfm_split_planes(job, 1, _l.stream_handle()))
    torch.cuda.synchronize()
    return pf, ps


@pytest.mark.parametrize("shape", [(400, 624), (64, 64), (8, 8), (40, 200), (416, 32)])
def test_split_is_exact_in_both_roles(shape):
    _l, lib = _lib()
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32) * np.exp(rng.uniform(-20, 20, shape)).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1.0, -1.5]
    xt = torch.from_numpy(x).cuda()
    pf, ps = split_both(lib, _l, xt)
    f = planes_to_matrix(pf, shape[0], shape[1])
    s = planes_to_matrix(ps, shape[1], shape[0])
    want = x.astype(np.float64)
    assert np.array_equal(f.sum(0), want), "role F: h + m + l must equal the fp32 value exactly"
    assert np.array_equal(s.sum(0).T, want), "role S"
    # each plane is what truncation leaves: |m| < 2^-7 |h|, |l| < 2^-7 |m| wherever they are non-zero
    nz = f[0] != 0
    assert (np.abs(f[1][nz]) <= np.abs(f[0][nz]) * 2.0 ** -7).all()
    nz = f[1] != 0
    assert (np.abs(f[2][nz]) <= np.abs(f[1][nz]) * 2.0 ** -7).all()


def _err(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


@pytest.mark.parametrize("B,N,K,planes_x", [(4096, 400, 624, False), (4096, 400, 400, True), (256, 64, 64, True),
                                            (128, 8, 8, False), (192, 200, 1248, False), (4096, 400, 1248, True),
                                            (320, 136, 72, True)])
def test_forward_x6_vs_fp64(B, N, K, planes_x):
    """z = x W^T + b and the per-tile statistics workspace: x either fp32 (split by the kernel: layer 1) or role-F planes."""
    _l, lib = _lib()
    st = _l.stream_handle()
    rng = np.random.default_rng(B + N + K)
    x = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
    want = (x.double() @ w.double().T + b.double()).cpu().numpy()
    ws_bytes = lib.dfm_linear_bn_workspace_bytes(B, N)
    ws0 = torch.zeros(ws_bytes // 4, device="cuda")
    ws6 = torch.zeros(ws_bytes // 4, device="cuda")
    z0 = torch.empty(B, N, device="cuda")
    z6 = torch.full((B, N), float("nan"), device="cuda")
    _l.check(lib.dfm_linear_bn_forward(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), B, N, K, z0.data_ptr(),
                                       ws0.data_ptr(), st))
    wf, _ = split_both(lib, _l, w)
    xf = split_both(lib, _l, x)[0] if planes_x else None
    _l.check(lib.dfm_linear_bn_forward_x6(None if planes_x else x.data_ptr(), K, xf.data_ptr() if planes_x else None,
                                          wf.data_ptr(), b.data_ptr(), B, N, K, z6.data_ptr(), ws6.data_ptr(), st))
    torch.cuda.synchronize()
    assert_close(npy(z6), want, what="z (bf16 x 6)")
    e0, e6 = _err(npy(z0), want), _err(npy(z6), want)
    assert e6 <= 2 * e0 + 1e-7, f"bf16 x 6 error {e6:.2e} against fp32 MFMA's {e0:.2e}"
    # same per-tile statistics (tile mean, M2) to rounding
    T = (B + 31) // 32
    s0, s6 = npy(ws0)[:T * 2 * N], npy(ws6)[:T * 2 * N]
    assert_close(s6, s0, rtol=1e-4, atol_scale=1e-5, what="tile statistics")


@pytest.mark.parametrize("B,N,K,epi,planes_x", [(4096, 400, 624, "fm", False), (4096, 400, 400, "plain", True),
                                               (256, 64, 64, "plain", True), (128, 8, 8, "plain", False),
                                               (4096, 400, 1248, "plain", False), (320, 136, 72, "plain", True),
                                               (4096, 400, 400, "bn", True)])
def test_backward_x6_vs_fp64(B, N, K, epi, planes_x):
    """d W = d z^T x (batch-split slabs + finish) and d x = d z W with each epilogue, operands as planes."""
    _l, lib = _lib()
    st = _l.stream_handle()
    rng = np.random.default_rng(B + N + K + 1)
    dz = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32) * 1e-3).cuda()
    x = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).cuda()
    want_dw = (dz.double().T @ x.double()).cpu().numpy()
    want_dx = (dz.double() @ w.double()).cpu().numpy()
    dzf, dzs = split_both(lib, _l, dz)
    _, ws_ = split_both(lib, _l, w)
    xs = split_both(lib, _l, x)[1] if planes_x else None
    fm = None
    bn = None
    keep = []
    if epi == "fm":
        D = 16
        g_fm = torch.from_numpy(rng.standard_normal(B).astype(np.float32) * 1e-3).cuda()
        fm_sum = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
        fm = _l.FmBwd()
        fm.g_fm, fm.fm_sum, fm.e, fm.addend, fm.dim = g_fm.data_ptr(), fm_sum.data_ptr(), x.data_ptr(), None, D
        want_dx = want_dx + (g_fm.double()[:, None] * (fm_sum.double().repeat(1, K // D) - x.double())).cpu().numpy()
        keep += [g_fm, fm_sum]
    if epi == "bn":      # the lower layer's BatchNorm mask: compare with the fp32 kernel's dy and partial sums
        zb = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda()
        stats = torch.stack([zb.mean(0), 1.0 / torch.sqrt(zb.var(0, unbiased=False) + 1e-5)]).contiguous()
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, K).astype(np.float32)).cuda()
        beta = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).cuda()
        seed = torch.tensor([12345], dtype=torch.int64, device="cuda")
        outs = []
        for _ in range(2):
            dy = torch.zeros(B, K, device="cuda")
            gg, gb = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
            wsb = torch.zeros(lib.dfm_bn_bwd_workspace_bytes(B, K) // 4, device="cuda")
            c = _l.BnBwd()
            c.z, c.mean_rstd, c.gamma, c.beta = zb.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
            c.dy, c.g_gamma, c.g_beta, c.seed = dy.data_ptr(), gg.data_ptr(), gb.data_ptr(), seed.data_ptr()
            c.workspace, c.p_drop, c.salt = wsb.data_ptr(), 0.25, 1
            outs.append((c, dy, wsb, gg, gb))
        keep += [zb, stats, gamma, beta, seed, outs]
    g_w0, g_w6 = torch.zeros(N, K, device="cuda"), torch.zeros(N, K, device="cuda")
    g_x0 = torch.empty(B, K, device="cuda")
    g_x6 = torch.full((B, K), float("nan"), device="cuda")
    ws0 = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, N, K) // 4, 1), device="cuda")
    ws6 = torch.zeros(max(lib.dfm_linear_backward_x6_workspace_bytes(B, N, K) // 4, 1), device="cuda")
    _l.check(lib.dfm_linear_backward(dz.data_ptr(), B, N, x.data_ptr(), K, w.data_ptr(), g_x0.data_ptr(),
                                     C.byref(outs[0][0]) if epi == "bn" else None,
                                     C.byref(fm) if fm is not None else None, 3, ws0.data_ptr(), st))
    _l.check(lib.dfm_linear_backward_x6(dzf.data_ptr(), dzs.data_ptr(), B, N, None if planes_x else x.data_ptr(),
                                        xs.data_ptr() if planes_x else None, K, ws_.data_ptr(), g_x6.data_ptr(),
                                        C.byref(outs[1][0]) if epi == "bn" else None,
                                        C.byref(fm) if fm is not None else None, ws6.data_ptr(), st))
    refs = (_l.SlabRef * 2)()
    refs[0].workspace, refs[0].g_w = ws0.data_ptr(), g_w0.data_ptr()
    refs[0].batch, refs[0].out_features, refs[0].in_features = B, N, K
    refs[1].workspace, refs[1].g_w = ws6.data_ptr(), g_w6.data_ptr()
    refs[1].batch, refs[1].out_features, refs[1].in_features = B, N, K
    refs[1].splits = lib.dfm_linear_backward_x6_splits(B, N, K)
    _l.check(lib.dfm_linear_backward_finish(refs, 2, st))
    torch.cuda.synchronize()
    assert_close(npy(g_w6), want_dw, what="dW (bf16 x 6)")
    e0, e6 = _err(npy(g_w0), want_dw), _err(npy(g_w6), want_dw)
    assert e6 <= 2 * e0 + 1e-7, f"dW: bf16 x 6 error {e6:.2e} against fp32 MFMA's {e0:.2e}"
    if epi == "bn":
        assert_close(npy(outs[1][1]), npy(outs[0][1]), rtol=1e-4, atol_scale=1e-5, what="masked dy")
        assert_close(npy(outs[1][2]), npy(outs[0][2]), rtol=1e-4, atol_scale=1e-5, what="dy column sums")
    else:
        assert_close(npy(g_x6), want_dx, what="dx (bf16 x 6)")
        e0, e6 = _err(npy(g_x0), want_dx), _err(npy(g_x6), want_dx)
        assert e6 <= 2 * e0 + 1e-7, f"dx: bf16 x 6 error {e6:.2e} against fp32 MFMA's {e0:.2e}"


@pytest.mark.parametrize("B,N", [(4096, 400), (64, 8), (96, 72)])
def test_apply_kernels_write_exact_planes(B, N):
    """dfm_bn_relu_dropout_apply_planes / dfm_bn_backward_apply_planes: the fp32 result is bit-identical to the plain
    entry point's and the planes of both roles sum to it exactly."""
    _l, lib = _lib()
    st = _l.stream_handle()
    rng = np.random.default_rng(B + N)
    x = torch.from_numpy(rng.standard_normal((B, 16)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.standard_normal((N, 16)).astype(np.float32)).cuda()
    z = torch.empty(B, N, device="cuda")
    ws = torch.zeros(lib.dfm_linear_bn_workspace_bytes(B, N) // 4, device="cuda")
    _l.check(lib.dfm_linear_bn_forward(x.data_ptr(), 16, w.data_ptr(), None, B, N, 16, z.data_ptr(), ws.data_ptr(), st))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.standard_normal(N).astype(np.float32) * 0.1).cuda()
    seed = torch.tensor([777], dtype=torch.int64, device="cuda")
    st0, st1 = torch.empty(2, N, device="cuda"), torch.empty(2, N, device="cuda")
    a0, a1 = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    pf, ps = planes_buffer(lib, B, N), planes_buffer(lib, N, B)
    common = (z.data_ptr(), B, N, ws.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    tail = (None, None, None, 0.1, 1e-5, 0.3, seed.data_ptr(), 2)
    _l.check(lib.dfm_bn_relu_dropout_apply(*common, st0.data_ptr(), *tail, a0.data_ptr(), st))
    _l.check(lib.dfm_bn_relu_dropout_apply_planes(*common, st1.data_ptr(), *tail, a1.data_ptr(), pf.data_ptr(),
                                                  ps.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(a0, a1) and torch.equal(st0, st1)
    want = npy(a0).astype(np.float64)
    assert np.array_equal(planes_to_matrix(pf, B, N).sum(0), want)
    assert np.array_equal(planes_to_matrix(ps, N, B).sum(0).T, want)
    # planes only (no fp32 output)
    pf2, ps2 = planes_buffer(lib, B, N), planes_buffer(lib, N, B)
    _l.check(lib.dfm_bn_relu_dropout_apply_planes(*common, st1.data_ptr(), *tail, None, pf2.data_ptr(), ps2.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(pf, pf2) and torch.equal(ps, ps2)

    # backward apply: dy -> dz
    dy = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32) * 1e-3).cuda()
    T = (B + 31) // 32
    part = torch.from_numpy(rng.standard_normal(T * 2 * N).astype(np.float32)).cuda()
    wsb = torch.zeros(max(lib.dfm_bn_bwd_workspace_bytes(B, N) // 4, T * 2 * N), device="cuda")
    wsb[:T * 2 * N] = part
    outs = []
    for planes in (False, True):
        gg, gb = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        dz = torch.empty(B, N, device="cuda")
        c = _l.BnBwd()
        c.z, c.mean_rstd, c.gamma, c.beta = z.data_ptr(), st0.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        c.dy, c.g_gamma, c.g_beta, c.seed = dy.data_ptr(), gg.data_ptr(), gb.data_ptr(), seed.data_ptr()
        c.workspace, c.p_drop, c.salt = wsb.data_ptr(), 0.3, 2
        qf, qs = planes_buffer(lib, B, N), planes_buffer(lib, N, B)
        if planes:
            _l.check(lib.dfm_bn_backward_apply_planes(C.byref(c), B, N, None, dz.data_ptr(), qf.data_ptr(), qs.data_ptr(), st))
        else:
            _l.check(lib.dfm_bn_backward_apply(C.byref(c), B, N, None, dz.data_ptr(), st))
        torch.cuda.synchronize()
        outs.append((dz, gg, gb, qf, qs))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    want = npy(outs[1][0]).astype(np.float64)
    assert np.array_equal(planes_to_matrix(outs[1][3], B, N).sum(0), want)
    assert np.array_equal(planes_to_matrix(outs[1][4], N, B).sum(0).T, want)


def test_unsupported_shapes_are_refused():
    _l, lib = _lib()
    assert lib.dfm_tower_x6_supported(4096, 400, 624) == 1
    assert lib.dfm_tower_x6_supported(4000, 400, 624) == 0      # batch % 64
    assert lib.dfm_tower_x6_supported(4096, 396, 624) == 0      # features % 8
    assert lib.dfm_tower_x6_supported(4096, 400, 429) == 0


def test_fused_step_on_planes_vs_oracle_and_graph():
    """dfm_tower_set_mode(2) end to end, in the default test run: the fused DeepFM step takes the planes path
    (weight split launch, *_x6 GEMMs, apply kernels that write planes only), three steps against the numpy oracle at the
    bar of tests/test_gpu_fused_tower.py::test_fused_deepfm_steps_vs_oracle, and its graph replay is bit-identical to
    its eager run.  (The whole suite under this mode: DFM_TEST_TOWER_MODE=2, tests/conftest.py.)"""
    from oracle import ctr_oracle as O
    from tests.test_gpu_fused_tower import _fused_pair, _oracle_state, _pool
    _l, lib = _lib()
    old = lib.dfm_tower_get_mode()
    B = 512
    try:
        _l.check(lib.dfm_tower_set_mode(2))
        fields, cfg, model, hp, opt, Step = _fused_pair(B)
        params, state = _oracle_state(model)
        step = Step(model, opt, B, use_graph=False)
        assert step.x6, "batch 512, hidden units multiples of 8: the planes path must be taken"
        rng = np.random.default_rng(3)
        ids, dense, labels = _pool(fields, 3, B, rng)
        ocfg = dict(fm_dim=16, hidden_units=cfg.dnn.hidden_units)
        for i in range(3):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
            batch = {f["name"]: ids[i, j] for j, f in enumerate(fields[:26])}
            batch.update({f["name"]: dense[i, j] for j, f in enumerate(fields[26:])})
            oloss = O.deepfm_train_step_rowsparse(fields, params, state, batch, labels[i], ocfg, hp, i + 1, exact_order=True)
            assert abs(float(step.loss) - float(oloss)) < 2e-5 + 1e-4 * abs(float(oloss)), (i, float(step.loss), float(oloss))
        got = {k: npy(v) for k, v in model.state_dict().items()}
        for k, want in params.items():
            if "running_" in k:
                continue
            if k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0:
                continue        # zero-gradient parameter (Linear bias in front of BatchNorm)
            assert_close(got[k], want, rtol=1e-4, atol_scale=0.0, floor=1e-4, what=k)
        # graph replay == eager, bit for bit
        results = []
        for use_graph in (False, True):
            _, _, model, hp, opt, Step = _fused_pair(B, seed=4)
            step = Step(model, opt, B, use_graph=use_graph)
            recs = step.pack_batches(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
            step.capture()
            for i in range(3):
                step.run_from(recs[i])
            torch.cuda.synchronize()
            results.append({k: npy(v).copy() for k, v in model.state_dict().items()})
        for k in results[0]:
            assert np.array_equal(results[0][k], results[1][k]), f"planes path: graph != eager: {k}"
    finally:
        _l.check(lib.dfm_tower_set_mode(old))
