"""GPU parity: FeatureEmbedding / FMInteraction HIP kernels (through the C ABI) against the
reference's golden vectors and the numpy oracle.

Bars (BASELINE.json): SPARSE row gathers bit-exact; everything floating point within
1e-4 relative (helpers.assert_close: |err| <= 1e-4*|ref| + 1e-5*max|ref|).
"""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import (assert_close, fields_of, group, load, load_params, npy, random_fields_batch,
                           schema_from_fields, to_device_batch)

pytestmark = pytest.mark.gpu

EMB_CASES = ["emb_movielens_mean", "emb_movielens_sum", "emb_movielens_max", "emb_criteo_d16",
             "emb_criteo_d32", "emb_layers_test_schema"]


def _module(g):
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    emb = FeatureEmbedding(schema_from_fields(fields_of(g)), fm_embed_dim=int(g["fm_dim"]))
    emb.strict_indices = True
    return load_params(emb, group(g, "param/"))


@pytest.mark.parametrize("case", EMB_CASES)
def test_embedding_forward_backward_vs_golden(case):
    g = load(case)
    fields = fields_of(g)
    emb = _module(g)
    assert sorted(emb.state_dict().keys()) == sorted(group(g, "param/").keys())
    fo, fe, fl = emb(to_device_batch(group(g, "batch/")))
    assert fo.shape == g["out/first_order"].shape and fe.shape == g["out/field_embeddings"].shape
    assert fl.shape == g["out/flat_embeddings"].shape
    assert_close(npy(fo), g["out/first_order"], what="first_order")
    assert_close(npy(fe), g["out/field_embeddings"], what="field_embeddings")
    assert_close(npy(fl), g["out/flat_embeddings"], what="flat_embeddings")
    off = 0
    for f in fields:                                   # pure gathers: bit-exact
        if f["type"] == "sparse":
            assert np.array_equal(npy(fl)[:, off:off + f["dim"]], g["out/flat_embeddings"][:, off:off + f["dim"]]), f["name"]
        off += f["dim"]
    up = {k: torch.from_numpy(g["upstream/" + k]).cuda() for k in ("first_order", "field_embeddings", "flat_embeddings")}
    loss = (fo * up["first_order"]).sum() + (fe * up["field_embeddings"]).sum() + (fl * up["flat_embeddings"]).sum()
    loss.backward()
    want = group(g, "grad/")
    for k, p in emb.named_parameters():
        assert p.grad is not None, k                   # tests/test_layers.py:53-62
        assert_close(npy(p.grad), want[k], what=k)


def test_padding_idx_zero_gives_exact_zeros():
    """tests/test_layers.py:43-51"""
    g = load("emb_layers_test_schema")
    emb = _module(g)
    batch = {f["name"]: torch.zeros(2, dtype=torch.long, device="cuda") for f in fields_of(g)}
    with torch.no_grad():
        for out in emb(batch):
            assert float(out.abs().sum()) == 0.0


def test_out_of_range_index_raises_index_error():
    g = load("emb_layers_test_schema")
    emb = _module(g)
    batch = {f["name"]: torch.ones(2, dtype=torch.long, device="cuda") for f in fields_of(g)}
    batch["g"] = torch.tensor([1, 3], device="cuda")           # vocab of "g" is 3
    with pytest.raises(IndexError):
        emb(batch)
    with pytest.raises(KeyError):
        emb({"u": batch["u"]})


def test_empty_batch_and_ragged_tail():
    g = load("emb_criteo_d16")
    emb = _module(g)
    fields, params = fields_of(g), group(g, "param/")
    rng = np.random.default_rng(5)
    for B in (0, 1, 15, 17, 33):
        batch = random_fields_batch(fields, B, rng, zero_frac=0.2)
        with torch.no_grad():
            fo, fe, fl = emb(to_device_batch(batch))
        assert fo.shape == (B, 1) and fe.shape == (B, 39, 16) and fl.shape == (B, 624)
        if B:
            ofo, ofe, ofl = O.embedding_forward(fields, params, batch, 16)
            assert np.array_equal(npy(fe)[:, :26], ofe[:, :26])
            assert_close(npy(fo), ofo, what="fo")
            assert_close(npy(fl), ofl, what="flat")


def test_criteo_batch4096_vs_oracle_and_fused_fm():
    """Full batch size of BASELINE.json config 2 at a vocabulary the oracle handles in seconds."""
    from tools_shared import criteo_fields
    fields = criteo_fields(20000, 16)
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    torch.manual_seed(0)
    emb = FeatureEmbedding(schema_from_fields(fields), 16).cuda()
    params = {k: npy(v) for k, v in emb.state_dict().items()}
    rng = np.random.default_rng(1)
    batch = random_fields_batch(fields, 4096, rng)
    dbatch = to_device_batch(batch)
    with torch.no_grad():
        fo, fe, fl = emb(dbatch)
        inputs, B = emb._gather_inputs(dbatch)
        _, _, _, fm_fused = emb._launch_forward(inputs, B, want_fm=True)
    ofo, ofe, ofl = O.embedding_forward(fields, params, batch, 16)
    assert np.array_equal(npy(fe)[:, :26], ofe[:, :26])      # gathers bit-exact
    assert_close(npy(fe), ofe, what="fe")
    assert_close(npy(fo), ofo, what="fo")
    assert fl.data_ptr() == fe.data_ptr()                    # aliased view, no second write
    assert_close(npy(fm_fused), O.fm_forward(ofe), what="fused fm")


def test_fm_vs_golden_and_known_answers():
    from deepfm_amd.models.layers.fm import FMInteraction
    g = load("fm")
    fm = FMInteraction()
    assert len(list(fm.parameters())) == 0                   # tests/test_layers.py:75-77
    x = torch.from_numpy(g["x"]).cuda().requires_grad_()
    out = fm(x)
    assert out.shape == (64, 1)
    assert_close(npy(out), g["out"], what="fm")
    (out * torch.from_numpy(g["upstream"]).cuda()).sum().backward()
    assert_close(npy(x.grad), g["d_x"], what="fm d_x")
    assert float(fm(torch.from_numpy(g["known_x"]).cuda())[0, 0]) == 67.0   # notes/deepfm.md:72-90
    assert np.allclose(npy(fm(torch.from_numpy(g["single_x"]).cuda())), 0.0, atol=1e-5)
    assert_close(npy(out), O.fm_pairwise(g["x"]), what="pairwise")          # tests/test_layers.py:79-92


@pytest.mark.parametrize("shape", [(3, 5, 6), (130, 39, 32), (7, 2, 1), (1, 1, 4)])
def test_fm_odd_shapes(shape):
    from deepfm_amd.models.layers.fm import FMInteraction
    rng = np.random.default_rng(shape[0])
    x = rng.standard_normal(shape).astype(np.float32)
    g = rng.standard_normal((shape[0], 1)).astype(np.float32)
    t = torch.from_numpy(x).cuda().requires_grad_()
    out = FMInteraction()(t)
    (out * torch.from_numpy(g).cuda()).sum().backward()
    # a single field gives exactly 0 in exact arithmetic: absolute floor for that case
    assert_close(npy(out), O.fm_forward(x), what="fm", floor=1e-6)
    assert_close(npy(t.grad), O.fm_backward(x, g), what="fm bwd", floor=1e-6)


def test_row_gradients_with_hot_ids_vs_oracle():
    """A skewed batch (most samples of a field on ONE id): runs far longer than the 64 contributions a row's
    own lanes sum in sample order go through the workgroup-wide reduction of csrc/tail_bodies.h::rowgrad_body
    — against the oracle's reduction (tolerance: the summation tree differs), and bitwise equal from run to run."""
    import numpy as np
    import torch
    from deepfm_amd import _lib
    from oracle import ctr_oracle as O
    lib = _lib.load()
    rng = np.random.default_rng(9)
    S, F, D, B, V = 3, 5, 16, 4096 + 300, 50
    ids = np.where(rng.random((S, B)) < 0.55, V - 1, rng.integers(1, V, size=(S, B))).astype(np.int64)
    ids[1, :] = 7                                                     # a field with a single id: one run of B
    g_fe = rng.standard_normal((B, F, D)).astype(np.float32)
    g_fo = rng.standard_normal(B).astype(np.float32)
    fmap = [0, 2, 4]
    t_ids = torch.from_numpy(ids).cuda()
    t_fe, t_fo = torch.from_numpy(g_fe).cuda(), torch.from_numpy(g_fo).cuda()
    CH = _lib.ROWPLAN_CHUNK
    chunks = (B + CH - 1) // CH
    i32 = dict(dtype=torch.int32, device="cuda")
    sorted_pos, uniq = torch.empty(chunks, S, CH, **i32), torch.empty(chunks, S, CH, **i32)
    seg, num = torch.empty(chunks, S, CH + 1, **i32), torch.zeros(chunks, S, **i32)
    err = torch.zeros(1, **i32)
    import ctypes as C
    idp = (C.c_void_p * S)(*[t_ids[s].data_ptr() for s in range(S)])
    vocab = (C.c_int32 * S)(*([V] * S))
    _lib.check(lib.dfm_rowplan_build(idp, vocab, S, B, sorted_pos.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                     num.data_ptr(), err.data_ptr(), None, 0, _lib.stream_handle()))
    outs = []
    for _ in range(2):
        g2 = torch.zeros(chunks, S, CH, D, device="cuda")
        g1 = torch.zeros(chunks, S, CH, device="cuda")
        _lib.check(lib.dfm_rowgrad_build((C.c_int32 * S)(*fmap), S, F, D, B, t_fo.data_ptr(), t_fe.data_ptr(),
                                         sorted_pos.data_ptr(), seg.data_ptr(), num.data_ptr(), g2.data_ptr(),
                                         g1.data_ptr(), _lib.stream_handle()))
        outs.append((g2.cpu().numpy(), g1.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    nu, un = num.cpu().numpy(), uniq.cpu().numpy()
    longest = 0
    for c in range(chunks):
        lo, hi = c * CH, min(B, (c + 1) * CH)
        for s in range(S):
            u, a2, a1 = O.rowsparse_reduce_fast(ids[s, lo:hi], g_fe[lo:hi, fmap[s]], g_fo[lo:hi])
            n = int(nu[c, s])
            assert np.array_equal(un[c, s, :n], u)
            np.testing.assert_allclose(outs[0][0][c, s, :n], a2, rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(outs[0][1][c, s, :n], a1, rtol=1e-4, atol=1e-3)
            longest = max(longest, int(np.bincount(ids[s, lo:hi]).max()))
    assert longest > 2000
