"""GPU parity: fused field self-attention kernels against reference golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, group, load, npy
from tests.test_gpu_models_step import check_model_case

pytestmark = pytest.mark.gpu

CASES = ["attn_cfg4", "attn_two_layers", "attn_no_residual", "attn_odd"]


def _module(g):
    from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
    D = g["x"].shape[2]
    att = MultiHeadSelfAttention(D, int(g["num_heads"]), int(g["attention_dim"]), int(g["num_layers"]),
                                 bool(g["use_residual"]))
    want = group(g, "param/")
    assert sorted(att.state_dict().keys()) == sorted(want.keys())
    att.load_state_dict({k: torch.from_numpy(v) for k, v in want.items()})
    return att.cuda()


@pytest.mark.parametrize("case", CASES)
def test_attention_vs_golden(case):
    g = load(case)
    att = _module(g)
    x = torch.from_numpy(g["x"]).cuda().requires_grad_()
    out = att(x)
    assert out.shape == x.shape                                   # tests/test_layers.py:175-201
    assert_close(npy(out), g["out"], what="attn out")
    (out * torch.from_numpy(g["upstream"]).cuda()).sum().backward()
    assert_close(npy(x.grad), g["d_x"], what="attn d_x")
    for k, p in att.named_parameters():
        assert p.grad is not None, k                              # tests/test_layers.py:203-210
        assert_close(npy(p.grad), g["grad/" + k], what=k, floor=2e-5 if k.endswith("W_k.bias") else 0.0)


@pytest.mark.parametrize("case", CASES)
def test_per_sample_lds_kernel_vs_golden(case):
    """csrc/attention.hip (one fused LDS kernel per sample: the path for F > 64 or a head_dim outside
    {4, 8, 16, 32}) forced on the golden shapes with ``gemm_path = False``."""
    g = load(case)
    att = _module(g)
    for block in att.layers:
        block.gemm_path = False
    x = torch.from_numpy(g["x"]).cuda().requires_grad_()
    out = att(x)
    assert_close(npy(out), g["out"], what="attn out")
    (out * torch.from_numpy(g["upstream"]).cuda()).sum().backward()
    assert_close(npy(x.grad), g["d_x"], what="attn d_x")
    for k, p in att.named_parameters():
        assert_close(npy(p.grad), g["grad/" + k], what=k, floor=2e-5 if k.endswith("W_k.bias") else 0.0)


@pytest.mark.parametrize("shape", [
    dict(B=33, F=39, D=32, heads=1, A=64, layers=1, residual=True),     # head_dim 64
    dict(B=17, F=5, D=8, heads=4, A=8, layers=2, residual=True),        # head_dim 2
    dict(B=9, F=70, D=8, heads=2, A=16, layers=1, residual=False),      # F > 64
    dict(B=12, F=11, D=10, heads=3, A=36, layers=1, residual=True),     # head_dim 12, embed_dim % 4 != 0
])
def test_shapes_the_core_kernel_rejects_vs_oracle(shape):
    """Reference-valid configurations (attention.py:26-50 accepts any head_dim / field count) that
    ``dfm_attention_core_supported`` turns down run on csrc/attention.hip: forward and every gradient
    against the oracle (itself pinned by the attention goldens)."""
    from deepfm_amd import _lib
    from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
    c = shape
    torch.manual_seed(c["F"])
    att = MultiHeadSelfAttention(c["D"], c["heads"], c["A"], c["layers"], c["residual"]).cuda()
    core = _lib.load().dfm_attention_core_supported(c["F"], c["A"], c["heads"])
    assert not (core and c["D"] % 4 == 0 and c["A"] % 4 == 0 and c["D"] <= 64), "shape would take the GEMM + core path"
    with torch.no_grad():
        for p in att.parameters():
            p.uniform_(-0.4, 0.4)
    params = {k: npy(v) for k, v in att.state_dict().items()}
    rng = np.random.default_rng(c["B"])
    x = rng.standard_normal((c["B"], c["F"], c["D"])).astype(np.float32)
    up = rng.standard_normal(x.shape).astype(np.float32)
    t = torch.from_numpy(x).cuda().requires_grad_()
    out = att(t)
    (out * torch.from_numpy(up).cuda()).sum().backward()
    assert_close(npy(out), O.attention_forward(x, params, c["heads"], c["layers"], c["residual"]), what="out")
    d_x, grads = O.attention_backward(x, params, c["heads"], c["layers"], c["residual"], up)
    assert_close(npy(t.grad), d_x, what="d_x")
    for k, p in att.named_parameters():
        # W_k.bias: softmax-invariant (exact-zero gradient); a final LayerNorm-free block has none either
        assert_close(npy(p.grad), grads[k], what=k, floor=1e-4 if k.endswith("W_k.bias") else 0.0)


def _check_vs_oracle(c, whole_block_kernel=True):
    from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
    torch.manual_seed(c["F"])
    att = MultiHeadSelfAttention(c["D"], c["heads"], c["A"], c["layers"], c["residual"]).cuda()
    for blk in att.layers:
        blk.whole_block_kernel = whole_block_kernel
    with torch.no_grad():
        for p in att.parameters():
            p.uniform_(-0.4, 0.4)
    params = {k: npy(v) for k, v in att.state_dict().items()}
    rng = np.random.default_rng(c["B"])
    x = rng.standard_normal((c["B"], c["F"], c["D"])).astype(np.float32)
    up = rng.standard_normal(x.shape).astype(np.float32)
    t = torch.from_numpy(x).cuda().requires_grad_()
    out = att(t)
    (out * torch.from_numpy(up).cuda()).sum().backward()
    assert_close(npy(out), O.attention_forward(x, params, c["heads"], c["layers"], c["residual"]), what="out")
    d_x, grads = O.attention_backward(x, params, c["heads"], c["layers"], c["residual"], up)
    assert_close(npy(t.grad), d_x, what="d_x")
    for k, p in att.named_parameters():
        assert_close(npy(p.grad), grads[k], what=k, floor=1e-4 if k.endswith("W_k.bias") else 0.0)


@pytest.mark.parametrize("shape,path", [
    (dict(B=37, F=39, D=32, heads=4, A=64, layers=2, residual=True), "whole_block"),    # config 4's shape: one launch
    (dict(B=37, F=39, D=32, heads=4, A=64, layers=2, residual=True), "qkv_inside"),     # the same through core + GEMM + LayerNorm
    (dict(B=21, F=20, D=16, heads=2, A=32, layers=1, residual=True), "qkv_inside"),     # two token tiles
    (dict(B=13, F=48, D=64, heads=1, A=16, layers=1, residual=False), "qkv_inside"),    # full tiles, widest input
    (dict(B=50, F=7, D=48, heads=4, A=64, layers=1, residual=True), "whole_block"),     # one tile, three chunks, 64-lane rows
    (dict(B=19, F=26, D=16, heads=4, A=64, layers=1, residual=False), "whole_block"),   # no residual / LayerNorm
    (dict(B=11, F=33, D=64, heads=4, A=64, layers=1, residual=True), "whole_block"),    # widest rows
    (dict(B=29, F=39, D=40, heads=4, A=64, layers=1, residual=True), "mfma_core"),      # embed_dim % 16 != 0
    (dict(B=29, F=26, D=32, heads=8, A=64, layers=1, residual=True), "vector_core"),    # head_dim 8
])
def test_every_core_variant_vs_oracle(shape, path):
    """The routes of the GEMM path (models/layers/attention.py ``_AttnGemmFn``): the whole forward in one
    kernel (4 heads), projection inside the matrix-core kernel, projection GEMM + matrix-core kernel,
    projection GEMM + vector kernel."""
    from deepfm_amd import _lib
    lib = _lib.load()
    c = shape
    inside = bool(lib.dfm_attention_qkv_core_supported(c["F"], c["D"], c["A"], c["heads"]))
    assert lib.dfm_attention_core_supported(c["F"], c["A"], c["heads"])
    assert inside == (path in ("qkv_inside", "whole_block"))
    if path == "whole_block":
        assert lib.dfm_attention_block_supported(c["F"], c["D"], c["A"], c["heads"])
    if path == "vector_core":
        assert c["A"] // c["heads"] != 16
    _check_vs_oracle(c, whole_block_kernel=(path == "whole_block"))


def test_indivisible_heads_raise():
    from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
    with pytest.raises(ValueError):
        MultiHeadSelfAttention(32, num_heads=3, attention_dim=64)   # attention.py:41-44


def test_attention_cfg4_batch4096_vs_oracle():
    """BASELINE.json config 4 shape (F=39, D=32, 4 heads, A=64) at the full batch."""
    g = load("attn_cfg4")
    att = _module(g)
    params = group(g, "param/")
    rng = np.random.default_rng(11)
    x = rng.standard_normal((4096, 39, 32)).astype(np.float32)
    up = rng.standard_normal((4096, 39, 32)).astype(np.float32)
    t = torch.from_numpy(x).cuda().requires_grad_()
    out = att(t)
    (out * torch.from_numpy(up).cuda()).sum().backward()
    assert_close(npy(out), O.attention_forward(x, params, 4, 1, True), what="out")
    d_x, grads = O.attention_backward(x, params, 4, 1, True, up)
    assert_close(npy(t.grad), d_x, what="d_x")
    for k, p in att.named_parameters():
        assert_close(npy(p.grad), grads[k], what=k, floor=1e-2 if k.endswith("W_k.bias") else 0.0)


def test_attention_deepfm_vs_golden():
    check_model_case("model_attention_deepfm")
