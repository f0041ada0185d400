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
