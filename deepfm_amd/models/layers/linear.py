"""nn.Linear whose forward/backward run on the exact-fp32 MFMA GEMM (``dfm_gemm_f32``).

Used for the model heads (``output_linear`` / ``cin_linear`` / ``dnn_linear``; reference
deepfm.py:28, xdeepfm.py:33-34, attention_deepfm.py:46).  It IS an ``nn.Linear`` (same
parameters, init and state_dict keys); only ``forward`` differs, and only for float32 HIP
tensors — anything else takes ``nn.Linear.forward``.  Like the DNN layers it can accumulate its
parameter gradients straight into existing ``.grad`` buffers (``direct_grads``, set by the row-sparse
training step); by default it returns them to autograd.
"""

from __future__ import annotations

import torch
import torch.nn as nn

from deepfm_amd.models.layers.dnn import _gemm, _grad_target


_ONES = {}


def ones_column(m: int, device: torch.device) -> torch.Tensor:
    """Cached (m, 1) vector of ones: bias gradients are column sums computed as a GEMM."""
    key = (m, device.type, device.index)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(m, 1, dtype=torch.float32, device=device)
    return t


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, direct=False):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _gemm(x, K, True, weight, K, True, out, M, N, K, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        ctx.direct = direct
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        w_p, b_p = ctx.params
        g = g.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        target = _grad_target if ctx.direct else (lambda p: None)
        tw = target(w_p)
        d_w = None if tw is not None else torch.empty_like(weight)
        _gemm(g, N, False, x, K, False, tw if tw is not None else d_w, N, K, M, accumulate=tw is not None)
        d_b = None
        if b_p is not None:
            tb = target(b_p)
            ones = ones_column(M, g.device)
            d_b = None if tb is not None else torch.empty_like(b_p)
            tgt = (tb if tb is not None else d_b).view(N, 1)
            _gemm(g, N, False, ones, 1, False, tgt, N, 1, M, accumulate=tb is not None)     # column sums of g
        d_x = None
        if ctx.needs_input_grad[0]:
            d_x = torch.empty_like(x)
            _gemm(g, N, True, weight, K, False, d_x, M, K, N)
        return d_x, d_w, d_b, None


class MfmaLinear(nn.Linear):
    direct_grads = False       # see module docstring

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and self.weight.is_contiguous():
            return _LinearFn.apply(x, self.weight, self.bias, self.direct_grads)
        return super().forward(x)
