"""DNN tower (reference ``deepfm/models/layers/dnn.py:9-59``).

Not one of the four hot-path layers (row f-2): same constructor, ``mlp``
``nn.Sequential`` and ``dnn.mlp.<i>`` state_dict layout — ``[Linear, (BatchNorm1d), activation,
Dropout] * n``.  Under autograd, on an MI355X in training mode with BatchNorm + ReLU (the reference
default), a layer runs as ``_LinearBnReluDropoutFn``: the three Linear GEMMs on the exact-fp32 MFMA
kernel ``dfm_gemm_f32`` (no transposes, no precision loss) and the BatchNorm -> ReLU -> Dropout chain
on the fused HIP kernels ``dfm_bn_relu_dropout_forward/backward``; every other configuration
(eval mode, other activations, no BatchNorm) uses the plain ``nn.Sequential``.  The training step of
DeepFM does not come through here at all: ``deepfm_amd.training.fused_step`` drives the same
parameters with the launch-fused tower kernels of ``csrc/tower.hip``.
"""

from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from deepfm_amd import _lib


def _gemm(a, lda, a_kc, b, ldb, b_kc, c, m, n, k, bias=None, accumulate=False):
    """c (m, n) (+)= a(m, k) * b(n, k) (+ bias): exact fp32 on the matrix cores (csrc/gemm_f32.hip)."""
    lib = _lib.load()
    ws_bytes = lib.dfm_gemm_workspace_bytes(m, n, k)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=c.device) if ws_bytes else None
    _lib.check(lib.dfm_gemm_f32(a.data_ptr(), lda, int(a_kc), b.data_ptr(), ldb, int(b_kc), c.data_ptr(),
                                c.stride(0), m, n, k, _lib.ptr(bias), int(accumulate), _lib.ptr(ws),
                                _lib.stream_handle()))


def _grad_target(p: torch.Tensor):
    """An existing, contiguous .grad buffer we may accumulate into directly (the row-sparse
    optimizer makes every .grad a view of one flat buffer), else None."""
    g = p.grad
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32) else None


class _LinearBnReluDropoutFn(torch.autograd.Function):
    """One DNN layer: Linear (dfm_gemm_f32) -> BatchNorm1d(train) -> ReLU -> Dropout (fused HIP).

    Backward writes parameter gradients straight into existing ``.grad`` buffers
    (``addmm_`` with beta = 1 for dW, ``+=`` inside the BN kernels) and returns ``None`` for
    them, which removes one AccumulateGrad add + one zero-fill launch per parameter.  The
    Linear bias in front of a training-mode BatchNorm has an identically-zero gradient (BN
    subtracts the batch mean), so none is computed."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn: nn.BatchNorm1d, p: float, seed, salt: int, direct: bool):
        lib = _lib.load()
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        z = torch.empty(M, N, dtype=torch.float32, device=x.device)
        out = torch.empty_like(z)
        stats = torch.empty(2, N, dtype=torch.float32, device=z.device)
        track = bn.track_running_stats and bn.running_mean is not None
        rm = bn.running_mean.data_ptr() if track else None
        rv = bn.running_var.data_ptr() if track else None
        nb = bn.num_batches_tracked.data_ptr() if track else None
        if N % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (gamma, beta)):
            # two launches (csrc/tower.hip): GEMM + per-tile column statistics, then merge + normalise
            ws = torch.empty(max(lib.dfm_linear_bn_workspace_bytes(M, N) // 4, 1), dtype=torch.float32, device=z.device)
            _lib.check(lib.dfm_linear_bn_forward(x.data_ptr(), K, weight.data_ptr(), _lib.ptr(bias), M, N, K,
                                                 z.data_ptr(), ws.data_ptr(), _lib.stream_handle()))
            _lib.check(lib.dfm_bn_relu_dropout_apply(
                z.data_ptr(), M, N, ws.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), rm, rv, nb,
                float(bn.momentum), float(bn.eps), float(p), _lib.ptr(seed), salt, out.data_ptr(),
                _lib.stream_handle()))
        else:
            _gemm(x, K, True, weight, K, True, z, M, N, K, bias=bias)                 # z = x W^T + b
            ws = torch.empty(max(lib.dfm_bn_workspace_bytes(M, N) // 4, 1), dtype=torch.float32, device=z.device)
            _lib.check(lib.dfm_bn_relu_dropout_forward(
                z.data_ptr(), M, N, gamma.data_ptr(), beta.data_ptr(), rm, rv, nb, float(bn.momentum), float(bn.eps),
                float(p), _lib.ptr(seed), salt, out.data_ptr(), stats.data_ptr(), ws.data_ptr(), _lib.stream_handle()))
        ctx.save_for_backward(x, weight, z, gamma, beta, stats)
        ctx.p, ctx.seed, ctx.salt, ctx.direct = p, seed, salt, direct
        ctx.params = (weight, bias, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        x, weight, z, gamma, beta, stats = ctx.saved_tensors
        w_p, b_p, gamma_p, beta_p = ctx.params
        M, N = z.shape
        dz = torch.empty_like(z)
        # the write-into-.grad shortcut only when the caller asked for it (the row-sparse training step,
        # whose optimizer owns the flat gradient buffer): torch.autograd.grad() / hooks get real gradients
        target = _grad_target if ctx.direct else (lambda p: None)
        tg, tb = target(gamma_p), target(beta_p)
        d_gamma = tg if tg is not None else torch.zeros_like(gamma)
        d_beta = tb if tb is not None else torch.zeros_like(beta)
        ws = torch.empty(max(lib.dfm_bn_workspace_bytes(M, N) // 4, 1), dtype=torch.float32, device=z.device)
        _lib.check(lib.dfm_bn_relu_dropout_backward(
            g_out.contiguous().data_ptr(), z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, N,
            float(ctx.p), _lib.ptr(ctx.seed), ctx.salt, dz.data_ptr(), d_gamma.data_ptr(), d_beta.data_ptr(),
            ws.data_ptr(), _lib.stream_handle()))
        K = x.shape[1]
        tw = target(w_p)
        d_w = None if tw is not None else torch.empty_like(weight)
        # dW (N, K) (+)= dz^T x : both operands strided over the batch (the reduction index)
        _gemm(dz, N, False, x, K, False, tw if tw is not None else d_w, N, K, M, accumulate=tw is not None)
        # d bias == 0 exactly (BatchNorm removes the batch mean): leave an existing buffer as is
        d_b = None if target(b_p) is not None else torch.zeros_like(b_p)
        d_x = None
        if ctx.needs_input_grad[0]:
            d_x = torch.empty_like(x)
            _gemm(dz, N, True, weight, K, False, d_x, M, K, N)                      # dx = dz W
        return (d_x, d_w, d_b, None if tg is not None else d_gamma, None if tb is not None else d_beta,
                None, None, None, None, None)


class DNN(nn.Module):
    ACTIVATIONS = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "gelu": nn.GELU, "tanh": nn.Tanh}

    def __init__(self, input_dim: int, hidden_units: List[int], activation: str = "relu",
                 dropout: float = 0.1, use_batch_norm: bool = True) -> None:
        super().__init__()
        if not hidden_units:
            raise ValueError("hidden_units must be non-empty")
        try:
            act = self.ACTIVATIONS[activation.lower()]
        except KeyError:
            raise ValueError(f"Unknown activation: {activation}. Choose from {list(self.ACTIVATIONS)}") from None
        stack: List[nn.Module] = []
        width = input_dim
        for units in hidden_units:
            stack.append(nn.Linear(width, units))
            if use_batch_norm:
                stack.append(nn.BatchNorm1d(units))
            stack += [act(), nn.Dropout(p=dropout)]
            width = units
        self.mlp = nn.Sequential(*stack)
        self.output_dim = width
        self.fused = True          # use the fused HIP BatchNorm/ReLU/Dropout kernels when eligible
        self._fusable = use_batch_norm and activation.lower() == "relu"
        self._n_layers = len(hidden_units)
        self._seed = None
        # set by the row-sparse training step: backward accumulates parameter gradients straight into the
        # optimizer's flat .grad views and returns None for them (no AccumulateGrad add / zero-fill launches)
        self.direct_grads = False

    def _fused_ok(self, x: torch.Tensor) -> bool:
        if not (self.fused and self._fusable and self.training and x.is_cuda and x.dtype == torch.float32
                and x.dim() == 2 and x.shape[0] > 1):
            return False
        bn = self.mlp[1]
        return bn.momentum is not None and bn.affine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._fused_ok(x):
            return self.mlp(x)
        if self._seed is None or self._seed.device != x.device:
            self._seed = torch.randint(1, 2 ** 40, (1,), dtype=torch.int64, device=x.device)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                self._seed += 7919 * torch.distributed.get_rank()      # replicas share parameters, not masks
        self._seed.add_(1)
        # backward regenerates the dropout mask from the seed: with dropout on, each forward keeps its own
        # copy, so a second forward before the backward (gradient accumulation) cannot change the mask
        any_drop = any(self.mlp[4 * i + 3].p > 0 for i in range(self._n_layers))
        seed = self._seed.clone() if any_drop else self._seed
        h = x
        for i in range(self._n_layers):
            lin, bn, _, drop = (self.mlp[4 * i + j] for j in range(4))
            h = _LinearBnReluDropoutFn.apply(h, lin.weight, lin.bias, bn.weight, bn.bias, bn, drop.p,
                                             seed, i, self.direct_grads)
        return h
