"""FMInteraction on MI355X (reference ``deepfm/models/layers/fm.py:9-23``).

Parameter-free second-order term ``0.5 * sum_d[(sum_f e)^2 - sum_f e^2] -> (B, 1)``;
forward and backward are the HIP kernels ``dfm_fm_forward`` / ``dfm_fm_backward``
(backward: ``d e[b,f,:] = g[b] * (S[b,:] - e[b,f,:])``).
"""

from __future__ import annotations

import torch
import torch.nn as nn

from deepfm_amd import _lib


class _FMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, field_embeddings: torch.Tensor) -> torch.Tensor:
        e = field_embeddings.contiguous()
        B, F, D = e.shape
        out = torch.empty(B, 1, dtype=torch.float32, device=e.device)
        _lib.check(_lib.load().dfm_fm_forward(e.data_ptr(), B, F, D, out.data_ptr(), _lib.stream_handle()))
        ctx.save_for_backward(e)
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        (e,) = ctx.saved_tensors
        B, F, D = e.shape
        g_e = torch.empty_like(e)
        _lib.check(_lib.load().dfm_fm_backward(e.data_ptr(), g_out.contiguous().data_ptr(), B, F, D,
                                               g_e.data_ptr(), _lib.stream_handle()))
        return g_e


class FMInteraction(nn.Module):
    """Input ``(B, F, D)`` float32 on the HIP device, output ``(B, 1)``."""

    def forward(self, field_embeddings: torch.Tensor) -> torch.Tensor:
        if field_embeddings.dim() != 3:
            raise ValueError(f"expected (B, F, D), got {tuple(field_embeddings.shape)}")
        _lib.require_device(field_embeddings, "field_embeddings")
        if field_embeddings.dtype != torch.float32:
            field_embeddings = field_embeddings.float()
        return _FMFn.apply(field_embeddings)
