"""BCEWithLogitsLoss (mean) as one fused HIP pass (reference ``trainer.py:59, 221``)."""

from __future__ import annotations

import torch

from deepfm_amd import _lib


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        z = logits.contiguous().view(-1)
        y = labels.contiguous().view(-1).float()
        n = z.numel()
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        ws = torch.empty(max(lib.dfm_bce_workspace_bytes(n) // 4, 1), dtype=torch.float32, device=z.device)
        _lib.check(lib.dfm_bce_with_logits(z.data_ptr(), y.data_ptr(), n, loss.data_ptr(), dz.data_ptr(),
                                           ws.data_ptr(), _lib.stream_handle()))
        ctx.save_for_backward(dz)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return (dz * g).view(ctx.shape), None


def bce_with_logits_mean(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``nn.BCEWithLogitsLoss()(logits, labels)`` for float32 HIP tensors of equal numel."""
    _lib.require_device(logits, "logits")
    if logits.numel() != labels.numel():
        raise ValueError(f"logits {tuple(logits.shape)} and labels {tuple(labels.shape)} differ in size")
    return _BCEFn.apply(logits.float(), labels)
