"""Compressed Interaction Network on MI355X (reference ``deepfm/models/layers/cin.py:9-105``).

Same constructor, attributes (``conv_layers``, ``direct_sizes``, ``next_sizes``,
``output_dim``) and ``forward(field_embeddings (B,F,D)) -> (B, output_dim)`` as the
reference; ``conv_layers.<i>`` are ``nn.Conv1d`` *parameter holders* (state_dict keys
``conv_layers.<i>.weight (C, H*F, 1)`` / ``.bias``, PyTorch default init) whose forward is
never called.  The whole stack — outer product, 1x1 convolution, ReLU, split, sum-pool,
concat — runs in ``dfm_cin_forward`` / ``dfm_cin_backward``; the ``(B, H*F, D)`` outer
product the reference materialises per layer is never written to memory.
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from deepfm_amd import _lib


class CIN(nn.Module):
    def __init__(self, num_fields: int, embed_dim: int, layer_sizes: Optional[List[int]] = None,
                 split_half: bool = True) -> None:
        super().__init__()
        layer_sizes = list(layer_sizes or [128, 128])
        self.num_fields, self.embed_dim, self.split_half = num_fields, embed_dim, split_half
        self.layer_sizes = layer_sizes
        self.conv_layers = nn.ModuleList()
        self.direct_sizes: List[int] = []
        self.next_sizes: List[int] = []
        maps = num_fields
        last = len(layer_sizes) - 1
        for i, size in enumerate(layer_sizes):        # bookkeeping of cin.py:45-62
            self.conv_layers.append(nn.Conv1d(maps * num_fields, size, kernel_size=1))
            direct = size // 2 if (split_half and i < last) else size
            nxt = size - direct if (split_half and i < last) else size
            self.direct_sizes.append(direct)
            self.next_sizes.append(nxt)
            maps = nxt
        self.output_dim = sum(self.direct_sizes)

    def forward(self, field_embeddings: torch.Tensor) -> torch.Tensor:
        if field_embeddings.dim() != 3 or field_embeddings.shape[1] != self.num_fields \
                or field_embeddings.shape[2] != self.embed_dim:
            raise ValueError(f"expected (B, {self.num_fields}, {self.embed_dim}), got {tuple(field_embeddings.shape)}")
        _lib.require_device(field_embeddings, "field_embeddings")
        params = []
        for conv in self.conv_layers:
            params += [conv.weight, conv.bias]
        return _CINFn.apply(self, field_embeddings.float(), *params)


def _ptrs(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class _CINFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: CIN, x0: torch.Tensor, *params):
        lib = _lib.load()
        x0 = x0.contiguous()
        B, F, D = x0.shape
        L = len(module.layer_sizes)
        sizes = (C.c_int32 * L)(*module.layer_sizes)
        split = 1 if module.split_half else 0
        weights = [p.contiguous() for p in params[0::2]]
        biases = [p.contiguous() for p in params[1::2]]
        out = torch.empty(B, module.output_dim, dtype=torch.float32, device=x0.device)
        saved = torch.empty(max(lib.dfm_cin_saved_bytes(sizes, L, split, B, F, D) // 4, 1),
                            dtype=torch.float32, device=x0.device)
        ws = torch.empty(max(lib.dfm_cin_forward_workspace_bytes(sizes, L, split, F, D), 16), dtype=torch.uint8,
                         device=x0.device)
        _lib.check(lib.dfm_cin_forward(x0.data_ptr(), B, F, D, _ptrs(weights), _ptrs(biases), sizes, L, split,
                                       out.data_ptr(), saved.data_ptr(), ws.data_ptr(), _lib.stream_handle()))
        ctx.module = module
        ctx.save_for_backward(x0, saved, *weights)
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        lib = _lib.load()
        module = ctx.module
        x0, saved, *weights = ctx.saved_tensors
        B, F, D = x0.shape
        L = len(module.layer_sizes)
        sizes = (C.c_int32 * L)(*module.layer_sizes)
        split = 1 if module.split_half else 0
        g_out = g_out.contiguous()
        g_x0 = torch.empty_like(x0)
        g_w = [torch.zeros_like(w) for w in weights]
        g_b = [torch.zeros(w.shape[0], dtype=torch.float32, device=x0.device) for w in weights]
        ws = torch.empty(max(lib.dfm_cin_backward_workspace_bytes(sizes, L, split, B, F, D) // 4, 1),
                         dtype=torch.float32, device=x0.device)
        _lib.check(lib.dfm_cin_backward(x0.data_ptr(), B, F, D, _ptrs(weights), sizes, L, split,
                                        saved.data_ptr(), g_out.data_ptr(), g_x0.data_ptr(), _ptrs(g_w),
                                        _ptrs(g_b), ws.data_ptr(), _lib.stream_handle()))
        grads = []
        for gw, gb in zip(g_w, g_b):
            grads += [gw, gb]
        return (None, g_x0) + tuple(grads)
