"""Multi-head self-attention over feature fields on MI355X
(reference ``deepfm/models/layers/attention.py:11-120``).

Same constructor (``ValueError`` when ``attention_dim % num_heads``), same
``forward((B,F,D)) -> (B,F,D)`` and the same ``state_dict`` layout
(``layers.<i>.{W_q,W_k,W_v,W_out}.{weight,bias}``, ``layers.<i>.layer_norm.*`` with
``use_residual``).  ``_AttentionBlock`` keeps ``nn.Linear`` / ``nn.LayerNorm`` parameter
holders (PyTorch default init, like the reference) whose forward is never called: each
block is one fused HIP launch (``dfm_attention_forward`` / ``dfm_attention_backward``).
"""

from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from deepfm_amd import _lib


class _AttentionBlock(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, attention_dim: int, use_residual: bool) -> None:
        super().__init__()
        self.embed_dim, self.attention_dim = embed_dim, attention_dim
        self.num_heads = num_heads
        self.head_dim = attention_dim // num_heads
        self.scale = math.sqrt(self.head_dim)
        self.use_residual = use_residual
        self.W_q = nn.Linear(embed_dim, attention_dim)
        self.W_k = nn.Linear(embed_dim, attention_dim)
        self.W_v = nn.Linear(embed_dim, attention_dim)
        self.W_out = nn.Linear(attention_dim, embed_dim)
        if use_residual:
            self.layer_norm = nn.LayerNorm(embed_dim)

    def _param_list(self):
        ps = [self.W_q.weight, self.W_q.bias, self.W_k.weight, self.W_k.bias, self.W_v.weight,
              self.W_v.bias, self.W_out.weight, self.W_out.bias]
        if self.use_residual:
            ps += [self.layer_norm.weight, self.layer_norm.bias]
        return ps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _AttnFn.apply(self, x, *self._param_list())


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int = 4, attention_dim: int = 64, num_layers: int = 1,
                 use_residual: bool = True) -> None:
        super().__init__()
        self.embed_dim, self.num_heads, self.attention_dim = embed_dim, num_heads, attention_dim
        self.head_dim = attention_dim // num_heads
        self.use_residual = use_residual
        if attention_dim % num_heads != 0:
            raise ValueError(f"attention_dim ({attention_dim}) must be divisible by num_heads ({num_heads})")
        self.layers = nn.ModuleList(
            _AttentionBlock(embed_dim, num_heads, attention_dim, use_residual) for _ in range(num_layers))

    def forward(self, field_embeddings: torch.Tensor) -> torch.Tensor:
        if field_embeddings.dim() != 3 or field_embeddings.shape[2] != self.embed_dim:
            raise ValueError(f"expected (B, F, {self.embed_dim}), got {tuple(field_embeddings.shape)}")
        _lib.require_device(field_embeddings, "field_embeddings")
        x = field_embeddings.float()
        for block in self.layers:
            x = block(x)
        return x


def _ptrs(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, block: _AttentionBlock, x: torch.Tensor, *params):
        x = x.contiguous()
        B, F, D = x.shape
        params = [p.contiguous() for p in params]
        out = torch.empty_like(x)
        _lib.check(_lib.load().dfm_attention_forward(
            x.data_ptr(), B, F, D, block.attention_dim, block.num_heads, int(block.use_residual),
            _ptrs(params), out.data_ptr(), _lib.stream_handle()))
        ctx.block = block
        ctx.save_for_backward(x, *params)
        return out

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        lib = _lib.load()
        block = ctx.block
        x, *params = ctx.saved_tensors
        B, F, D = x.shape
        g_x = torch.empty_like(x)
        grads = [torch.zeros_like(p) for p in params]
        ws = torch.empty(max(lib.dfm_attention_backward_workspace_bytes(B, D, block.attention_dim) // 4, 1),
                         dtype=torch.float32, device=x.device)
        _lib.check(lib.dfm_attention_backward(
            x.data_ptr(), g_out.contiguous().data_ptr(), B, F, D, block.attention_dim, block.num_heads,
            int(block.use_residual), _ptrs(params), g_x.data_ptr(), _ptrs(grads), ws.data_ptr(),
            _lib.stream_handle()))
        return (None, g_x) + tuple(grads)
