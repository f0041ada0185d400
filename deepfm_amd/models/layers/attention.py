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
from typing import Optional
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
        # projections as GEMMs over the B*F rows (dfm_gemm_f32) + per-(sample, head) core kernel;
        # False (or an unsupported shape) selects the single fused LDS kernel of csrc/attention.hip
        self.gemm_path = True
        # the forward as ONE kernel where its shape allows (4 heads of 16); False: core + GEMM + LayerNorm launches
        self.whole_block_kernel = True

    def adjacent_parameters(self):
        """Parameters an optimizer with ONE flat buffer should lay out back to back, in this order: the
        kernels take W_q | W_k | W_v as one stacked (3A, D) weight, and with this layout the stack is a
        view of the parameters (and of their gradients) instead of a torch.cat per step."""
        return [[self.W_q.weight, self.W_k.weight, self.W_v.weight], [self.W_q.bias, self.W_k.bias, self.W_v.bias]]

    def _param_list(self):
        ps = [self.W_q.weight, self.W_q.bias, self.W_k.weight, self.W_k.bias, self.W_v.weight,
              self.W_v.bias, self.W_out.weight, self.W_out.bias]
        if self.use_residual:
            ps += [self.layer_norm.weight, self.layer_norm.bias]
        return ps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        if self.gemm_path and lib.dfm_attention_core_supported(x.shape[1], self.attention_dim, self.num_heads) \
                and self.embed_dim % 4 == 0 and self.attention_dim % 4 == 0 and self.embed_dim <= 64:
            return _AttnGemmFn.apply(self, x, *self._param_list())
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


def stacked_view(ts) -> "torch.Tensor | None":
    """(sum of rows, cols) view over tensors that lie back to back in one storage (RowSparseAdam lays
    W_q | W_k | W_v out that way, see ``_AttentionBlock.adjacent_parameters``), else None."""
    t0 = ts[0]
    cols = t0.shape[1] if t0.dim() == 2 else 1
    nxt, store = t0.data_ptr(), t0.untyped_storage().data_ptr()
    for t in ts:
        # same storage, not merely neighbouring allocations: the view must stay inside one storage
        if not t.is_contiguous() or t.data_ptr() != nxt or t.untyped_storage().data_ptr() != store \
                or t.dtype != t0.dtype or (t.shape[1] if t.dim() == 2 else 1) != cols:
            return None
        nxt += t.numel() * t.element_size()
    rows = sum(t.shape[0] for t in ts)
    return torch.as_strided(t0, (rows, cols) if t0.dim() == 2 else (rows,), (cols, 1) if t0.dim() == 2 else (1,))


def _ptrs(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _weight_grad(g: torch.Tensor, x: torch.Tensor, rows: int, n1: int, n2: int, d_w: torch.Tensor,
                 d_b: torch.Tensor, finish: Optional[list] = None) -> bool:
    """dW = g^T x and db = column sums of g in one streamed pass (csrc/gemm_skinny.hip) when the shape
    is one the kernel takes; False -> the caller uses the general GEMM (+ a ones-column GEMM).
    ``finish``: a list -> only the streamed pass runs and the reduction of its partial sums is appended
    as a job for ``_finish_partials`` (one launch for all of a block's reductions)."""
    lib = _lib.load()
    ws_bytes = lib.dfm_weight_grad_workspace_bytes(rows, n1, n2)
    if not ws_bytes:
        return False
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=g.device)
    if finish is None:
        _lib.check(lib.dfm_weight_grad_f32(g.data_ptr(), n1, x.data_ptr(), n2, rows, n1, n2, d_w.data_ptr(), n2,
                                           d_b.data_ptr(), 0, ws.data_ptr(), _lib.stream_handle()))
        return True
    _lib.check(lib.dfm_weight_grad_partials_f32(g.data_ptr(), n1, x.data_ptr(), n2, rows, n1, n2, ws.data_ptr(),
                                                _lib.stream_handle()))
    finish.append(dict(kind=0, blocks=lib.dfm_weight_grad_partial_blocks(rows), n1=n1, n2=n2, accumulate=0,
                       partial=ws, out_w=d_w.data_ptr(), out_b=d_b.data_ptr(), ldw=n2))
    return True


def _weight_grad_pair(ga, xa, n1a, n2a, dwa, dba, gb, xb, n1b, n2b, dwb, dbb, rows: int, finish: list) -> bool:
    """Two ``_weight_grad`` streamed passes over the same ``rows`` as one launch; False when the library has no joint
    kernel for the two shapes (nothing was enqueued)."""
    lib = _lib.load()
    wa, wb = lib.dfm_weight_grad_workspace_bytes(rows, n1a, n2a), lib.dfm_weight_grad_workspace_bytes(rows, n1b, n2b)
    if not wa or not wb:
        return False
    ws_a = torch.empty(wa // 4, dtype=torch.float32, device=ga.device)
    ws_b = torch.empty(wb // 4, dtype=torch.float32, device=ga.device)
    rc = lib.dfm_weight_grad_partials_pair_f32(ga.data_ptr(), n1a, xa.data_ptr(), n2a, n1a, n2a, ws_a.data_ptr(),
                                               gb.data_ptr(), n1b, xb.data_ptr(), n2b, n1b, n2b, ws_b.data_ptr(), rows,
                                               _lib.stream_handle())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc)
    blocks = lib.dfm_weight_grad_partial_blocks(rows)
    for n1, n2, ws, dw, db in ((n1a, n2a, ws_a, dwa, dba), (n1b, n2b, ws_b, dwb, dbb)):
        finish.append(dict(kind=0, blocks=blocks, n1=n1, n2=n2, accumulate=0, partial=ws, out_w=dw.data_ptr(),
                           out_b=db.data_ptr(), ldw=n2))
    return True


def _finish_partials(jobs: list) -> None:
    """The deferred reductions of a backward pass (weight-gradient partials, LayerNorm partials) in ONE launch."""
    if not jobs:
        return
    arr = (_lib.PartialJob * len(jobs))()
    for a, j in zip(arr, jobs):
        a.kind, a.blocks, a.n1, a.n2, a.accumulate = j["kind"], j["blocks"], j["n1"], j["n2"], j["accumulate"]
        a.partial, a.out_w, a.out_b, a.ldw = j["partial"].data_ptr(), j["out_w"], j["out_b"], j["ldw"]
    _lib.check(_lib.load().dfm_partials_finish(arr, len(jobs), _lib.stream_handle()))


class _AttnGemmFn(torch.autograd.Function):
    """One _AttentionBlock as: QKV GEMM -> attention core -> output GEMM -> (+x, LayerNorm)."""

    @staticmethod
    def forward(ctx, block: _AttentionBlock, x: torch.Tensor, *params):
        from deepfm_amd.models.layers.dnn import _gemm
        lib = _lib.load()
        x = x.contiguous()
        B, F, D = x.shape
        A, H = block.attention_dim, block.num_heads
        M = B * F
        wq, bq, wk, bk, wv, bv, wo, bo = (p.contiguous() for p in params[:8])
        w_qkv = stacked_view([wq, wk, wv])                          # (3A, D): a view when the optimizer laid them out so
        b_qkv = stacked_view([bq, bk, bv])
        if w_qkv is None or b_qkv is None:
            w_qkv, b_qkv = torch.cat([wq, wk, wv], dim=0), torch.cat([bq, bk, bv], dim=0)
        X = x.view(M, D)
        o = torch.empty(M, A, dtype=torch.float32, device=x.device)
        aligned = X.data_ptr() % 16 == 0 and w_qkv.data_ptr() % 16 == 0 and b_qkv.data_ptr() % 16 == 0
        # out_into (fused training step, on its ctx stand-in): (buffer, floats between samples) — the block's
        # output goes straight into a wider per-sample layout (the DNN's concatenated input)
        into = getattr(ctx, "out_into", None) if block.use_residual else None
        if block.whole_block_kernel and aligned and wo.data_ptr() % 16 == 0 \
                and lib.dfm_attention_block_supported(F, D, A, H):
            # ONE launch: projection, softmax(QK^T)V, W_out, bias, residual LayerNorm (csrc/attention_mfma.hip)
            y = torch.empty(M, D, dtype=torch.float32, device=x.device)
            gamma = beta = stats = None
            if block.use_residual:
                gamma, beta = params[8].contiguous(), params[9].contiguous()
                stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
                out = into[0] if into is not None else torch.empty(M, D, dtype=torch.float32, device=x.device)
            else:
                out = y
            # x_copy_into (fused training step): (data_ptr, row stride) of a second home of the block's input rows
            xc = getattr(ctx, "x_copy_into", None)
            _lib.check(lib.dfm_attention_block_forward(
                X.data_ptr(), w_qkv.data_ptr(), b_qkv.data_ptr(), wo.data_ptr(), bo.data_ptr(), _lib.ptr(gamma),
                _lib.ptr(beta), float(block.layer_norm.eps) if block.use_residual else 0.0, B, F, D, A, H, o.data_ptr(),
                y.data_ptr(), out.data_ptr(), _lib.ptr(stats), into[1] if into is not None else 0,
                xc[0] if xc else None, xc[1] if xc else 0, _lib.stream_handle()))
            ctx.x_copied = xc is not None
            ctx.block, ctx.dims = block, (B, F, D, A, H)
            ctx.save_for_backward(X, None, o, y, stats, w_qkv, wo, gamma, b_qkv)
            return out if into is not None else out.view(B, F, D)
        # projection inside the core kernel where its shape allows: the (M, 3A) Q|K|V is never materialised
        inside = bool(lib.dfm_attention_qkv_core_supported(F, D, A, H)) and aligned
        if inside:
            qkv = None
            _lib.check(lib.dfm_attention_qkv_core_forward(X.data_ptr(), w_qkv.data_ptr(), b_qkv.data_ptr(), B, F, D, A,
                                                          H, o.data_ptr(), _lib.stream_handle()))
        else:
            qkv = torch.empty(M, 3 * A, dtype=torch.float32, device=x.device)
            _gemm(X, D, True, w_qkv, D, True, qkv, M, 3 * A, D, bias=b_qkv)
            _lib.check(lib.dfm_attention_core_forward(qkv.data_ptr(), B, F, A, H, o.data_ptr(),
                                                      _lib.stream_handle()))
        y = torch.empty(M, D, dtype=torch.float32, device=x.device)
        _gemm(o, A, True, wo, A, True, y, M, D, A, bias=bo)
        stats = None
        if block.use_residual:
            gamma, beta = params[8].contiguous(), params[9].contiguous()
            out = into[0] if into is not None else torch.empty(M, D, dtype=torch.float32, device=x.device)
            stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
            _lib.check(lib.dfm_layernorm_forward(y.data_ptr(), X.data_ptr(), M, D, gamma.data_ptr(), beta.data_ptr(),
                                                 float(block.layer_norm.eps), out.data_ptr(), stats.data_ptr(),
                                                 F if into is not None else 0, into[1] if into is not None else 0,
                                                 _lib.stream_handle()))
            if into is not None:
                ctx.block, ctx.dims = block, (B, F, D, A, H)
                ctx.save_for_backward(X, qkv, o, y, stats, w_qkv, wo, gamma, b_qkv)
                return out
        else:
            gamma = beta = None
            out = y
        ctx.block, ctx.dims = block, (B, F, D, A, H)
        ctx.save_for_backward(X, qkv, o, y, stats, w_qkv, wo, gamma, b_qkv)
        return out.view(B, F, D)

    @staticmethod
    def backward(ctx, g_out: torch.Tensor):
        from deepfm_amd.models.layers.dnn import _gemm
        lib = _lib.load()
        block = ctx.block
        B, F, D, A, H = ctx.dims
        M = B * F
        X, qkv, o, y, stats, w_qkv, wo, gamma, b_qkv = ctx.saved_tensors
        dev = X.device
        # direct (set by the fused training step on its ctx stand-in): parameter gradients are written
        # straight into the (zeroed) .grad views of the optimizer's flat buffer — no temporaries, no adds
        direct = getattr(ctx, "direct", False)
        gq = gb = None
        if direct:
            gq = stacked_view([block.W_q.weight.grad, block.W_k.weight.grad, block.W_v.weight.grad])
            gb = stacked_view([block.W_q.bias.grad, block.W_k.bias.grad, block.W_v.bias.grad])
            direct = gq is not None and gb is not None and block.W_out.weight.grad.is_contiguous()
        # g_from (fused training step): the incoming gradient lives in a wider per-sample layout
        # (floats between samples); only the residual LayerNorm's backward can read it that way
        g_stride = getattr(ctx, "g_from", 0) if block.use_residual else 0
        g = g_out if g_stride else g_out.contiguous().view(M, D)
        from deepfm_amd.models.layers.linear import ones_column
        grads = []
        finish: list = []          # deferred reductions of this block's backward: ONE launch at its end
        if block.use_residual:
            g_y = torch.empty(M, D, dtype=torch.float32, device=dev)
            if direct:                           # accumulated into: zero at this point of the step
                d_gamma, d_beta = block.layer_norm.weight.grad, block.layer_norm.bias.grad
            else:
                d_gamma = torch.zeros(D, dtype=torch.float32, device=dev)
                d_beta = torch.zeros(D, dtype=torch.float32, device=dev)
            ws = torch.empty(max(lib.dfm_layernorm_workspace_bytes(M, D) // 4, 1), dtype=torch.float32, device=dev)
            # d gamma / d beta: the partial planes stay in ws, added by the block's one finish launch below
            _lib.check(lib.dfm_layernorm_backward(g.data_ptr(), y.data_ptr(), X.data_ptr(), stats.data_ptr(), M, D,
                                                  gamma.data_ptr(), g_y.data_ptr(), None, None, ws.data_ptr(),
                                                  F if g_stride else 0, g_stride, _lib.stream_handle()))
            finish.append(dict(kind=1, blocks=lib.dfm_layernorm_partial_blocks(M), n1=D, n2=0, accumulate=1,
                               partial=ws, out_w=d_gamma.data_ptr(), out_b=d_beta.data_ptr(), ldw=0))
        else:
            g_y = g
        d_wo = block.W_out.weight.grad if direct else torch.empty(D, A, dtype=torch.float32, device=dev)
        d_bo = block.W_out.bias.grad.view(D, 1) if direct else torch.empty(D, 1, dtype=torch.float32, device=dev)
        d_qkv = torch.empty(M, 3 * A, dtype=torch.float32, device=dev)
        d_wqkv = gq if direct else torch.empty(3 * A, D, dtype=torch.float32, device=dev)
        d_bqkv = gb.view(3 * A, 1) if direct else torch.empty(3 * A, 1, dtype=torch.float32, device=dev)
        whole = (qkv is None and block.whole_block_kernel
                 and lib.dfm_attention_block_supported(F, D, A, H) == 1)
        if whole:                                                                # dO, the core and dX in one launch
            # grad_tail (fused training step, first block): dict(out, g_flat, ld_flat, g_fm, fm_sum) — the other
            # gradients of the field embeddings, added in the kernel's one store of d x (-> ctx.tail_done)
            tail = getattr(ctx, "grad_tail", None)
            d_x = tail["out"].view(M, D) if tail else torch.empty(M, D, dtype=torch.float32, device=dev)
            _lib.check(lib.dfm_attention_block_backward(
                X.data_ptr(), w_qkv.data_ptr(), b_qkv.data_ptr(), wo.data_ptr(), g_y.data_ptr(),
                int(block.use_residual), B, F, D, A, H, d_qkv.data_ptr(), d_x.data_ptr(),
                tail["g_flat"] if tail else None, tail["ld_flat"] if tail else 0,
                tail["g_fm"] if tail else None, tail["fm_sum"] if tail else None, _lib.stream_handle()))
            ctx.tail_done = tail is not None
        else:
            d_o = torch.empty(M, A, dtype=torch.float32, device=dev)
            _gemm(g_y, D, True, wo, A, False, d_o, M, A, D)                      # dO = g_y Wo
            if qkv is None:                                                      # Q, K, V recomputed from X in-kernel
                _lib.check(lib.dfm_attention_qkv_core_backward(X.data_ptr(), w_qkv.data_ptr(), b_qkv.data_ptr(),
                                                               d_o.data_ptr(), B, F, D, A, H, d_qkv.data_ptr(),
                                                               _lib.stream_handle()))
            else:
                _lib.check(lib.dfm_attention_core_backward(qkv.data_ptr(), d_o.data_ptr(), B, F, A, H,
                                                           d_qkv.data_ptr(), _lib.stream_handle()))
        # dWqkv = dQKV^T X and dWo = g_y^T O (+ their bias gradients): both streamed passes in one launch when the pair
        # of shapes has a joint kernel, their reductions and the LayerNorm's in the block's one finish launch
        if not _weight_grad_pair(d_qkv, X, 3 * A, D, d_wqkv, d_bqkv, g_y, o, D, A, d_wo, d_bo, M, finish):
            if not _weight_grad(g_y, o, M, D, A, d_wo, d_bo, finish):
                _gemm(g_y, D, False, o, A, False, d_wo, D, A, M)
                _gemm(g_y, D, False, ones_column(M, dev), 1, False, d_bo, D, 1, M)
            if not _weight_grad(d_qkv, X, M, 3 * A, D, d_wqkv, d_bqkv, finish):
                _gemm(d_qkv, 3 * A, False, X, D, False, d_wqkv, 3 * A, D, M)
                _gemm(d_qkv, 3 * A, False, ones_column(M, dev), 1, False, d_bqkv, 3 * A, 1, M)
        _finish_partials(finish)
        if whole:
            pass
        elif block.use_residual:
            d_x = g_y                                                            # residual branch, then +=
            _gemm(d_qkv, 3 * A, True, w_qkv, D, False, d_x, M, D, 3 * A, accumulate=True)
        else:
            d_x = torch.empty(M, D, dtype=torch.float32, device=dev)
            _gemm(d_qkv, 3 * A, True, w_qkv, D, False, d_x, M, D, 3 * A)
        if direct:
            return (None, d_x.view(B, F, D))
        d_bqkv = d_bqkv.view(-1)
        grads = [d_wqkv[:A], d_bqkv[:A], d_wqkv[A:2 * A], d_bqkv[A:2 * A], d_wqkv[2 * A:], d_bqkv[2 * A:],
                 d_wo, d_bo.view(-1)]
        if block.use_residual:
            grads += [d_gamma, d_beta]
        return (None, d_x.view(B, F, D)) + tuple(grads)


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
