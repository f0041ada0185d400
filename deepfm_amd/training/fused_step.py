"""Training steps on the fused tower kernels (csrc/tower.hip): no autograd, ~25 launches.

Same step as ``RowSparseTrainStep`` (reference ``Trainer._train_epoch`` body, trainer.py:212-240), but
the DNN tower, the head and what flows back into the embeddings run on ``dfm_linear_bn_forward`` /
``dfm_bn_relu_dropout_apply`` / ``dfm_head_bce`` / ``dfm_bn_backward_apply`` / ``dfm_linear_backward``
with hand-written backward wiring:

    gather (+ fm value, + S = sum_f e)                                        1 launch (graph node)
    interaction layer forward (xDeepFM: CIN + its head)                       model specific
    per layer: GEMM + batch statistics, BN/ReLU/Dropout apply                 2 launches
    head: logits + BCE + d logits + head grads + last BN's mask               1 launch
    interaction layer backward (xDeepFM: CIN head, CIN)                       model specific
    per layer (top down): BN backward apply, [dW | dx + lower BN mask / FM / addend]   2 launches
    embedding backward (dense fields + row gradients), optimizer              unchanged

On an MI355X every dependent launch costs ~4.5 us, and the autograd path needs ~65 of them.
``FusedDeepFMStep``  (deepfm.py:30-42): logits = (fo + fm) + output_linear(dnn(flat)).
``FusedXDeepFMStep`` (xdeepfm.py:36-48): logits = (fo + cin_linear(cin(fe))) + dnn_linear(dnn(flat)).
Eligible: the reference-default tower (BatchNorm + ReLU), hidden sizes multiples of 4 (last one a
multiple of 32, <= 256), uniform embedding schema in ``rowsparse`` mode, training mode.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from deepfm_amd import _lib
from deepfm_amd.training.rowsparse import RowSparseAdam
from deepfm_amd.training.step import RowSparseTrainStep


def _zeros_bytes(nbytes: int, device) -> torch.Tensor:
    return torch.zeros(max((nbytes + 3) // 4, 1), dtype=torch.int32, device=device)


def _tower_ok(model) -> bool:
    if not model.training:
        return False
    dnn = getattr(model, "dnn", None)
    if dnn is None or not getattr(dnn, "_fusable", False):
        return False
    widths = [dnn.mlp[4 * i].out_features for i in range(dnn._n_layers)]
    if any(w % 4 for w in widths) or widths[-1] % 32 or widths[-1] > 256:
        return False
    bn = dnn.mlp[1]
    if bn.momentum is None or not bn.affine:
        return False
    return dnn.mlp[0].in_features % 4 == 0 and model.embedding.grad_mode == "rowsparse"


class _FusedTowerStep(RowSparseTrainStep):
    """Tower + head + optimizer wiring shared by the fused steps; subclasses say what else feeds the
    logit (``_interaction_forward`` -> per-sample scalar) and the embeddings' gradient
    (``_interaction_backward`` -> the epilogue of the first Linear's d input)."""

    head_name = "output_linear"
    slabs_travel = False         # True: _embedding_backward consumes self._slab_refs (training/sharded.py)

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        super().__init__(model, optimizer, batch_size, use_graph)
        self._slab_refs = None
        # DENSE-field Linear gradients over 4 batch slices, added with the tower's d-weight slabs: as one
        # slice the 65 workgroups of that part walk the whole batch (a ~11 us latency chain)
        self._dense_parts = 4 if (optimizer.n_l2 > 0 and self.n_dense > 0) else 0
        self._dense_partial = (torch.zeros(self._dense_parts * optimizer.n_l2, dtype=torch.float32,
                                           device=optimizer.device) if self._dense_parts else None)
        if not self.eligible(model):
            raise ValueError(f"{type(self).__name__}: model/configuration not eligible (use RowSparseTrainStep)")
        lib = _lib.load()
        dev, B = optimizer.device, batch_size
        dnn = model.dnn
        self.L = dnn._n_layers
        f32 = dict(dtype=torch.float32, device=dev)
        F, D = self.fe.shape[1], self.fe.shape[2]
        self.x0 = self._tower_input()                      # (B, K) input of the first Linear
        self.g_fe = torch.empty(B, F, D, **f32)
        self.g_x0 = self._tower_input_grad()               # where the first Linear's d input goes
        self.logits = torch.empty(B, **f32)
        self.g_logits = torch.empty(B, 1, **f32)
        self.lin, self.bn, self.drop_p = [], [], []
        self.z, self.a, self.stats, self.dy = [], [], [], []
        self.ws_fwd, self.ws_bn, self.ws_lin = [], [], []
        for i in range(self.L):
            lin, bn, _, drop = (dnn.mlp[4 * i + j] for j in range(4))
            n, k = lin.out_features, lin.in_features
            self.lin.append(lin); self.bn.append(bn); self.drop_p.append(float(drop.p))
            self.z.append(torch.empty(B, n, **f32))
            self.a.append(torch.empty(B, n, **f32))
            self.dy.append(torch.empty(B, n, **f32))
            self.stats.append(torch.empty(2, n, **f32))
            self.ws_fwd.append(_zeros_bytes(lib.dfm_linear_bn_workspace_bytes(B, n), dev))
            self.ws_bn.append(_zeros_bytes(lib.dfm_bn_bwd_workspace_bytes(B, n), dev))
            self.ws_lin.append(_zeros_bytes(lib.dfm_linear_backward_workspace_bytes(B, n, k), dev))
        # dfm_tower_set_mode(2): the tower's GEMMs on the bf16 pipe with the exact three-way operand split
        # (csrc/gemm_x6.h).  Weights are split once per step (one launch), activations and d z by the kernels that
        # produce them; the fp32 copies of a / d z are not written at all.  Shapes the planes cannot hold run mode 0.
        self.tower_mode = int(lib.dfm_tower_get_mode())
        self.x6 = self.tower_mode == 2 and all(
            lib.dfm_tower_x6_supported(B, l.out_features, l.in_features) for l in self.lin)
        if self.x6:
            def planes(rows, contraction):
                return _zeros_bytes(lib.dfm_planes_bytes(rows, contraction), dev)
            self.w_f = [planes(l.out_features, l.in_features) for l in self.lin]     # z = x W^T
            self.w_s = [planes(l.in_features, l.out_features) for l in self.lin]     # d x = d z W
            self.a_f = [planes(B, l.out_features) for l in self.lin[:-1]]            # next layer's z
            self.a_s = [planes(l.out_features, B) for l in self.lin[:-1]]            # next layer's d W
            self.dz_f = [planes(B, l.out_features) for l in self.lin]                # d x
            self.dz_s = [planes(l.out_features, B) for l in self.lin]                # d W
            self.ws_lin = [_zeros_bytes(lib.dfm_linear_backward_x6_workspace_bytes(B, l.out_features, l.in_features), dev)
                           for l in self.lin]
            self._split_jobs = (_lib.SplitJob * self.L)()
            for i, l in enumerate(self.lin):
                j = self._split_jobs[i]
                j.src, j.rows, j.cols = l.weight.data_ptr(), l.out_features, l.in_features
                j.planes_f, j.planes_s = self.w_f[i].data_ptr(), self.w_s[i].data_ptr()
        self.seed = torch.randint(1, 2 ** 40, (1,), dtype=torch.int64, device=dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # replicas are built from one seed (identical parameters) but must not share dropout masks
            self.seed += 7919 * torch.distributed.get_rank()
        optimizer.seed_tick = self.seed          # advanced by the optimizer's norm-finalize kernel
        self.rowplan_side_stream = os.environ.get("DFM_ROWPLAN_SIDE_STREAM") == "1"
        self.head = getattr(model, self.head_name)
        for p in list(dnn.parameters()) + list(self.head.parameters()):
            if p.grad is None or not p.grad.is_contiguous():
                raise RuntimeError("fused steps need RowSparseAdam's flat gradient views on every dense parameter")

    # ------------------------------------------------------------------ model-specific hooks
    def _tower_input(self) -> torch.Tensor:
        """Input of the first Linear: the flat embeddings (same bytes as field_embeddings)."""
        return self.fe.view(self.B, -1)

    def _tower_input_grad(self) -> torch.Tensor:
        return self.g_fe

    def _finish_embedding_grad(self) -> None:
        """After the tower's backward: whatever is still missing in ``g_fe`` (default: nothing)."""

    def _interaction_forward(self) -> Optional[torch.Tensor]:
        """Runs the model's interaction layer; returns the per-sample scalar added to the logit
        next to first_order ((B,) tensor) or None."""
        raise NotImplementedError

    def _interaction_backward(self) -> Optional[_lib.FmBwd]:
        """Called right after the head (g_logits is known): enqueues the interaction layer's backward
        and returns what the first Linear's d-input epilogue adds to d field_embeddings."""
        raise NotImplementedError

    # ------------------------------------------------------------------ pieces
    def _bn_ctx(self, i: int) -> _lib.BnBwd:
        bn = self.bn[i]
        c = _lib.BnBwd()
        c.z, c.mean_rstd = self.z[i].data_ptr(), self.stats[i].data_ptr()
        c.gamma, c.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
        c.dy = self.dy[i].data_ptr()
        c.g_gamma, c.g_beta = bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr()
        c.seed = self.seed.data_ptr()
        c.workspace = self.ws_bn[i].data_ptr()
        c.p_drop, c.salt = self.drop_p[i], i
        return c

    def _body_a(self) -> None:
        lib, st, B = _lib.load(), _lib.stream_handle(), self.B
        self.opt.zero_grad()
        cur = torch.cuda.current_stream()
        # The row plan (needs only the ids) runs IN LINE.  On a side stream it overlapped the forward
        # "for free" — and cost more than its own 27 us: the concurrent sort slowed the widest GEMM by
        # 7 us, and the fork/join put the graph on two hardware queues with ~10 us per cross-queue edge
        # (measured 0.253 ms in line vs 0.255 overlapped when the sort still took 38 us).  Round 2 tried ONE
        # fork / join per four-step graph with all four plans built ahead on the side branch: 0.227 ms/step
        # against 0.217 in line — a graph that spans two queues is slower as a whole.
        inline = not self.rowplan_side_stream
        if inline:
            self._build_rowplan()
        else:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self._build_rowplan()
        # ---- forward ----
        extra = self._interaction_forward()
        x = self.x0
        head = self.head
        x6 = self.x6
        if x6:      # this step's weights as planes (the optimizer wrote them at the end of the last step)
            _lib.check(lib.dfm_split_planes(self._split_jobs, self.L, st))
        for i in range(self.L):
            lin, bn = self.lin[i], self.bn[i]
            n, k = lin.out_features, lin.in_features
            track = bn.track_running_stats and bn.running_mean is not None
            if x6:
                _lib.check(lib.dfm_linear_bn_forward_x6(
                    x.data_ptr() if i == 0 else None, k, None if i == 0 else self.a_f[i - 1].data_ptr(),
                    self.w_f[i].data_ptr(), _lib.ptr(lin.bias), B, n, k, self.z[i].data_ptr(),
                    self.ws_fwd[i].data_ptr(), st))
            else:
                _lib.check(lib.dfm_linear_bn_forward(
                    x.data_ptr(), k, lin.weight.data_ptr(), _lib.ptr(lin.bias), B, n, k, self.z[i].data_ptr(),
                    self.ws_fwd[i].data_ptr(), st))
            stats = (self.stats[i].data_ptr(), bn.running_mean.data_ptr() if track else None,
                     bn.running_var.data_ptr() if track else None,
                     bn.num_batches_tracked.data_ptr() if track else None, float(bn.momentum), float(bn.eps))
            if i == self.L - 1:
                # ---- last block + head in one launch: logits, d logits, the BatchNorm's statistics and mask
                # (its output a is never stored; loss + head gradients: next launch) ----
                ctx = self._bn_ctx(i)
                _lib.check(lib.dfm_head_bn_bce(
                    self.ws_fwd[i].data_ptr(), *stats, B, head.in_features, head.weight.data_ptr(),
                    _lib.ptr(head.bias), self.fo.data_ptr(), _lib.ptr(extra), self.labels.data_ptr(),
                    self.logits.data_ptr(), self.g_logits.data_ptr(), C.byref(ctx), st))
                break
            if x6:
                _lib.check(lib.dfm_bn_relu_dropout_apply_planes(
                    self.z[i].data_ptr(), B, n, self.ws_fwd[i].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                    *stats, self.drop_p[i], self.seed.data_ptr(), i, None, self.a_f[i].data_ptr(),
                    self.a_s[i].data_ptr(), st))
            else:
                _lib.check(lib.dfm_bn_relu_dropout_apply(
                    self.z[i].data_ptr(), B, n, self.ws_fwd[i].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                    *stats, self.drop_p[i], self.seed.data_ptr(), i, self.a[i].data_ptr(), st))
            x = self.a[i]
        tail = _lib.HeadTail()
        tail.g_w = head.weight.grad.data_ptr()
        tail.g_b = head.bias.grad.data_ptr() if head.bias is not None else None
        tail.loss = self.loss.data_ptr()
        tail.g_b2 = self._second_logit_bias_grad()
        # ---- backward: the interaction layer first (its d embeddings ride in layer 1's epilogue) ----
        fmb = self._interaction_backward()
        for i in range(self.L - 1, -1, -1):
            lin = self.lin[i]
            n, k = lin.out_features, lin.in_features
            if x6:
                _lib.check(lib.dfm_bn_backward_apply_planes(
                    C.byref(ctx), B, n, C.byref(tail) if i == self.L - 1 else None, None, self.dz_f[i].data_ptr(),
                    self.dz_s[i].data_ptr(), st))
                if i > 0:
                    ctx = self._bn_ctx(i - 1)
                _lib.check(lib.dfm_linear_backward_x6(
                    self.dz_f[i].data_ptr(), self.dz_s[i].data_ptr(), B, n, self.x0.data_ptr() if i == 0 else None,
                    self.a_s[i - 1].data_ptr() if i > 0 else None, k, self.w_s[i].data_ptr(),
                    self.g_x0.data_ptr() if i == 0 else None, C.byref(ctx) if i > 0 else None,
                    C.byref(fmb) if (i == 0 and fmb is not None) else None, self.ws_lin[i].data_ptr(), st))
                continue
            _lib.check(lib.dfm_bn_backward_apply(C.byref(ctx), B, n, C.byref(tail) if i == self.L - 1 else None,
                                                 self.dy[i].data_ptr(), st))
            xin = self.a[i - 1] if i > 0 else self.x0
            if i > 0:
                ctx = self._bn_ctx(i - 1)
                _lib.check(lib.dfm_linear_backward(
                    self.dy[i].data_ptr(), B, n, xin.data_ptr(), k, lin.weight.data_ptr(), None, C.byref(ctx),
                    None, 3, self.ws_lin[i].data_ptr(), st))
            else:
                _lib.check(lib.dfm_linear_backward(
                    self.dy[i].data_ptr(), B, n, xin.data_ptr(), k, lin.weight.data_ptr(), self.g_x0.data_ptr(),
                    None, C.byref(fmb) if fmb is not None else None, 3, self.ws_lin[i].data_ptr(), st))
        self._finish_embedding_grad()
        # the batch-split d weight products of all layers -> the flat gradient buffer, one launch
        extra = self._extra_slab_refs()
        n_refs = self.L + (1 if self._dense_parts else 0) + len(extra)
        refs = (_lib.SlabRef * n_refs)()
        for j, (ws_ptr, g_ptr, elems, splits) in enumerate(extra):
            r = refs[n_refs - len(extra) + j]
            r.workspace, r.g_w = ws_ptr, g_ptr
            r.batch, r.out_features, r.in_features, r.splits = 1, 1, elems, splits
        for i in range(self.L):
            r, lin = refs[i], self.lin[i]
            r.workspace, r.g_w = self.ws_lin[i].data_ptr(), lin.weight.grad.data_ptr()
            r.batch, r.out_features, r.in_features = B, lin.out_features, lin.in_features
            if x6:
                r.splits = lib.dfm_linear_backward_x6_splits(B, lin.out_features, lin.in_features)
        if self._dense_parts:            # the sliced DENSE-field gradients: the flat buffer's first n_l2 floats
            r = refs[self.L]
            r.workspace, r.g_w = self._dense_partial.data_ptr(), self.opt.flat_grad.data_ptr()
            r.batch, r.out_features, r.in_features, r.splits = 1, 1, self.opt.n_l2, self._dense_parts
        if self.slabs_travel:
            self._slab_refs = (refs, n_refs)         # summed by the gradient pack kernel (field-sharded tables)
        elif not self.opt.split:
            self.opt.slab_refs = (refs, n_refs)      # summed by the optimizer's prepare launch
        if not inline:
            cur.wait_stream(self.side)
        self._embedding_backward(self.g_logits, self.g_fe)
        if not self.slabs_travel and self.opt.split:
            # replicated tables, exchange outside the graph: every slab must be in the flat gradient first
            _lib.check(lib.dfm_linear_backward_finish(refs, n_refs, st))

    def _dense_slices(self):
        if not self._dense_parts:
            return None
        return self._dense_partial, self._dense_parts, self.opt.flat_grad

    def _second_logit_bias_grad(self):
        """data_ptr of another bias added to the logit (its gradient is sum(d logits), like the head's), or None."""
        return None

    def _extra_slab_refs(self):
        """(workspace ptr, gradient ptr, elements, slabs): further partial sums the step's slab reduction adds."""
        return []


class FusedDeepFMStep(_FusedTowerStep):
    head_name = "output_linear"

    @staticmethod
    def eligible(model) -> bool:
        from deepfm_amd.models.deepfm import DeepFM
        return isinstance(model, DeepFM) and _tower_ok(model)

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        super().__init__(model, optimizer, batch_size, use_graph)
        f32 = dict(dtype=torch.float32, device=optimizer.device)
        self.fm = torch.empty(batch_size, **f32)
        self.fm_sum = torch.empty(batch_size, self.fe.shape[2], **f32)

    def _gather_args(self) -> dict:
        return dict(fm_out=self.fm, fm_sum=self.fm_sum)      # FM value and S = sum_f e from the gather itself

    def _interaction_forward(self):
        return self.fm

    def _interaction_backward(self):
        fmb = _lib.FmBwd()
        fmb.g_fm, fmb.fm_sum, fmb.e = self.g_logits.data_ptr(), self.fm_sum.data_ptr(), self.x0.data_ptr()
        fmb.dim = self.fe.shape[2]
        return fmb


class FusedXDeepFMStep(_FusedTowerStep):
    """xDeepFM (xdeepfm.py:36-48): the CIN stack and its Linear head called directly on preallocated
    buffers (``dfm_cin_forward`` / ``dfm_cin_backward``, ``dfm_gemm_f32`` for the 1-wide head), the CIN's
    d field_embeddings added inside the first tower Linear's d-input epilogue."""

    head_name = "dnn_linear"

    @staticmethod
    def eligible(model) -> bool:
        from deepfm_amd.models.xdeepfm import xDeepFM
        return isinstance(model, xDeepFM) and _tower_ok(model)

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        super().__init__(model, optimizer, batch_size, use_graph)
        lib = _lib.load()
        dev, B = optimizer.device, batch_size
        f32 = dict(dtype=torch.float32, device=dev)
        cin = model.cin
        F, D = self.fe.shape[1], self.fe.shape[2]
        self.cin = cin
        self.cin_L = len(cin.layer_sizes)
        self.cin_sizes = (C.c_int32 * self.cin_L)(*cin.layer_sizes)
        self.cin_split = 1 if cin.split_half else 0
        self.cin_out = torch.empty(B, cin.output_dim, **f32)
        self.cin_lin = torch.empty(B, 1, **f32)
        self.g_cin_out = torch.empty(B, cin.output_dim, **f32)
        self.g_cin_fe = torch.empty(B, F, D, **f32)
        self.cin_saved = torch.empty(max(lib.dfm_cin_saved_bytes(self.cin_sizes, self.cin_L, self.cin_split, B, F, D) // 4, 1), **f32)
        self.cin_ws_f = torch.empty(max(lib.dfm_cin_forward_workspace_bytes(self.cin_sizes, self.cin_L, self.cin_split, F, D), 16),
                                    dtype=torch.uint8, device=dev)
        self.cin_ws_b = torch.empty(max(lib.dfm_cin_backward_workspace_bytes(self.cin_sizes, self.cin_L, self.cin_split, B, F, D) // 4, 1),
                                    **f32)
        self.ones = torch.ones(B, 1, **f32)
        head = model.cin_linear
        # cin_linear (256 -> 1): a row-dot / outer-product kernel pair instead of six generic GEMM launches
        self.head1 = bool(lib.dfm_linear1_supported(head.in_features)) and head.out_features == 1
        if self.head1:
            self.head1_splits = lib.dfm_linear1_backward_splits(B)
            self.head1_ws = torch.empty(self.head1_splits, head.in_features, **f32)
        for p in list(cin.parameters()) + list(model.cin_linear.parameters()):
            if p.grad is None or not p.grad.is_contiguous() or not p.is_contiguous():
                raise RuntimeError("fused steps need RowSparseAdam's flat gradient views on every dense parameter")

    def _ptrs(self, tensors):
        arr = (C.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr()
        return arr

    def _interaction_forward(self):
        from deepfm_amd.models.layers.dnn import _gemm
        lib, B = _lib.load(), self.B
        F, D = self.fe.shape[1], self.fe.shape[2]
        ws = [c.weight for c in self.cin.conv_layers]
        bs = [c.bias for c in self.cin.conv_layers]
        _lib.check(lib.dfm_cin_forward(self.fe.data_ptr(), B, F, D, self._ptrs(ws), self._ptrs(bs), self.cin_sizes,
                                       self.cin_L, self.cin_split, self.cin_out.data_ptr(), self.cin_saved.data_ptr(),
                                       self.cin_ws_f.data_ptr(), _lib.stream_handle()))
        head = self.model.cin_linear                       # explicit = cin_linear(cin(fe))   (xdeepfm.py:41-42)
        K = head.in_features
        if self.head1:
            _lib.check(lib.dfm_linear1_forward(self.cin_out.data_ptr(), B, K, head.weight.data_ptr(), _lib.ptr(head.bias),
                                               self.cin_lin.data_ptr(), _lib.stream_handle()))
        else:
            _gemm(self.cin_out, K, True, head.weight, K, True, self.cin_lin, B, 1, K, bias=head.bias)
        return self.cin_lin

    def _second_logit_bias_grad(self):
        head = self.model.cin_linear
        return head.bias.grad.data_ptr() if self.head1 and head.bias is not None else None

    def _extra_slab_refs(self):
        if not self.head1:
            return []
        head = self.model.cin_linear
        return [(self.head1_ws.data_ptr(), head.weight.grad.data_ptr(), head.in_features, self.head1_splits)]

    def _interaction_backward(self):
        from deepfm_amd.models.layers.dnn import _gemm
        lib, B = _lib.load(), self.B
        F, D = self.fe.shape[1], self.fe.shape[2]
        head = self.model.cin_linear
        K = head.in_features
        g = self.g_logits                                                     # (B, 1) = d loss / d logit
        if self.head1:     # d cin_out = g w; dW as slabs for the step's slab reduction; d bias: the head's tail
            _lib.check(lib.dfm_linear1_backward(g.data_ptr(), self.cin_out.data_ptr(), B, K, head.weight.data_ptr(),
                                                self.g_cin_out.data_ptr(), self.head1_ws.data_ptr(),
                                                _lib.stream_handle()))
        else:
            _gemm(g, 1, True, head.weight, K, False, self.g_cin_out, B, K, 1)    # d cin_out = g w
            _gemm(g, 1, False, self.cin_out, K, False, head.weight.grad, 1, K, B, accumulate=True)   # dW += g^T cin_out
            if head.bias is not None:
                _gemm(g, 1, False, self.ones, 1, False, head.bias.grad.view(1, 1), 1, 1, B, accumulate=True)
        ws = [c.weight for c in self.cin.conv_layers]
        g_w = [c.weight.grad for c in self.cin.conv_layers]
        g_b = [c.bias.grad for c in self.cin.conv_layers]
        _lib.check(lib.dfm_cin_backward(self.fe.data_ptr(), B, F, D, self._ptrs(ws), self.cin_sizes, self.cin_L,
                                        self.cin_split, self.cin_saved.data_ptr(), self.g_cin_out.data_ptr(),
                                        self.g_cin_fe.data_ptr(), self._ptrs(g_w), self._ptrs(g_b),
                                        self.cin_ws_b.data_ptr(), _lib.stream_handle()))
        fmb = _lib.FmBwd()
        fmb.addend = self.g_cin_fe.data_ptr()
        return fmb


class _Ctx:
    """Stand-in for autograd's ctx: lets a torch.autograd.Function's forward / backward bodies be called
    directly (no graph recording, no AccumulateGrad launches)."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class FusedAttentionDeepFMStep(_FusedTowerStep):
    """AttentionDeepFM (attention_deepfm.py:48-66): logits = (fo + fm) + output_linear(dnn(cat[attention(fe),
    flat])).  The attention blocks run through the same kernels as the module (``_AttnGemmFn`` called
    directly), their parameter gradients land in the flat buffer with one multi-tensor add, and the three
    gradients of the embeddings (flat half of the DNN's d input, attention, FM) are summed in one pass
    (``dfm_embedding_grad_combine``)."""

    head_name = "output_linear"

    @staticmethod
    def eligible(model) -> bool:
        from deepfm_amd.models.attention_deepfm import AttentionDeepFM
        if not (isinstance(model, AttentionDeepFM) and _tower_ok(model)):
            return False
        att = model.attention
        F = model.schema.num_fields
        ok = _lib.load().dfm_attention_core_supported(F, att.attention_dim, att.num_heads)
        return bool(ok) and att.embed_dim % 4 == 0 and att.attention_dim % 4 == 0 and att.embed_dim <= 64 \
            and all(b.gemm_path for b in att.layers)

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        super().__init__(model, optimizer, batch_size, use_graph)
        f32 = dict(dtype=torch.float32, device=optimizer.device)
        B, D = batch_size, self.fe.shape[2]
        self.fm = torch.empty(B, **f32)
        self.fm_sum = torch.empty(B, D, **f32)
        self.blocks = list(model.attention.layers)
        self._ctxs: List[_Ctx] = []
        self._att_params = [p for b in self.blocks for p in b._param_list()]
        for p in self._att_params:
            if p.grad is None:
                raise RuntimeError("fused steps need RowSparseAdam's flat gradient views on every dense parameter")

    def _tower_input(self):
        F, D = self.fe.shape[1], self.fe.shape[2]
        self.xcat = torch.empty(self.B, 2 * F * D, dtype=torch.float32, device=self.fe.device)
        self.g_xcat = torch.empty_like(self.xcat)
        self.g_att = torch.empty(self.B, F * D, dtype=torch.float32, device=self.fe.device)
        return self.xcat

    def _tower_input_grad(self):
        return self.g_xcat

    def _gather_args(self) -> dict:
        return dict(fm_out=self.fm, fm_sum=self.fm_sum)

    def _interaction_forward(self):
        from deepfm_amd.models.layers.attention import _AttnGemmFn
        x = self.fe
        self._ctxs = []
        FD = self.fe.shape[1] * self.fe.shape[2]
        lib, st = _lib.load(), _lib.stream_handle()
        # dnn_in = cat([attention(fe).flatten(1), flat], dim=1)   (attention_deepfm.py:57-61): the last
        # block's residual LayerNorm writes its rows straight into the first half of xcat
        last = self.blocks[-1]
        for block in self.blocks:
            ctx = _Ctx()
            ctx.direct = True          # parameter gradients straight into the flat buffer's .grad views
            if block is last and block.use_residual:
                ctx.out_into = (self.xcat, 2 * FD)
            if block is self.blocks[0]:    # its input IS fe: the whole-block kernel writes the flat half of xcat too
                ctx.x_copy_into = (self.xcat.data_ptr() + FD * 4, 2 * FD)
            x = _AttnGemmFn.forward(ctx, block, x, *block._param_list())
            self._ctxs.append(ctx)
        if not last.use_residual:
            _lib.check(lib.dfm_copy_2d(x.data_ptr(), FD, self.xcat.data_ptr(), 2 * FD, self.B, FD, st))
        if not getattr(self._ctxs[0], "x_copied", False):
            _lib.check(lib.dfm_copy_2d(self.fe.data_ptr(), FD, self.xcat.data_ptr() + FD * 4, 2 * FD, self.B, FD, st))
        return self.fm

    def _interaction_backward(self):
        return None            # the first Linear stores its d input (B, 2 F D) as it is

    def _finish_embedding_grad(self) -> None:
        from deepfm_amd.models.layers.attention import _AttnGemmFn
        B, F, D = self.fe.shape
        FD = F * D
        # d attention(fe) = the first half of d dnn_in: read in place by the last block's LayerNorm backward
        if self.blocks[-1].use_residual:
            g = self.g_xcat
            self._ctxs[-1].g_from = 2 * FD
        else:
            g = self.g_att.view(B, F, D)
            _lib.check(_lib.load().dfm_copy_2d(self.g_xcat.data_ptr(), 2 * FD, g.data_ptr(), FD, B, FD, _lib.stream_handle()))
        # the first block's d x is d fe once the DNN's flat half and the FM backward are added: its whole-block
        # kernel does that in its one store (else: dfm_embedding_grad_combine below)
        self._ctxs[0].grad_tail = dict(out=self.g_fe, g_flat=self.g_xcat.data_ptr() + FD * 4, ld_flat=2 * FD,
                                       g_fm=self.g_logits.data_ptr(), fm_sum=self.fm_sum.data_ptr())
        for block, ctx in zip(reversed(self.blocks), reversed(self._ctxs)):
            out = _AttnGemmFn.backward(ctx, g)
            g = out[1]
            if len(out) > 2:           # the flat buffer is not laid out for direct writes: add the temporaries
                ps = block._param_list()
                torch._foreach_add_([p.grad for p in ps], [t.view_as(p) for t, p in zip(out[2:], ps)])
        if getattr(self._ctxs[0], "tail_done", False):
            return
        _lib.check(_lib.load().dfm_embedding_grad_combine(
            self.g_xcat.data_ptr() + FD * 4, 2 * FD, g.data_ptr(), self.g_logits.data_ptr(), self.fm_sum.data_ptr(),
            self.fe.data_ptr(), B, F, D, self.g_fe.data_ptr(), _lib.stream_handle()))


def fused_step_class(model):
    """The fused step that takes ``model``, or None (-> RowSparseTrainStep over autograd)."""
    for cls in (FusedDeepFMStep, FusedXDeepFMStep, FusedAttentionDeepFMStep):
        if cls.eligible(model):
            return cls
    return None
