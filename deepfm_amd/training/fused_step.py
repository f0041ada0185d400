"""DeepFM training step on the fused tower kernels (csrc/tower.hip): no autograd, ~25 launches.

Same step as ``RowSparseTrainStep`` (reference ``Trainer._train_epoch`` body, trainer.py:212-240,
with the model of deepfm.py:30-42), but the DNN tower, the head and the FM backward run on
``dfm_linear_bn_forward`` / ``dfm_bn_relu_dropout_apply`` / ``dfm_head_bce`` /
``dfm_bn_backward_apply`` / ``dfm_linear_backward`` with hand-written backward wiring:

    gather (+ fm value, + S = sum_f e)                                        1 launch (eager, timed)
    per layer: GEMM + batch statistics, BN/ReLU/Dropout apply                 2 launches
    head: logits + BCE + d logits + head grads + last BN's mask               1 launch
    per layer (top down): BN backward apply, [dW | dx + lower BN mask / FM]   2 launches
    embedding backward (dense fields + row gradients), optimizer              unchanged

On an MI355X every dependent launch costs ~4.5 us, and the autograd path needs ~65 of them.
Eligible: DeepFM with the reference-default tower (BatchNorm + ReLU), hidden sizes multiples of 4
(last one a multiple of 32, <= 256), uniform embedding schema in ``rowsparse`` mode, training mode.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import List

import torch

from deepfm_amd import _lib
from deepfm_amd.training.rowsparse import RowSparseAdam
from deepfm_amd.training.step import RowSparseTrainStep


def _zeros_bytes(nbytes: int, device) -> torch.Tensor:
    return torch.zeros(max((nbytes + 3) // 4, 1), dtype=torch.int32, device=device)


class FusedDeepFMStep(RowSparseTrainStep):
    @staticmethod
    def eligible(model) -> bool:
        from deepfm_amd.models.deepfm import DeepFM
        if not isinstance(model, DeepFM) or not model.training:
            return False
        dnn = model.dnn
        if not getattr(dnn, "_fusable", False):
            return False
        widths = [dnn.mlp[4 * i].out_features for i in range(dnn._n_layers)]
        if any(w % 4 for w in widths) or widths[-1] % 32 or widths[-1] > 256:
            return False
        bn = dnn.mlp[1]
        if bn.momentum is None or not bn.affine:
            return False
        return dnn.mlp[0].in_features % 4 == 0 and model.embedding.grad_mode == "rowsparse"

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        super().__init__(model, optimizer, batch_size, use_graph)
        if not self.eligible(model):
            raise ValueError("FusedDeepFMStep: model/configuration not eligible (use RowSparseTrainStep)")
        lib = _lib.load()
        dev, B = optimizer.device, batch_size
        dnn = model.dnn
        self.L = dnn._n_layers
        f32 = dict(dtype=torch.float32, device=dev)
        F, D = self.fe.shape[1], self.fe.shape[2]
        self.fm = torch.empty(B, **f32)
        self.fm_sum = torch.empty(B, D, **f32)
        self.x0 = self.fe.view(B, F * D)
        self.g_fe = torch.empty(B, F, D, **f32)
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
        self.seed = torch.randint(1, 2 ** 40, (1,), dtype=torch.int64, device=dev)
        optimizer.seed_tick = self.seed          # advanced by the optimizer's norm-finalize kernel
        self.rowplan_side_stream = os.environ.get("DFM_ROWPLAN_SIDE_STREAM") == "1"
        for p in list(dnn.parameters()) + list(model.output_linear.parameters()):
            if p.grad is None or not p.grad.is_contiguous():
                raise RuntimeError("FusedDeepFMStep needs RowSparseAdam's flat gradient views on every dense parameter")

    # ------------------------------------------------------------------ pieces
    def _gather_args(self) -> dict:
        return dict(fm_out=self.fm, fm_sum=self.fm_sum)

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
        # (measured 0.253 ms in line vs 0.255 overlapped when the sort still took 38 us).
        inline = not self.rowplan_side_stream
        if inline:
            self.emb.build_rowplan(self.inputs, B)
        else:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.emb.build_rowplan(self.inputs, B)
        # ---- forward ----
        x = self.x0
        for i in range(self.L):
            lin, bn = self.lin[i], self.bn[i]
            n, k = lin.out_features, lin.in_features
            track = bn.track_running_stats and bn.running_mean is not None
            _lib.check(lib.dfm_linear_bn_forward(
                x.data_ptr(), k, lin.weight.data_ptr(), _lib.ptr(lin.bias), B, n, k, self.z[i].data_ptr(),
                self.ws_fwd[i].data_ptr(), st))
            _lib.check(lib.dfm_bn_relu_dropout_apply(
                self.z[i].data_ptr(), B, n, self.ws_fwd[i].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                self.stats[i].data_ptr(), bn.running_mean.data_ptr() if track else None,
                bn.running_var.data_ptr() if track else None, bn.num_batches_tracked.data_ptr() if track else None,
                float(bn.momentum), float(bn.eps), self.drop_p[i], self.seed.data_ptr(), i, self.a[i].data_ptr(), st))
            x = self.a[i]
        # ---- head: logits, d logits, mask of the last BatchNorm (loss + head gradients: next launch) ----
        head = self.model.output_linear
        ctx = self._bn_ctx(self.L - 1)
        _lib.check(lib.dfm_head_bce(
            x.data_ptr(), B, head.in_features, head.weight.data_ptr(), _lib.ptr(head.bias), self.fo.data_ptr(),
            self.fm.data_ptr(), self.labels.data_ptr(), self.logits.data_ptr(), self.g_logits.data_ptr(),
            C.byref(ctx), st))
        tail = _lib.HeadTail()
        tail.g_w = head.weight.grad.data_ptr()
        tail.g_b = head.bias.grad.data_ptr() if head.bias is not None else None
        tail.loss = self.loss.data_ptr()
        # ---- backward, top layer first ----
        for i in range(self.L - 1, -1, -1):
            lin = self.lin[i]
            n, k = lin.out_features, lin.in_features
            _lib.check(lib.dfm_bn_backward_apply(C.byref(ctx), B, n, C.byref(tail) if i == self.L - 1 else None,
                                                 self.dy[i].data_ptr(), st))
            xin = self.a[i - 1] if i > 0 else self.x0
            if i > 0:
                ctx = self._bn_ctx(i - 1)
                _lib.check(lib.dfm_linear_backward(
                    self.dy[i].data_ptr(), B, n, xin.data_ptr(), k, lin.weight.data_ptr(), None, C.byref(ctx),
                    None, 3, self.ws_lin[i].data_ptr(), st))
            else:
                fmb = _lib.FmBwd()
                fmb.g_fm, fmb.fm_sum, fmb.e = self.g_logits.data_ptr(), self.fm_sum.data_ptr(), self.x0.data_ptr()
                fmb.dim = self.fe.shape[2]
                _lib.check(lib.dfm_linear_backward(
                    self.dy[i].data_ptr(), B, n, xin.data_ptr(), k, lin.weight.data_ptr(), self.g_fe.data_ptr(),
                    None, C.byref(fmb), 3, self.ws_lin[i].data_ptr(), st))
        # the batch-split d weight products of all layers -> the flat gradient buffer, one launch
        refs = (_lib.SlabRef * self.L)()
        for i in range(self.L):
            r, lin = refs[i], self.lin[i]
            r.workspace, r.g_w = self.ws_lin[i].data_ptr(), lin.weight.grad.data_ptr()
            r.batch, r.out_features, r.in_features = B, lin.out_features, lin.in_features
        if not self.opt.split:
            self.opt.slab_refs = (refs, self.L)      # summed by the optimizer's prepare launch
        else:
            _lib.check(lib.dfm_linear_backward_finish(refs, self.L, st))    # must precede the all-reduce
        if not inline:
            cur.wait_stream(self.side)
        self.emb.backward_rowsparse(self.inputs, self.g_logits, self.g_fe, self.dense_grads)
