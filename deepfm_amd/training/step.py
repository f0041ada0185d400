"""One training step of a BaseCTRModel on the HIP path, optionally as a HIP graph.

Step = the body of the reference's ``Trainer._train_epoch`` (trainer.py:212-240):
forward -> BCEWithLogits (+ L2 term) -> backward -> clip -> Adam, with the embedding
tables in row-sparse mode (``RowSparseAdam``).  Layout of one step on the stream:

    [eager]  fused embedding gather (``dfm_embedding_forward``)  <- optionally bracketed by
             HIP events so bench.py can time exactly this kernel in the timed region
    [graph A] row plan (side stream), interaction layers + DNN forward, loss, backward, row gradients
    [eager]  data-parallel exchange (RCCL all-reduce / all-gather), world_size > 1 only
    [graph B] merge + clip + row-wise Adam + dense Adam

With one rank A and B are a single graph.  No host synchronisation inside a step.
"""

from __future__ import annotations

import os
from typing import List, Optional

import torch

from deepfm_amd.data.schema import FeatureType
from deepfm_amd.training.losses import bce_with_logits_mean
from deepfm_amd.training.rowsparse import RowSparseAdam


class RowSparseTrainStep:
    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        self.model, self.opt, self.B = model, optimizer, batch_size
        self.emb = model.embedding
        if self.emb.grad_mode != "rowsparse":
            raise ValueError("RowSparseTrainStep needs model.embedding in 'rowsparse' grad mode")
        dev = optimizer.device
        specs = list(model.schema.fields.values())
        self.n_sparse = sum(s.feature_type is FeatureType.SPARSE for s in specs)
        self.n_dense = sum(s.feature_type is FeatureType.DENSE for s in specs)
        # static, packed inputs in ONE buffer [ids (S,B) int64 | dense (Dn,B) f32 | labels (B) f32]: a
        # whole batch is loaded with a single device-to-device copy; the per-field (B,) views keep
        # the reference's dict contract
        ns, nd = max(self.n_sparse, 1), max(self.n_dense, 1)
        self.packed_bytes = ns * batch_size * 8 + nd * batch_size * 4 + batch_size * 4
        self.packed = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=dev)
        o1 = ns * batch_size * 8
        o2 = o1 + nd * batch_size * 4
        self.ids = self.packed[:o1].view(torch.int64).view(ns, batch_size)
        self.dense = self.packed[o1:o2].view(torch.float32).view(nd, batch_size)
        self.labels = self.packed[o2:].view(torch.float32)
        self.inputs: List[torch.Tensor] = []
        self._rec_offsets: List[int] = []          # byte offset of every field's input inside a batch record
        si = di = 0
        for s in specs:
            if s.feature_type is FeatureType.SPARSE:
                self.inputs.append(self.ids[si]); self._rec_offsets.append(si * batch_size * 8); si += 1
            else:
                self.inputs.append(self.dense[di]); self._rec_offsets.append(o1 + di * batch_size * 4); di += 1
        self._rec_labels = o2
        self._record: Optional[torch.Tensor] = None   # batch record the next gather reads (run_from)
        F, D = len(specs), self.emb.fm_embed_dim
        self.fo = torch.empty(batch_size, 1, dtype=torch.float32, device=dev)
        self.fe = torch.empty(batch_size, F, D, dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self.side = torch.cuda.Stream(device=dev)   # row plan (needs only the ids) overlaps fwd/bwd
        self.graph_a: Optional[torch.cuda.CUDAGraph] = None
        self.graph_b: Optional[torch.cuda.CUDAGraph] = None
        self.dense_grads = {id(p): p.grad for p in self.emb.non_table_parameters() if p.grad is not None}
        self.gather_events = None          # list of (start, end) torch.cuda.Event pairs when timing
        self.emb.pin_plan(dev)             # the step holds raw parameter pointers from here on
        for m in model.modules():          # DNN / head backward: accumulate straight into the flat .grad views
            if hasattr(m, "direct_grads"):
                m.direct_grads = True

    # ------------------------------------------------------------------ pieces
    def load_batch(self, ids: torch.Tensor, dense: torch.Tensor, labels: torch.Tensor) -> None:
        """ids (S,B) int64, dense (Dn,B) float32, labels (B,) — device-to-device copies."""
        if self.n_sparse:
            self.ids.copy_(ids, non_blocking=True)
        if self.n_dense:
            self.dense.copy_(dense, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)

    def pack_batches(self, ids: torch.Tensor, dense: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """(n, S, B) int64, (n, Dn, B) f32, (n, B) f32 -> (n, packed_bytes) uint8 records for
        ``load_packed`` (done once, outside any timed region)."""
        n = labels.shape[0]
        parts = []
        if self.n_sparse:
            parts.append(ids.contiguous().view(n, -1).view(torch.uint8))
        else:
            parts.append(torch.zeros(n, self.B * 8, dtype=torch.uint8, device=labels.device))
        if self.n_dense:
            parts.append(dense.contiguous().view(n, -1).view(torch.uint8))
        else:
            parts.append(torch.zeros(n, self.B * 4, dtype=torch.uint8, device=labels.device))
        parts.append(labels.contiguous().view(n, -1).view(torch.uint8))
        return torch.cat(parts, dim=1).contiguous()

    def load_packed(self, record: torch.Tensor) -> None:
        """One device-to-device copy of a pack_batches() record into the static inputs."""
        self.packed.copy_(record, non_blocking=True)

    def _gather_args(self) -> dict:
        """Extra outputs of the gather (subclasses: the fused step adds the FM value and S)."""
        return {}

    def _gather(self) -> None:
        rec = self._record
        if rec is None:
            self.emb.forward_into(self.inputs, self.B, self.fo, self.fe, **self._gather_args())
            return
        base = rec.data_ptr()
        self.emb.forward_staged([base + o for o in self._rec_offsets], self.inputs, self.B, self.fo, self.fe,
                                extra_src_ptr=base + self._rec_labels, extra_dst=self.labels, **self._gather_args())
        self._record = None

    def _body_a(self) -> None:
        self.opt.zero_grad()
        cur = torch.cuda.current_stream()
        side = os.environ.get("DFM_ROWPLAN_SIDE_STREAM") == "1"    # see fused_step.py: in line is faster
        if side:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.emb.build_rowplan(self.inputs, self.B)
        else:
            self.emb.build_rowplan(self.inputs, self.B)
        fo = self.fo.detach().requires_grad_()
        fe = self.fe.detach().requires_grad_()
        logits = self.model._forward_components(fo, fe, fe.view(self.B, -1))
        # plain BCE: the L2 term (base.py:78-83) is applied as g += 2*l2*p by the optimizer
        loss = bce_with_logits_mean(logits.view(-1), self.labels)
        loss.backward()
        self.loss.copy_(loss.detach())
        if side:
            cur.wait_stream(self.side)
        self.emb.backward_rowsparse(self.inputs, fo.grad, fe.grad, self.dense_grads)

    def _body_b(self) -> None:
        self.opt.apply()

    # ------------------------------------------------------------------ capture / run
    def _mutable_state(self) -> List[torch.Tensor]:
        """Everything a training step writes besides the table rows of the ids it is given: the flat dense
        parameter / moment / gradient buffers, step count, norm scalars, dropout seed, every module buffer
        (BatchNorm running statistics) and the static inputs."""
        opt = self.opt
        ts = [opt.flat_param, opt.flat_m, opt.flat_v, opt.flat_grad, opt.step_count, opt.sq_norm, opt.clip_coef,
              self.loss, self.packed]
        if opt.seed_tick is not None:
            ts.append(opt.seed_tick)
        seed = getattr(getattr(self.model, "dnn", None), "_seed", None)
        if seed is not None:
            ts.append(seed)
        ts += list(self.model.buffers())
        return ts

    def capture(self, warmup_iters: int = 1) -> None:
        """Capture graph A (and B) after ``warmup_iters`` eager steps.  Side-effect free: the warm-up
        steps run on an all-padding batch (id 0 everywhere: no table row receives a gradient, so the
        row-wise Adam touches nothing) and every other piece of state a step writes — dense parameters,
        Adam moments, step count, dropout seed, BatchNorm running statistics, the static inputs — is
        restored afterwards, bit for bit."""
        if not self.use_graph:
            return
        state = self._mutable_state()
        saved = [t.clone() for t in state]
        record, self._record = self._record, None
        if self.n_sparse:
            self.ids.zero_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._gather()
                self._body_a()
                self.opt.exchange()
                self._body_b()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        single = not self.opt.split
        # opt-in (DFM_DP_GRAPH_COLLECTIVE=1): capture the exchange inside the graph — one graph launch
        # per step under data parallelism.  Verified with a single-rank RCCL communicator only
        # (DFM_FORCE_DP_PATH=1); not validated on a multi-GPU box, hence not the default.
        fused_exchange = ((not single) and os.environ.get("DFM_DP_GRAPH_COLLECTIVE") == "1"
                          and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl")
        # thread_local capture mode: another thread (the RCCL watchdog polling its events under
        # data parallelism) must not invalidate the capture
        mode = dict(capture_error_mode="thread_local")
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, **mode):
            if os.environ.get("DFM_EXP_GATHER_IN_GRAPH") == "1":      # timing experiment only
                self._gather()
            self._body_a()
            if single or fused_exchange:
                self.opt.exchange()          # one rank: no device work, selects the local row lists
                self._body_b()
        if not single and not fused_exchange:
            self.opt.exchange()
            torch.cuda.synchronize()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, **mode):
                self._body_b()
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, v in zip(state, saved):
                t.copy_(v)
        self._record = record
        torch.cuda.synchronize()

    def run_from(self, record: torch.Tensor) -> None:
        """One step on a ``pack_batches()`` record without the separate load: the (eager) gather reads
        its inputs from the record and refreshes the static input buffers — ids, dense values,
        labels — that the captured part of the step reads."""
        if record.numel() != self.packed_bytes or record.dtype != torch.uint8 or not record.is_contiguous():
            raise ValueError("run_from expects one contiguous pack_batches() record")
        if record.data_ptr() % 16:
            raise ValueError("batch records must be 16-byte aligned")
        self._record = record
        self.run()

    def run(self, time_gather: bool = False) -> None:
        if time_gather:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            self._gather()
            end.record()
            if self.gather_events is None:
                self.gather_events = []
            self.gather_events.append((start, end))
        elif not (os.environ.get("DFM_EXP_GATHER_IN_GRAPH") == "1" and self.graph_a is not None):
            self._gather()
        if self.graph_a is not None:
            self.graph_a.replay()
            if self.graph_b is not None:
                # the replay produced the row gradients; the Python-side flag was only set while
                # the graph was being captured
                self.emb.rowsparse.has_grad = True
                self.opt.exchange()
                self.graph_b.replay()
        else:
            self._body_a()
            self.opt.exchange()
            self._body_b()
