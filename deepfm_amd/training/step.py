"""One training step of a BaseCTRModel on the HIP path, as ONE HIP graph per step.

Step = the body of the reference's ``Trainer._train_epoch`` (trainer.py:212-240):
forward -> BCEWithLogits (+ L2 term) -> backward -> clip -> Adam, with the embedding
tables in row-sparse mode (``RowSparseAdam``).  Layout of one step on the stream:

    [graph A] fused embedding gather, reading the batch from its packed record and refreshing the
              step's static inputs on the way (``dfm_embedding_forward_staged``); row plan,
              interaction layers + DNN forward, loss, backward, row gradients;
              one rank: merge + clip + row-wise Adam + dense Adam as well
    [eager]   data-parallel exchange (RCCL all-gather), world_size > 1 only
    [graph B] merge + clip + row-wise Adam + dense Adam, world_size > 1 only

The batch changes every step and a graph's kernel arguments are frozen at capture, so the gather's
kernel NODE is re-pointed at the next batch record from the host before every launch
(``dfm_embedding_forward_staged_update`` -> hipGraphExecKernelNodeSetParams; nothing is enqueued).  An
exec must not be updated while a launch of it may still be pending, so graph A is instantiated TWICE
and the two execs alternate: while one runs, the other — whose previous launch finished a whole step
ago — is updated and queued.  Measured on one MI355X: 0.216 ms/step against 0.225 with the gather
launched eagerly in front of the graph (two eager <-> graph transitions of ~10 us each disappear).
No host synchronisation inside a step beyond waiting for the launch before the previous one.

Timing the gather kernel: event-record NODES around it inside the graph cost ~5.5 us of device timeline
each and read 3 us more than the kernel runs (measured), so they are not used.  ``capture(timed_variant=
True)`` additionally captures the step WITHOUT its gather; ``run(eager_gather=True)`` then launches the
gather eagerly in front of that copy, where ``dfm_gather_timing_begin`` can attach start/stop events to
the dispatch itself.  bench.py does this on every 8th step of its timed region.
"""

from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref
from typing import List, Optional

import torch

from deepfm_amd import _lib
from deepfm_amd.data.schema import FeatureType
from deepfm_amd.training.losses import bce_with_logits_mean
from deepfm_amd.training.rowsparse import RowSparseAdam


# Arithmetic of the tower's backward GEMMs that bench.py / the tools select by default (dfm_tower_set_mode): the library
# itself starts in mode 0 and only changes on an explicit call.
TOWER_MODE_DEFAULT = 0

# Every live step, so that one call drops all captured graphs: graphs that hold captured RCCL kernels must be
# destroyed BEFORE the communicator (``destroy_process_group()`` under live graphs does not return — found on
# the one-GPU RCCL run).  Also registered with atexit, for processes that end on an exception.
_LIVE_STEPS: "weakref.WeakSet" = weakref.WeakSet()


def release_all_graphs() -> None:
    for step in list(_LIVE_STEPS):
        try:
            step.release_graphs()
        except Exception:      # teardown: best effort
            pass


atexit.register(release_all_graphs)


class _GraphSlot:
    """One instantiated copy of graph A: exec + the gather kernel node of every step it holds."""

    def __init__(self) -> None:
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.nodes: List[C.c_void_p] = []
        self.done: Optional[torch.cuda.Event] = None      # recorded after the slot's latest launch


class RowSparseTrainStep:
    exchange_in_body = False      # True: the step's collectives are part of _gather / _body_a / _body_b themselves
    rowplan_first_default = True  # row plan + row touch in front of the gather (False: in line behind it, round 2's order)
    plan_lookahead_default = True # steps 2.. of a multi-step graph: the plan is built by the previous step's apply launch

    def __init__(self, model, optimizer: RowSparseAdam, batch_size: int, use_graph: bool = True) -> None:
        self.model, self.opt, self.B = model, optimizer, batch_size
        self.emb = model.embedding
        if self.emb.grad_mode != "rowsparse":
            raise ValueError("RowSparseTrainStep needs model.embedding in 'rowsparse' grad mode")
        dev = optimizer.device
        specs = list(model.schema.fields.values())
        self.n_sparse = sum(s.feature_type is FeatureType.SPARSE for s in specs)
        self.n_dense = sum(s.feature_type is FeatureType.DENSE for s in specs)
        # packed record layout [ids (S,B) int64 | dense (Dn,B) f32 | labels (B) f32]; three buffers of it:
        #   packed : the step's STATIC inputs (read by the row plan, the embedding backward, the loss) —
        #            written by the gather itself from the record it reads
        #   inbox  : where load_batch / load_packed put a batch that is not already a device record
        #   pad    : all-padding batch (id 0 everywhere) used by capture()'s warm-up
        ns, nd = max(self.n_sparse, 1), max(self.n_dense, 1)
        self.packed_bytes = ns * batch_size * 8 + nd * batch_size * 4 + batch_size * 4
        self.packed = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=dev)
        self.inbox = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=dev)
        self.pad = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=dev)
        o1 = ns * batch_size * 8
        o2 = o1 + nd * batch_size * 4

        def views(buf):
            return (buf[:o1].view(torch.int64).view(ns, batch_size), buf[o1:o2].view(torch.float32).view(nd, batch_size),
                    buf[o2:].view(torch.float32))
        self.ids, self.dense, self.labels = views(self.packed)
        self.in_ids, self.in_dense, self.in_labels = views(self.inbox)
        self.inputs: List[torch.Tensor] = []
        self._rec_offsets: List[int] = []          # byte offset of every field's input inside a batch record
        si = di = 0
        for s in specs:
            if s.feature_type is FeatureType.SPARSE:
                self.inputs.append(self.ids[si]); self._rec_offsets.append(si * batch_size * 8); si += 1
            else:
                self.inputs.append(self.dense[di]); self._rec_offsets.append(o1 + di * batch_size * 4); di += 1
        self._rec_labels = o2
        self._rec_id_offsets = [k * batch_size * 8 for k in range(self.n_sparse)]   # SPARSE id columns of a record
        # Row plan IN FRONT of the gather, on the batch record itself, with row-touch workgroups (csrc/rowplan.hip):
        # the sort keeps 26 CUs busy for ~13 us either way; here the other CUs use that time to pull the batch's ids
        # and table rows on-die, and the gather that follows no longer pays cold ids + three HBM round trips behind
        # the optimizer's dirty lines.  Same plan, same results; one more graph node re-pointed per step.
        self.rowplan_first = self.n_sparse > 0 and self.rowplan_first_default
        self._plan_done = False
        # Inside a graph of several steps, step k + 1's row plan (+ row touch) rides in step k's last optimizer launch
        # (dfm_step_apply_plan) instead of opening step k + 1: two sets of plan buffers alternate, and a third node per
        # step is re-pointed at launch.  Only the first step of a graph builds its plan in front of its gather.
        self.plan_lookahead = self.rowplan_first and self.plan_lookahead_default
        self._plan_sets = None
        self.cont_slots: List[_GraphSlot] = []
        self._turn_cont = 0
        self._prepared = None                        # graph slot whose nodes prepare_group() has pointed at its records
        self._handoff_ptr: Optional[int] = None      # record (data_ptr) whose plan the previous launch left in the hand-off set
        self._record: torch.Tensor = self.inbox       # batch record the next gather reads
        F, D = len(specs), self.emb.fm_embed_dim
        self.fo = torch.empty(batch_size, 1, dtype=torch.float32, device=dev)
        self.fe = torch.empty(batch_size, F, D, dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self.side = torch.cuda.Stream(device=dev)   # row plan on a side stream (DFM_ROWPLAN_SIDE_STREAM=1 only)
        self.slots: List[_GraphSlot] = []
        self.steps_per_graph = 1
        self._turn = 0
        self.graph_b: Optional[torch.cuda.CUDAGraph] = None
        self.body_graph: Optional[torch.cuda.CUDAGraph] = None     # graph A without the gather (timed variant)
        self.dense_grads = {id(p): p.grad for p in self.emb.non_table_parameters() if p.grad is not None}
        self.emb.pin_plan(dev)             # the step holds raw parameter pointers from here on
        _LIVE_STEPS.add(self)
        for m in model.modules():          # DNN / head backward: accumulate straight into the flat .grad views
            if hasattr(m, "direct_grads"):
                m.direct_grads = True

    # kept for callers that ask whether a graph exists (tools, tests)
    @property
    def graph_a(self) -> Optional[torch.cuda.CUDAGraph]:
        return self.slots[0].graph if self.slots else None

    # ------------------------------------------------------------------ inputs
    def load_batch(self, ids: torch.Tensor, dense: torch.Tensor, labels: torch.Tensor) -> None:
        """ids (S,B) int64, dense (Dn,B) float32, labels (B,) — device-to-device copies into the inbox
        record; the next ``run()`` trains on it."""
        if self.n_sparse:
            self.in_ids.copy_(ids, non_blocking=True)
        if self.n_dense:
            self.in_dense.copy_(dense, non_blocking=True)
        self.in_labels.copy_(labels, non_blocking=True)
        self._record = self.inbox

    def pack_batches(self, ids: torch.Tensor, dense: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """(n, S, B) int64, (n, Dn, B) f32, (n, B) f32 -> (n, packed_bytes) uint8 records for
        ``run_from`` (done once, outside any timed region)."""
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
        """One device-to-device copy of a pack_batches() record into the inbox."""
        self.inbox.copy_(record, non_blocking=True)
        self._record = self.inbox

    # ------------------------------------------------------------------ pieces
    def _gather_args(self) -> dict:
        """Extra outputs of the gather (subclasses: the fused step adds the FM value and S)."""
        return {}

    def _gather_call(self, record: torch.Tensor):
        base = record.data_ptr()
        return ([base + o for o in self._rec_offsets], self.inputs, self.B, self.fo, self.fe), \
            dict(extra_src_ptr=base + self._rec_labels, extra_dst=self.labels, **self._gather_args())

    def _plan_from_record(self, record: torch.Tensor) -> None:
        base = record.data_ptr()
        self.emb.build_rowplan(self.inputs, self.B, ids_ptrs=[base + o for o in self._rec_id_offsets], touch=True)
        self._plan_done = True

    def _gather(self, record: Optional[torch.Tensor] = None) -> None:
        record = self._record if record is None else record
        self._plan_done = False
        if self.rowplan_first and self._handoff_ptr is not None and self._handoff_ptr == record.data_ptr():
            self._plan_done = True             # the previous launch's last apply built this record's plan (hand-off set)
        elif self.rowplan_first:
            self._plan_from_record(record)
        self._handoff_ptr = None
        a, kw = self._gather_call(record)
        self.emb.forward_staged(*a, **kw)

    def _capture_gather(self, record: torch.Tensor, with_plan: bool = True):
        """``_gather`` while the stream is being captured; returns the graph node(s) that read the batch
        record (the ones ``_update_gather`` re-points before every launch): (gather node, row-plan node or None).
        ``with_plan`` False: the plan of this step was built by the previous step's apply launch."""
        plan_node = None
        self._plan_done = False
        if self.rowplan_first and not with_plan:
            self._plan_done = True
        elif self.rowplan_first:
            self._plan_from_record(record)
            plan_node = C.c_void_p()
            _lib.check(_lib.load().dfm_graph_last_node(_lib.stream_handle(), C.byref(plan_node)))
        a, kw = self._gather_call(record)
        self.emb.forward_staged(*a, **kw)
        node = C.c_void_p()
        _lib.check(_lib.load().dfm_graph_last_node(_lib.stream_handle(), C.byref(node)))
        return (node, plan_node)

    def _update_gather(self, graph_exec: int, node, record: torch.Tensor) -> None:
        node, plan_node = node
        if plan_node is not None:
            base = record.data_ptr()
            self.emb.rowplan_update(graph_exec, plan_node, [base + o for o in self._rec_id_offsets], self.B, True)
        a, kw = self._gather_call(record)
        self.emb.forward_staged_update(graph_exec, node, *a, **kw)

    def _build_rowplan(self) -> None:
        """Row plan of the step's ids (sort / unique / segments per SPARSE field) — unless ``_gather`` already
        built it from the batch record (``rowplan_first``)."""
        if self._plan_done:
            self._plan_done = False
            return
        self.emb.build_rowplan(self.inputs, self.B)

    def _embedding_backward(self, g_fo: torch.Tensor, g_fe: torch.Tensor) -> None:
        """d first_order (B, 1), d field_embeddings (B, F, D) -> DENSE-field Linear gradients and one
        gradient row per distinct id."""
        self.emb.backward_rowsparse(self.inputs, g_fo, g_fe, self.dense_grads, dense_slices=self._dense_slices())

    def _dense_slices(self):
        """Batch-sliced DENSE-field gradients (see FeatureEmbedding.backward_rowsparse), or None."""
        return None

    def _body_a(self) -> None:
        self.opt.zero_grad()
        cur = torch.cuda.current_stream()
        side = os.environ.get("DFM_ROWPLAN_SIDE_STREAM") == "1"    # see fused_step.py: in line is faster
        if side:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self._build_rowplan()
        else:
            self._build_rowplan()
        fo = self.fo.detach().requires_grad_()
        fe = self.fe.detach().requires_grad_()
        logits = self.model._forward_components(fo, fe, fe.view(self.B, -1))
        # plain BCE: the L2 term (base.py:78-83) is applied as g += 2*l2*p by the optimizer
        loss = bce_with_logits_mean(logits.view(-1), self.labels)
        loss.backward()
        self.loss.copy_(loss.detach())
        if side:
            cur.wait_stream(self.side)
        self._embedding_backward(fo.grad, fe.grad)

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

    def capture(self, warmup_iters: int = 1, timed_variant: bool = False, steps_per_graph: int = 1) -> None:
        """Capture the step's graphs after ``warmup_iters`` eager steps (``timed_variant``: also a copy of
        graph A without the gather, for ``run(eager_gather=True)``; ``steps_per_graph`` > 1: that many
        consecutive steps in one graph, launched with ``run_group`` — back-to-back graph launches are
        ~14 us apart on the device, whatever they contain).  Side-effect free: the warm-up
        steps run on an all-padding batch (id 0 everywhere: no table row receives a gradient, so the
        row-wise Adam touches nothing) and every other piece of state a step writes — dense parameters,
        Adam moments, step count, dropout seed, BatchNorm running statistics, the static inputs — is
        restored afterwards, bit for bit."""
        if not self.use_graph:
            return
        state = self._mutable_state()
        saved = [t.clone() for t in state]
        try:
            self._capture(warmup_iters, timed_variant, steps_per_graph)
        except BaseException:
            # a refused capture (e.g. a runtime that will not capture the collectives) must not leave half a
            # set of graphs behind: the caller may go on eagerly (bench.py does, after agreeing over all ranks)
            self.slots, self.body_graph, self.graph_b = [], None, None
            self.cont_slots, self._turn_cont, self._handoff_ptr = [], 0, None
            self.steps_per_graph, self._turn = 1, 0
            self._plan_done = False
            raise
        finally:
            # whatever happened, the warm-up steps' writes are undone, bit for bit
            torch.cuda.synchronize()
            with torch.no_grad():
                for t, v in zip(state, saved):
                    t.copy_(v)
            torch.cuda.synchronize()

    def _capture(self, warmup_iters: int, timed_variant: bool, steps_per_graph: int) -> None:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._gather(self.pad)
                self._body_a()
                self.opt.exchange()
                self._body_b()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        single = not self.opt.split
        # opt-in (DFM_DP_GRAPH_COLLECTIVE=1): capture the exchange inside the graph — one graph launch
        # per step under data parallelism.  Verified with a single-rank RCCL communicator only
        # (DFM_FORCE_DP_PATH=1); not validated on a multi-GPU box, hence not the default.
        # A step whose collectives sit inside its body (exchange_in_body: the field-sharded step) has no
        # split form: it is captured whole.
        fused_exchange = self.exchange_in_body or (
            (not single) and os.environ.get("DFM_DP_GRAPH_COLLECTIVE") == "1"
            and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl")
        # thread_local capture mode: another thread (the RCCL watchdog polling its events under
        # data parallelism) must not invalidate the capture
        mode = dict(capture_error_mode="thread_local")
        def body():
            self._body_a()
            if single or fused_exchange:
                self.opt.exchange()          # one rank: no device work, selects the local row lists
                self._body_b()
        if steps_per_graph < 1 or (steps_per_graph > 1 and not single and not fused_exchange):
            raise ValueError("steps_per_graph > 1 needs the whole step inside one graph (one rank, or the in-graph exchange)")
        self.steps_per_graph = steps_per_graph
        self.slots = []
        lib = _lib.load()
        # plan look-ahead: whole steps in one graph, several of them, plan built from the batch record
        look = self.plan_lookahead and steps_per_graph > 1 and (single or fused_exchange) and self.emb.rowsparse is not None
        sets = None
        if look:
            # three sets of plan buffers: H ("hand-off": the plan a launch starts from — built by its own first node or
            # by the LAST apply of the previous launch) and A / B alternating inside the graph
            from deepfm_amd.models.layers.embedding import RowSparseBuffers
            rs = self.emb.rowsparse
            mk = lambda: RowSparseBuffers(rs.num_sparse, rs.dim, rs.batch, rs.row_g2.device)
            sets = self._plan_sets = [rs, mk(), mk()]          # [H, A, B]
        self.cont_slots = []
        ids0 = self.pad.data_ptr() + (self._rec_id_offsets[0] if self._rec_id_offsets else 0)

        def capture_slot(cont: bool) -> _GraphSlot:
            slot = _GraphSlot()
            # keep_graph: the captured graph stays alive, so the gather's node handle stays valid for
            # hipGraphExecKernelNodeSetParams on the exec instantiated from it
            slot.graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(slot.graph, **mode):
                for k in range(steps_per_graph):
                    if look:
                        self.emb.rowsparse = sets[0] if k == 0 else sets[1 + (k - 1) % 2]
                    # (subclasses with their own gather — the field-sharded step — keep their signature)
                    read_nodes = (self._capture_gather(self.pad, with_plan=not (look and (k > 0 or cont)))
                                  if self.rowplan_first else self._capture_gather(self.pad))
                    apply_node = cur = target = None
                    if look:
                        # this step's apply launch also builds the NEXT step's plan: the following step of the graph
                        # (other set), or — last step — the first step of the next launch (hand-off set)
                        target = sets[1 + k % 2] if k + 1 < steps_per_graph else sets[0]
                        self.opt.next_plan = (ids0, target)
                    body()
                    if target is not None:
                        apply_node = C.c_void_p()
                        _lib.check(lib.dfm_graph_last_node(_lib.stream_handle(), C.byref(apply_node)))
                        cur = self.opt._cur
                    slot.nodes.append((read_nodes, apply_node, cur, target))
            slot.graph.instantiate()
            return slot

        for _ in range(2):
            self.slots.append(capture_slot(False))
        if look:
            for _ in range(2):                 # "continuation" flavour: no plan node, the plan is already in the hand-off set
                self.cont_slots.append(capture_slot(True))
            self.emb.rowsparse = sets[0]       # single steps (eager, timed variant) and every launch's first step
        if timed_variant:
            self.body_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.body_graph, **mode):
                self._plan_done = self.rowplan_first      # run(eager_gather=True) launches plan + gather eagerly in front
                body()
        if not single and not fused_exchange:
            self.opt.exchange()
            torch.cuda.synchronize()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, **mode):
                self._body_b()

    def release_graphs(self) -> None:
        """Drop every captured graph (the step runs eagerly afterwards).  Call before
        ``torch.distributed.destroy_process_group()`` when collectives were captured: the graphs hold the
        communicator's kernels, and tearing the communicator down under them does not return."""
        if not self.slots and self.body_graph is None and self.graph_b is None:
            return
        torch.cuda.synchronize()
        self.slots, self.body_graph, self.graph_b = [], None, None
        self.cont_slots, self._turn_cont, self._handoff_ptr = [], 0, None
        self.steps_per_graph, self._turn, self._prepared = 1, 0, None
        import gc
        gc.collect()
        torch.cuda.synchronize()

    def run_from(self, record: torch.Tensor, eager_gather: bool = False) -> None:
        """One step on a ``pack_batches()`` record: the gather reads its inputs from the record and
        refreshes the static input buffers — ids, dense values, labels — the rest of the step reads.
        (A record that the previous ``run_group(..., next_record=record)`` planned for starts at its gather.)"""
        if record.numel() != self.packed_bytes or record.dtype != torch.uint8 or not record.is_contiguous():
            raise ValueError("run_from expects one contiguous pack_batches() record")
        if record.data_ptr() % 16:
            raise ValueError("batch records must be 16-byte aligned")
        self._record = record
        self.run(eager_gather)

    def run(self, eager_gather: bool = False) -> None:
        if not self.slots:
            self._gather()
            self._body_a()
            self.opt.exchange()
            self._body_b()
            return
        if eager_gather:
            if self.body_graph is None:
                raise RuntimeError("run(eager_gather=True) needs capture(timed_variant=True)")
            self._gather()                 # eager: dfm_gather_timing_begin may attach events to this dispatch
            self.body_graph.replay()       # (its apply plans for nobody: the next launch builds its own plan)
            self._after_graph_a(None)
            return
        if self.steps_per_graph != 1:
            raise RuntimeError(f"the graphs hold {self.steps_per_graph} steps each: use run_group()")
        self._launch([self._record])

    def run_group(self, records, next_record: Optional[torch.Tensor] = None) -> None:
        """``steps_per_graph`` consecutive steps, one graph launch: ``records[k]`` is step k's batch record.
        ``next_record`` (optional): the record the NEXT launch (``run_group`` or ``run_from``) starts with — this
        launch's last optimizer kernel then builds its row plan, and the next launch starts at its gather."""
        if len(records) != self.steps_per_graph:
            raise ValueError(f"run_group expects {self.steps_per_graph} records")
        if not self.slots:
            for r in records:
                self.run_from(r)
            return
        for r in records:
            if r.numel() != self.packed_bytes or r.dtype != torch.uint8 or not r.is_contiguous() or r.data_ptr() % 16:
                raise ValueError("run_group expects contiguous, 16-byte aligned pack_batches() records")
        self._launch(list(records), next_record)

    def prepare_group(self, records, next_record: Optional[torch.Tensor] = None) -> None:
        """The host half of ``run_group``: picks the graph copy, waits for its last launch and points its nodes at
        ``records`` — everything but the launch itself, which ``launch_prepared()`` does.  A training loop calls this
        for launch k + 1 while launch k is on the device (two copies of the graph alternate for that); ``run_group``
        is the two calls back to back."""
        if len(records) != self.steps_per_graph or not self.slots:
            raise ValueError(f"prepare_group needs captured graphs of {len(records)} steps")
        if self._prepared is not None:
            raise RuntimeError("a prepared launch is pending: call launch_prepared() first")
        self._prepared = self._prepare(list(records), next_record)

    def launch_prepared(self) -> None:
        if self._prepared is None:
            raise RuntimeError("nothing prepared")
        slot, self._prepared = self._prepared, None
        slot.graph.replay()
        self._after_graph_a(slot.done)

    def _launch(self, records, next_record: Optional[torch.Tensor] = None) -> None:
        if self._prepared is not None:
            raise RuntimeError("a prepared launch is pending: call launch_prepared() first")
        slot = self._prepare(records, next_record)
        slot.graph.replay()
        self._after_graph_a(slot.done)

    def _prepare(self, records, next_record: Optional[torch.Tensor] = None):
        cont = bool(self.cont_slots) and self._handoff_ptr is not None and self._handoff_ptr == records[0].data_ptr()
        if cont:
            slot = self.cont_slots[self._turn_cont]
            self._turn_cont ^= 1
        else:
            slot = self.slots[self._turn]
            self._turn ^= 1
        if slot.done is not None:
            slot.done.synchronize()        # its previous launch (two launches ago) has left the device
        else:
            slot.done = torch.cuda.Event()
        ex = slot.graph.raw_cuda_graph_exec()
        for k, ((read_nodes, apply_node, cur, target), rec) in enumerate(zip(slot.nodes, records)):
            self._update_gather(ex, read_nodes, rec)
            if apply_node is not None:       # step k's apply launch sorts the next step's ids: point it at that record
                nxt = records[k + 1] if k + 1 < len(records) else (next_record if next_record is not None else records[0])
                self.opt.apply_plan_update(ex, apply_node, cur, nxt.data_ptr() + self._rec_id_offsets[0], target)
        self._handoff_ptr = next_record.data_ptr() if (next_record is not None and self.cont_slots) else None
        return slot

    def _after_graph_a(self, done) -> None:
        if self.graph_b is not None:
            # the replay produced the row gradients; the Python-side flag was only set while
            # the graph was being captured
            self.emb.rowsparse.has_grad = True
            self.opt.exchange()
            self.graph_b.replay()
        if done is not None:
            done.record()
