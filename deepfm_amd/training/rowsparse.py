"""RowSparseAdam: the optimizer half of the row-sparse fast mode.

Reference step (``deepfm/training/trainer.py:212-240``): BCE + ``get_l2_reg_loss`` ->
backward -> ``clip_grad_norm_`` over all parameters -> dense ``Adam``.  At the Criteo
shape 98 % of that step is dense full-table traffic.  Here the (V, d)
tables are updated only on the rows the (global) batch touched:

    g_row = grad_scale * sum(contributions) + 2*l2*w_row        (lazy L2, base.py:78-83)
    clip  = min(1, max_norm / (||all grads|| + 1e-6))            (trainer.py:232-235)
    Adam(w_row, m_row, v_row, clip * g_row)                      (trainer.py:67-70,237)

All other parameters (DENSE-field Linears, DNN, heads, CIN, attention) live in ONE flat
buffer (each ``nn.Parameter`` becomes a view of it) and take a dense Adam step from one
kernel; the L2 term of the embedding's dense parameters is added to their gradient there
(``g += 2*l2*p``), so the training loss passed to ``backward`` is the plain BCE.
This is NOT trajectory-identical to the reference's dense Adam (untouched rows do not
move); DESIGN.md states the delta.  Everything runs on the current stream with no host
synchronisation, so a whole step can be captured in a HIP graph.

Data parallel (one process per GPU, tables replicated): the flat dense gradient is
all-reduced, the row lists (ids + gradient rows) are all-gathered, and every rank runs
the same deterministic merge (``csrc/rowadam.hip``) so replicas stay bit-identical.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from deepfm_amd import _lib
from deepfm_amd.data.schema import FeatureType
from deepfm_amd.training import exchange
from deepfm_amd.models.layers.embedding import FeatureEmbedding


class RowSparseAdam:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 l2: float = 0.0, max_grad_norm: Optional[float] = None,
                 process_group: Optional[dist.ProcessGroup] = None,
                 row_embedding: Optional[FeatureEmbedding] = None) -> None:
        """``row_embedding``: the module whose SPARSE tables take the row-wise updates and whose row
        lists (``.rowsparse``) feed them — ``model.embedding`` unless the tables are field-sharded
        (training/sharded.py passes this rank's shard, whose tables are the model's own Parameters)."""
        emb = model.embedding
        if not isinstance(emb, FeatureEmbedding) or emb.grad_mode != "rowsparse":
            raise ValueError("model.embedding must be a FeatureEmbedding in 'rowsparse' grad mode")
        self.model, self.emb = model, emb
        self.row_emb = row_embedding if row_embedding is not None else emb
        self.lr, self.betas, self.eps, self.l2 = lr, betas, eps, l2
        self.max_grad_norm = max_grad_norm
        self.group = process_group
        self.world = exchange.world_size(process_group)
        # DFM_FORCE_DP_PATH=1 with an initialised (even single-rank) process group: take the N > 1 code
        # path — two graphs around eager collectives — so that RCCL and its interplay with graph capture
        # can be exercised on a one-GPU box
        import os
        self.split = self.world > 1 or (os.environ.get("DFM_FORCE_DP_PATH") == "1" and dist.is_available()
                                        and dist.is_initialized())

        tables = self.row_emb.table_parameters()
        if not tables:
            raise ValueError("no SPARSE tables to optimise")
        dev = tables[0].device
        _lib.require_device(tables[0], "embedding tables")
        self.device = dev
        self.num_sparse = len(tables) // 2
        self.dim = emb.fm_embed_dim
        # Adam moments: inside the packed row records when the embedding was packed
        # (FeatureEmbedding.pack_tables_), else separate contiguous tensors
        self.exp_avg, self.exp_avg_sq = [], []
        sparse_names = [n for n, spec in self.row_emb.schema.fields.items() if spec.feature_type is FeatureType.SPARSE]
        for name, w2, w1 in zip(sparse_names, tables[0::2], tables[1::2]):
            rec = self.row_emb.packed.get(name) if getattr(self.row_emb, "packed", None) else None
            if rec is not None and rec["buffer"].data_ptr() == w2.data_ptr():
                self.exp_avg += [rec["m2"], rec["m1"]]
                self.exp_avg_sq += [rec["v2"], rec["v1"]]
            else:
                self.exp_avg += [torch.zeros_like(w2, memory_format=torch.contiguous_format),
                                 torch.zeros_like(w1, memory_format=torch.contiguous_format)]
                self.exp_avg_sq += [torch.zeros_like(w2, memory_format=torch.contiguous_format),
                                    torch.zeros_like(w1, memory_format=torch.contiguous_format)]
        self._tables = tables
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sq_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.clip_coef = torch.ones(1, dtype=torch.float32, device=dev)

        # dense parameters: ONE flat parameter buffer and ONE flat gradient buffer; every
        # parameter / .grad becomes a view (embedding parameters first: they take the L2 term)
        table_ids = {id(p) for p in tables} | {id(p) for p in emb.table_parameters()}
        emb_dense = [p for p in emb.non_table_parameters() if p.requires_grad]
        emb_ids = {id(p) for p in emb_dense}
        others = [p for p in model.parameters()
                  if id(p) not in table_ids and id(p) not in emb_ids and p.requires_grad]
        # modules may ask for groups of parameters to lie back to back (attention: W_q | W_k | W_v as one
        # stacked weight without a copy): each group moves, in its order, to where its first member is
        groups = [g for m in model.modules() if hasattr(m, "adjacent_parameters") for g in m.adjacent_parameters()]
        for g in groups:
            ids = {id(p) for p in g}
            if all(any(p is q for q in others) for p in g):
                first = min(i for i, q in enumerate(others) if id(q) in ids)
                rest = [q for q in others if id(q) not in ids]
                first -= sum(1 for q in others[:first] if id(q) in ids)
                others = rest[:first] + list(g) + rest[first:]
        self.dense_params = emb_dense + others
        # every parameter starts on a 64-byte boundary (16 floats) so kernels can use 16-byte
        # vector loads on the views; the padding stays 0 (zero grad -> zero Adam update)
        def padded(n):
            return (n + 15) // 16 * 16
        self.n_l2 = sum(padded(p.numel()) for p in emb_dense)
        total = sum(padded(p.numel()) for p in self.dense_params)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.dense_params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            p.grad = self.flat_grad[off:off + n].view_as(p)
            off += padded(n)
        self._owner = None
        self.next_plan = None            # (next batch's ids pointer, target RowSparseBuffers): set by the step, see apply()
        self._vocab_dev, self._max_vocab, self._keep_tabs = None, 0, None
        specs = list(self.row_emb.schema.fields.values())
        vocab = [specs[i].vocabulary_size for i in self.row_emb._sparse_pos]
        if vocab:                            # vocabulary sizes on the device, for dfm_step_apply_plan (not creatable under capture)
            self._vocab_dev = torch.tensor(vocab, dtype=torch.int32, device=self.device)
            self._max_vocab = max(vocab)
        self._gathered = None
        self._partials = None
        self._match = None
        self._cur = None
        self.seed_tick: Optional[torch.Tensor] = None    # int64 device counter advanced once per apply()
        # (dfm_slab_ref[], count): d-weight slabs of dfm_linear_backward that apply() folds into the flat
        # gradient (single rank only: under data parallelism they must be in before the all-reduce)
        self.slab_refs = None

    # ------------------------------------------------------------------ helpers
    def zero_grad(self, force: bool = False) -> None:
        """``optimizer.zero_grad()`` (trainer.py:219).  The buffer starts zeroed and ``apply()``
        leaves it zeroed (the Adam kernel clears what it consumed), so no fill is launched; pass
        ``force=True`` to discard gradients accumulated by a backward pass that was not applied."""
        if force:
            self.flat_grad.zero_()
        if self.row_emb.rowsparse is not None:
            self.row_emb.rowsparse.has_grad = False

    def _table_struct(self):
        arr = (_lib.Table * self.num_sparse)()
        for s in range(self.num_sparse):
            t = arr[s]
            t.w2, t.w1 = self._tables[2 * s].data_ptr(), self._tables[2 * s + 1].data_ptr()
            t.m2, t.m1 = self.exp_avg[2 * s].data_ptr(), self.exp_avg[2 * s + 1].data_ptr()
            t.v2, t.v1 = self.exp_avg_sq[2 * s].data_ptr(), self.exp_avg_sq[2 * s + 1].data_ptr()
            t.stride2, t.stride1 = self._tables[2 * s].stride(0), self._tables[2 * s + 1].stride(0)
            for a, b in ((self.exp_avg[2 * s], t.stride2), (self.exp_avg_sq[2 * s], t.stride2),
                         (self.exp_avg[2 * s + 1], t.stride1), (self.exp_avg_sq[2 * s + 1], t.stride1)):
                if a.stride(0) != b:
                    raise RuntimeError("Adam state and table row strides differ (re-create the optimizer after "
                                       "pack_tables_() / .to())")
        return arr

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def exchange(self) -> None:
        """Data-parallel gradient exchange (no-op for one rank): all-reduce of the flat dense
        gradient, all-gather of the row lists.  Plain RCCL collectives on the current stream."""
        rs = self.row_emb.rowsparse
        if rs is None or not rs.has_grad:
            raise RuntimeError("RowSparseAdam: no row gradients (run a backward pass first)")
        local = (rs.uniq_rows, rs.num_uniq, rs.row_g2, rs.row_g1)
        if not self.split:
            self._cur = local + (rs.chunks,)
            return
        # one grouped all-gather: [dense gradient buffer | row lists]; the dense mean over ranks is formed
        # by the optimizer's prepare launch in rank order (no all-reduce, no scaling launch)
        world = max(self.world, 1)
        full = (self.flat_grad.view(1, -1),) + local
        if self._gathered is None or self._gathered[1].shape[0] != world * rs.chunks:
            self._gathered = exchange.alloc_gathered(full, world)
        exchange.allgather_step(full, self._gathered, self.group)
        self._cur = self._gathered[1:] + (world * rs.chunks,)

    @torch.no_grad()
    def apply(self) -> None:
        """Merge lists, L2 + global norm + clip, row-wise Adam on the tables, Adam on the flat
        dense buffer: five kernel launches, no host synchronisation."""
        lib = _lib.load()
        stream = _lib.stream_handle()
        grad_scale = 1.0 / self.world
        uniq, num, g2, g1, lists = self._cur
        n_dense = self.flat_param.numel()
        n_partials = lib.dfm_step_prepare_num_partials(self.num_sparse, self.dim, lists, n_dense)
        if self._owner is None or self._owner.shape != uniq.shape:
            self._owner = torch.empty_like(uniq)
            self._partials = torch.zeros(n_partials + self._extra_partial_count(lists), dtype=torch.float32,
                                         device=self.device)
            mbytes = lib.dfm_step_match_bytes(self.num_sparse, lists)
            self._match = torch.empty(mbytes, dtype=torch.uint8, device=self.device) if mbytes else None
        dense_gathered, gathered_stride = self._dense_source()
        tabs = self._table_struct()
        refs, n_refs = self.slab_refs if (self.slab_refs is not None and not self.split) else (None, 0)
        # three launches: [row-list merge | dense L2 (+ d-weight slabs) + norm partials] -> clip coefficient
        # (+ step / seed tick) -> [row-wise Adam | dense Adam (+ clears the gradient buffer)]
        _lib.check(lib.dfm_step_prepare(tabs, self.num_sparse, self.dim, lists, uniq.data_ptr(), num.data_ptr(),
                                        g2.data_ptr(), g1.data_ptr(), self._owner.data_ptr(), grad_scale, self.l2,
                                        self.flat_grad.data_ptr(), self.flat_param.data_ptr(), n_dense, self.n_l2,
                                        refs, n_refs, _lib.ptr(dense_gathered), max(self.world, 1), gathered_stride,
                                        self._partials.data_ptr(), self._dense_partial_offset(lists),
                                        _lib.ptr(self._match), stream))
        norm_ptr, n_norm = self._norm_partials(n_partials, lists)
        _lib.check(lib.dfm_grad_norm_finalize(norm_ptr, n_norm,
                                              self.max_grad_norm or 0.0, self.sq_norm.data_ptr(),
                                              self.clip_coef.data_ptr(), self.step_count.data_ptr(),
                                              _lib.ptr(self.seed_tick), stream))
        if self.next_plan is not None:
            # the row plan (+ row touch) of the NEXT step rides in this launch (csrc/step_tail.hip)
            ids_ptr, target = self.next_plan
            self.next_plan = None
            _lib.check(lib.dfm_step_apply_plan(*self._apply_plan_args(self._cur, ids_ptr, target), stream))
        else:
            _lib.check(lib.dfm_step_apply(tabs, self.num_sparse, self.dim, lists, uniq.data_ptr(), num.data_ptr(),
                                          g2.data_ptr(), g1.data_ptr(), self._owner.data_ptr(), self.clip_coef.data_ptr(),
                                          self.lr, self.betas[0], self.betas[1], self.eps, self.step_count.data_ptr(),
                                          self.flat_param.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                          self.flat_grad.data_ptr(), n_dense, 1, stream))
        self.row_emb.rowsparse.has_grad = False

    def _apply_plan_args(self, cur, ids_ptr: int, target):
        """Arguments of dfm_step_apply_plan (without the stream): ``cur`` = this step's lists (``_cur``), ``ids_ptr`` =
        device address of the next batch's (S, B) int64 id columns, ``target`` = the RowSparseBuffers the next step's
        plan goes to."""
        uniq, num, g2, g1, lists = cur
        if self._vocab_dev is None:
            raise RuntimeError("RowSparseAdam: no SPARSE fields to plan for")
        tabs = self._table_struct()
        self._keep_tabs = tabs
        B = target.batch
        return (tabs, self.num_sparse, self.dim, lists, uniq.data_ptr(), num.data_ptr(), g2.data_ptr(), g1.data_ptr(),
                self._owner.data_ptr(), self.clip_coef.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                self.step_count.data_ptr(), self.flat_param.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                self.flat_grad.data_ptr(), self.flat_param.numel(), 1, ids_ptr, B, self._vocab_dev.data_ptr(),
                self._max_vocab, B, target.sorted_pos.data_ptr(), target.uniq_rows.data_ptr(), target.seg_start.data_ptr(),
                target.num_uniq.data_ptr(), self.row_emb._err.data_ptr())

    def apply_plan_update(self, graph_exec: int, node, cur, ids_ptr: int, target) -> None:
        """The captured dfm_step_apply_plan node of an instantiated graph -> the next launch's record (host-side only)."""
        _lib.check(_lib.load().dfm_step_apply_plan_update(C.c_void_p(graph_exec), node,
                                                          *self._apply_plan_args(cur, ids_ptr, target)))

    def _dense_source(self):
        """(every rank's dense gradient buffer, floats between ranks) for the prepare launch's rank-ordered
        mean, or (None, 0): the local flat gradient is the whole gradient."""
        if (self.split and self._gathered is not None and self._cur is not None
                and self._cur[0] is self._gathered[1]):
            return self._gathered[0], 0
        return None, 0

    def _extra_partial_count(self, lists: int) -> int:
        """Floats kept free behind the norm partials (subclasses that exchange partial norms)."""
        return 0

    def _dense_partial_offset(self, lists: int) -> int:
        """Where the dense buffer's norm partials start in ``_partials`` (0: right behind the row partials)."""
        return 0

    def _norm_partials(self, n_partials: int, lists: int):
        """(address, count) of the floats whose sum is the squared global gradient norm."""
        return self._partials.data_ptr(), n_partials

    def step(self) -> None:
        self.exchange()
        self.apply()

    # ------------------------------------------------------------------ checkpointing
    def _table_state(self):
        """[(table parameter, exp_avg, exp_avg_sq)] of every table a checkpoint covers."""
        return list(zip(self._tables, self.exp_avg, self.exp_avg_sq))

    def _named(self):
        names = {id(p): n for n, p in self.model.named_parameters()}
        return ([(names[id(p)], m, v) for p, m, v in self._table_state()],
                [(names[id(p)], p) for p in self.dense_params])

    def state_dict(self) -> dict:
        """Adam state keyed by parameter NAME (``exp_avg`` / ``exp_avg_sq`` with the parameter's
        shape, like ``torch.optim.Adam``'s per-parameter state) plus the shared step count."""
        tables, dense = self._named()
        state = {}
        for name, m, v in tables:
            state[name] = {"exp_avg": m.detach().clone().contiguous(), "exp_avg_sq": v.detach().clone().contiguous()}
        off = 0
        for name, p in dense:
            n = p.numel()
            state[name] = {"exp_avg": self.flat_m[off:off + n].view_as(p).clone(),
                           "exp_avg_sq": self.flat_v[off:off + n].view_as(p).clone()}
            off += (n + 15) // 16 * 16
        return {"step": int(self.step_count.item()), "state": state,
                "hyper": {"lr": self.lr, "betas": list(self.betas), "eps": self.eps, "l2": self.l2,
                          "max_grad_norm": self.max_grad_norm}}

    def load_state_dict(self, sd: dict) -> None:
        """This optimizer's own ``state_dict()``, or a ``torch.optim.Adam.state_dict()`` over
        ``model.parameters()`` — what the reference's trainer writes as ``optimizer_state_dict``
        (trainer.py:140-148; position-keyed ``state`` + ``param_groups``): moments are matched to
        parameters by position in ``model.named_parameters()``; parameters torch never stepped get zeros."""
        if "param_groups" in sd:
            names = [n for n, _ in self.model.named_parameters()]
            order = [i for g in sd["param_groups"] for i in g["params"]]
            if len(order) != len(names):
                raise KeyError(f"torch optimizer state covers {len(order)} parameters, the model has {len(names)}")
            shapes = dict(self.model.named_parameters())
            state, step = {}, 0
            for i, name in zip(order, names):
                st = sd["state"].get(i)
                if st is None:
                    z = torch.zeros_like(shapes[name], memory_format=torch.contiguous_format)
                    state[name] = {"exp_avg": z, "exp_avg_sq": z.clone()}
                else:
                    state[name] = {"exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
                    step = max(step, int(st["step"]))
            sd = {"step": step, "state": state}
        tables, dense = self._named()
        want = {n for n, _, _ in tables} | {n for n, _ in dense}
        if set(sd["state"]) != want:
            raise KeyError(f"optimizer state keys differ: {sorted(set(sd['state']) ^ want)[:5]} ...")
        with torch.no_grad():
            for name, m, v in tables:
                m.copy_(sd["state"][name]["exp_avg"])
                v.copy_(sd["state"][name]["exp_avg_sq"])
            off = 0
            for name, p in dense:
                n = p.numel()
                self.flat_m[off:off + n].copy_(sd["state"][name]["exp_avg"].reshape(-1))
                self.flat_v[off:off + n].copy_(sd["state"][name]["exp_avg_sq"].reshape(-1))
                off += (n + 15) // 16 * 16
            self.step_count.fill_(int(sd["step"]))
