"""Data parallelism with FIELD-SHARDED embedding tables (one process per GPU, RCCL over xGMI).

The reference trains on one device (``deepfm/training/trainer.py:47-56``); ``north_star`` asks for
data-parallel minibatches on the 8 GPUs of a node.  With replicated tables (``RowSparseAdam`` +
``exchange.allgather_step``) every replica has to receive and apply every other rank's row updates —
852 K random 256-byte records per step at 8 ranks, more than the whole single-GPU step — so here the
tables are split instead, the way recommendation models are usually laid out:

  * rank r OWNS the tables (rows + Adam moments) of a contiguous block of SPARSE fields
    (``FieldShards``) and nothing of the others: table memory / N, every row update local;
  * everything dense (DENSE-field Linears, FM/CIN/attention parameters, DNN, heads) is replicated
    and kept bit-identical by a rank-ordered mean of the gradients;
  * per step three all-to-alls move the batch's ACTIVATIONS, not row updates (csrc/shard.hip):

        ids    (S, B) int64 of the local batch          -> the fields' owners
        rows   e (B, nf, D) + first-order w (B, nf)     <- the owners              (dfm_shard_gather)
        grads  d e, d first_order, dense gradients      -> the owners              (dfm_shard_pack)

    then the owner reduces the gradients of the GLOBAL batch to one row per distinct id
    (``dfm_rowplan_build`` over N*B samples, ``dfm_shard_rowgrad``) and runs the row-wise Adam on its
    own rows; one all-gather of a float per rank carries the owned rows' share of |g|^2 for the clip.

Per rank and direction that is ~7 MB at the Criteo shape (B = 4096 per GPU, 26 fields, D = 16) whatever
N is, against 7 x 8.5 MB of row lists plus N times the optimizer work with replicated tables.  The
whole step — collectives included — is one HIP graph (RCCL kernels are captured like any other).

The local half of the forward is the unchanged fused gather (csrc/embedding.hip): the model's
``FeatureEmbedding`` is pointed at the receive buffer of the rows all-to-all
(``FeatureEmbedding.bind_row_source``) and fed each sample's own index as its "id", so FM sums,
staging and first-order reduction stay in that one kernel.
"""

from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from deepfm_amd import _lib
from deepfm_amd.data.schema import DatasetSchema, FeatureType
from deepfm_amd.models.layers.embedding import FeatureEmbedding
from deepfm_amd.training import exchange
from deepfm_amd.training.rowsparse import RowSparseAdam

log = logging.getLogger("deepfm_amd.sharded")


class FieldShards:
    """Contiguous blocks of the SPARSE fields, one per rank: rank r owns [first[r], first[r] + count[r])
    (indices among the SPARSE fields, schema order).  Block sizes differ by at most one."""

    def __init__(self, num_sparse: int, world: int) -> None:
        if world < 1 or world > num_sparse:
            raise ValueError(f"field sharding needs 1 <= world ({world}) <= SPARSE fields ({num_sparse})")
        if world > _lib.MAX_RANKS:
            raise ValueError(f"world {world} > DFM_MAX_RANKS")
        base, extra = divmod(num_sparse, world)
        self.world, self.num_sparse = world, num_sparse
        self.count = [base + (1 if r < extra else 0) for r in range(world)]
        self.first = [sum(self.count[:r]) for r in range(world)]

    def owner(self, s: int) -> Tuple[int, int]:
        """(rank, position inside that rank's block) of SPARSE field s."""
        for r in range(self.world):
            if s < self.first[r] + self.count[r]:
                return r, s - self.first[r]
        raise IndexError(s)


class TableShard:
    """This rank's tables: a ``FeatureEmbedding`` over the owned SPARSE fields only, sharing the
    model's own table Parameters (and packed row records) for them; the model's tables of every other
    field are released (zero rows) until ``restore_tables``."""

    def __init__(self, model, rank: int, world: int, group: Optional[dist.ProcessGroup] = None) -> None:
        emb = model.embedding
        if emb.grad_mode != "rowsparse" or not emb.packed:
            raise ValueError("field sharding needs model.embedding packed (pack_tables_) and in 'rowsparse' mode")
        self.model, self.rank, self.world, self.group = model, rank, world, group
        names = [n for n, sp in emb.schema.fields.items() if sp.feature_type is FeatureType.SPARSE]
        self.sparse_names = names
        self.shards = FieldShards(len(names), world)
        lo = self.shards.first[rank]
        self.owned = names[lo:lo + self.shards.count[rank]]
        sub = DatasetSchema(fields={n: emb.schema.fields[n] for n in self.owned})
        with torch.device("meta"):                       # holders only: the tables are adopted below
            shard = FeatureEmbedding(sub, emb.fm_embed_dim)
        for n in self.owned:
            shard.second_order_embeddings[n] = emb.second_order_embeddings[n]
            shard.first_order_embeddings[n] = emb.first_order_embeddings[n]
            shard.packed[n] = emb.packed[n]
        shard.grad_mode = "rowsparse"
        self.emb = shard
        self.released = False

    def release_foreign(self) -> None:
        """Free the tables this rank does not own (their rows arrive by all-to-all)."""
        emb = self.model.embedding
        for n in self.sparse_names:
            if n in self.owned:
                continue
            for holder in (emb.second_order_embeddings[n], emb.first_order_embeddings[n]):
                w = holder.weight
                w.data = torch.empty(0, w.shape[1], dtype=w.dtype, device=w.device)
            emb.packed.pop(n, None)
        emb._drop_plan()                 # a plan made by a plain forward still points at the tables just freed
        self.released = True
        self._guard_state_dict()

    def _guard_state_dict(self) -> None:
        """``model.state_dict()`` on a released shard would silently save 0-row tables: make it raise instead."""
        if getattr(self.model, "_dfm_shard_guard", None) is not None:
            return
        shard = self

        def hook(module, prefix, keep_vars):
            if shard.released:
                raise RuntimeError("state_dict() of a model whose foreign embedding tables are released: call "
                                   "shard.restore_tables() first (collective), shard.release_foreign() afterwards")
        self.model._dfm_shard_guard = self.model.register_state_dict_pre_hook(hook)

    @torch.no_grad()
    def restore_tables(self) -> None:
        """Collective: every rank gets every table back from its owner (weights + Adam moments in the
        packed records), e.g. before ``state_dict()`` / evaluation with the plain forward.  The shard
        keeps working afterwards — also a step captured as a graph: its kernels hold the receive buffers, not
        the restored copies — and ``release_foreign`` frees the copies again
        (tests/test_gpu_sharded.py::test_restore_state_dict_release_then_continue)."""
        emb = self.model.embedding
        for s, n in enumerate(self.sparse_names):
            spec = emb.schema.fields[n]
            d = spec.embedding_dim
            src, _ = self.shards.owner(s)
            if n in emb.packed:
                buf = emb.packed[n]["buffer"]
            else:
                rs = ((3 * d + 4 + 31) // 32) * 32
                buf = torch.zeros(spec.vocabulary_size, rs, dtype=torch.float32,
                                  device=emb.second_order_embeddings[self.owned[0]].weight.device)
            if self.world > 1:
                dist.broadcast(buf, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                               group=self.group)
            if n not in emb.packed:
                emb.second_order_embeddings[n].weight.data = buf[:, :d]
                emb.first_order_embeddings[n].weight.data = buf[:, d:d + 1]
                emb.packed[n] = dict(buffer=buf, m1=buf[:, d + 1:d + 2], v1=buf[:, d + 2:d + 3],
                                     m2=buf[:, d + 4:2 * d + 4], v2=buf[:, 2 * d + 4:3 * d + 4])
        emb._drop_plan()
        self.released = False


class ShardedRowAdam(RowSparseAdam):
    """``RowSparseAdam`` whose row-wise half runs on this rank's table shard only.  The dense gradients
    of all ranks arrive inside the gradient all-to-all (``dense_source``, set by the step) and are
    averaged in rank order by the prepare launch; the squared norm of the owned rows' gradients is
    all-gathered (one float per rank) so that every rank clips by the same global norm."""

    def __init__(self, model, shard: TableShard, **kw) -> None:
        super().__init__(model, row_embedding=shard.emb, process_group=shard.group, **kw)
        self.shard = shard
        self.world = shard.world
        self.split = True                     # d-weight slabs must be in the flat gradient before it travels
        self.dense_source: Optional[Tuple[torch.Tensor, int]] = None

    def _table_state(self):
        """Checkpoints cover EVERY table: the moments live in the packed row records, which
        ``TableShard.restore_tables()`` (collective) brings back to every rank first."""
        if self.shard.released:
            raise RuntimeError("ShardedRowAdam.state_dict / load_state_dict: call shard.restore_tables() first "
                               "(collective), and shard.release_foreign() afterwards to continue training")
        emb, out = self.model.embedding, []
        for name in self.shard.sparse_names:
            rec = emb.packed[name]
            out.append((emb.second_order_embeddings[name].weight, rec["m2"], rec["v2"]))
            out.append((emb.first_order_embeddings[name].weight, rec["m1"], rec["v1"]))
        return out

    @torch.no_grad()
    def exchange(self) -> None:
        rs = self.row_emb.rowsparse
        if rs is None or not rs.has_grad:
            raise RuntimeError("ShardedRowAdam: no row gradients (the step's gradient all-to-all has not run)")
        self._cur = (rs.uniq_rows, rs.num_uniq, rs.row_g2, rs.row_g1, rs.chunks)

    def _dense_source(self):
        if self.dense_source is None:
            raise RuntimeError("ShardedRowAdam: the step has not attached its gradient receive buffer")
        return self.dense_source

    def _row_partials(self, lists: int) -> int:
        """Row partials of the rank with the MOST fields: every rank pads its own to this length (the tail
        stays zero), so that the all-gather below has one size on all ranks."""
        return _lib.load().dfm_rowadam_num_partials(max(self.shard.shards.count), self.dim, lists)

    def _extra_partial_count(self, lists: int) -> int:
        lib = _lib.load()
        own = lib.dfm_rowadam_num_partials(self.num_sparse, self.dim, lists)
        return (self._row_partials(lists) - own) + self.world * self._row_partials(lists)

    def _dense_partial_offset(self, lists: int) -> int:
        return self._row_partials(lists)

    def _norm_partials(self, n_partials: int, lists: int):
        # _partials = [row partials of this rank's rows, zero-padded to `rows` | dense partials (identical on
        #              every rank) | every rank's padded row partials, all-gathered]
        # and the norm is the sum from the second block on, in the same order on every rank
        lib = _lib.load()
        rows = self._row_partials(lists)
        dense = n_partials - lib.dfm_rowadam_num_partials(self.num_sparse, self.dim, lists)
        if self.max_grad_norm is None:
            # no clip: nothing downstream reads the norm; the step / seed tick still happens in finalize
            return self._partials.data_ptr(), rows + dense
        exchange.all_gather_flat(self._partials[rows + dense:rows + dense + self.world * rows], self._partials[:rows],
                                 self.group)
        return self._partials.data_ptr() + 4 * rows, dense + self.world * rows


class ShardedStepMixin:
    """Mixed in FRONT of a fused step class (``sharded_step_class``): replaces the embedding's three
    touch points — gather, row plan, embedding backward — with their sharded forms."""

    exchange_in_body = True
    rowplan_first_default = False   # the row plan runs over the GLOBAL batch's ids of the owned fields, after the ids all-to-all
    slabs_travel = True          # the tower's d-weight slabs are summed by the gradient pack kernel

    def __init__(self, model, optimizer: ShardedRowAdam, batch_size: int, use_graph: bool = True) -> None:
        if not isinstance(optimizer, ShardedRowAdam):
            raise TypeError("sharded steps need a ShardedRowAdam")
        if batch_size % 4:
            raise ValueError("field-sharded steps need a batch size that is a multiple of 4")
        super().__init__(model, optimizer, batch_size, use_graph)
        shard = optimizer.shard
        self.shard = shard
        sh = shard.shards
        N, r, B = shard.world, shard.rank, batch_size
        D = self.emb.fm_embed_dim
        dev = optimizer.device
        nf = sh.count[r]
        self.nf = nf
        lib = _lib.load()
        n_dense = optimizer.flat_grad.numel()
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        # --- ids: the static (S, B) ids of the batch record are the send buffer
        self.ids_in_splits = [c * B for c in sh.count]
        self.ids_out_splits = [nf * B] * N
        self.ids_recv = torch.zeros(N * nf * B, **i64)
        self.gids = torch.zeros(nf, N * B, **i64)
        # --- rows
        self.rows_in_splits = [B * nf * (D + 1)] * N
        self.rows_out_splits = [B * c * (D + 1) for c in sh.count]
        self.rows_send = torch.zeros(sum(self.rows_in_splits), **f32)
        self.rows_recv = torch.zeros(sum(self.rows_out_splits), **f32)
        # --- gradients (+ the flat dense gradient)
        self.seg_mine = lib.dfm_shard_pack_segment(B, nf, D, n_dense)
        self.grad_in_splits = [lib.dfm_shard_pack_segment(B, c, D, n_dense) for c in sh.count]
        self.grad_out_splits = [self.seg_mine] * N
        self.grad_send = torch.zeros(sum(self.grad_in_splits), **f32)
        self.grad_recv = torch.zeros(sum(self.grad_out_splits), **f32)
        optimizer.dense_source = (self.grad_recv[B * nf * D + B:], self.seg_mine)
        self._first = (C.c_int32 * N)(*sh.first)
        self._count = (C.c_int32 * N)(*sh.count)
        self._fmap = (C.c_int32 * len(self.emb._sparse_pos))(*self.emb._sparse_pos)
        specs = shard.emb.schema.fields
        self._vocab = (C.c_int32 * nf)(*[specs[n].vocabulary_size for n in shard.owned])
        # --- the local gather reads the received rows: one "table" of B rows per SPARSE field, id = sample index
        views: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        off = 0
        for p in range(N):
            c = sh.count[p]
            for j in range(c):
                name = shard.sparse_names[sh.first[p] + j]
                views[name] = (self.rows_recv.as_strided((B, D), (c * D, 1), off + j * D),
                               self.rows_recv.as_strided((B, 1), (c, 1), off + B * c * D + j))
            off += B * c * (D + 1)
        if not shard.released:
            shard.release_foreign()                       # (drops the embedding's plan: before the row source is bound)
        self.emb.pin_plan(dev, False)
        self.emb.bind_row_source(views)
        self.emb.pin_plan(dev)
        shard.emb._ensure_plan(dev)                       # allocates the shard's error flag
        self.sample_ids = torch.arange(B, **i64)
        self.local_inputs: List[torch.Tensor] = []
        for t, spec in zip(self.inputs, self.model.schema.fields.values()):
            self.local_inputs.append(self.sample_ids if spec.feature_type is FeatureType.SPARSE else t)
        self.gid_inputs = [self.gids[j] for j in range(nf)]

    # ------------------------------------------------------------------ forward half
    def _stage(self, record: torch.Tensor) -> None:
        _lib.check(_lib.load().dfm_stage_record(record.data_ptr(), self.packed.data_ptr(), self.packed_bytes,
                                                _lib.stream_handle()))

    def _rows_in(self) -> None:
        """ids -> owners, rows of the owned tables -> the batches, local gather over the received rows."""
        lib, st, sh = _lib.load(), _lib.stream_handle(), self.shard
        grp = sh.group
        ids = self.ids.view(-1) if self.n_sparse else None
        exchange.all_to_all(self.ids_recv, ids, self.ids_out_splits, self.ids_in_splits, grp)
        tabs = self.opt._table_struct()
        _lib.check(lib.dfm_shard_gather(tabs, self._vocab, self.nf, self.emb.fm_embed_dim, sh.world, self.B,
                                        self.ids_recv.data_ptr(), self.rows_send.data_ptr(), self.gids.data_ptr(),
                                        sh.emb._err.data_ptr(), st))
        exchange.all_to_all(self.rows_recv, self.rows_send, self.rows_out_splits, self.rows_in_splits, grp)
        self.emb.forward_into(self.local_inputs, self.B, self.fo, self.fe, **self._gather_args())

    def _gather(self, record: Optional[torch.Tensor] = None) -> None:
        self._stage(self._record if record is None else record)
        self._rows_in()

    def _capture_gather(self, record: torch.Tensor) -> C.c_void_p:
        self._stage(record)
        node = C.c_void_p()
        _lib.check(_lib.load().dfm_graph_last_node(_lib.stream_handle(), C.byref(node)))
        self._rows_in()
        return node

    def _update_gather(self, graph_exec: int, node: C.c_void_p, record: torch.Tensor) -> None:
        _lib.check(_lib.load().dfm_stage_record_update(C.c_void_p(graph_exec), node, record.data_ptr(),
                                                       self.packed.data_ptr(), self.packed_bytes))

    # ------------------------------------------------------------------ backward half
    def _build_rowplan(self) -> None:
        self.shard.emb.build_rowplan(self.gid_inputs, self.shard.world * self.B)

    def _embedding_backward(self, g_fo: torch.Tensor, g_fe: torch.Tensor) -> None:
        lib, st, sh = _lib.load(), _lib.stream_handle(), self.shard
        B, F, D = g_fe.shape
        # DENSE-field Linear gradients -> the flat dense gradient (complete after this launch)
        self.emb.backward_rowsparse(self.local_inputs, g_fo, g_fe, self.dense_grads, sparse=False,
                                    dense_slices=self._dense_slices())
        flat = self.opt.flat_grad
        refs, n_refs = self._slab_refs if self._slab_refs is not None else (None, 0)
        _lib.check(lib.dfm_shard_pack(self._first, self._count, sh.world, self._fmap, len(self.emb._sparse_pos), F, D, B,
                                      g_fe.data_ptr(), g_fo.data_ptr(), flat.data_ptr(), flat.numel(), refs, n_refs,
                                      self.grad_send.data_ptr(), st))
        exchange.all_to_all(self.grad_recv, self.grad_send, self.grad_out_splits, self.grad_in_splits, sh.group)
        rs = sh.emb.rowsparse
        _lib.check(lib.dfm_shard_rowgrad(self.nf, D, sh.world, B, self.grad_recv.data_ptr(), self.seg_mine,
                                         rs.sorted_pos.data_ptr(), rs.seg_start.data_ptr(), rs.num_uniq.data_ptr(),
                                         rs.row_g2.data_ptr(), rs.row_g1.data_ptr(), st))
        rs.has_grad = True

    # ------------------------------------------------------------------ capture
    def _mutable_state(self) -> List[torch.Tensor]:
        return super()._mutable_state() + [self.ids_recv, self.gids, self.rows_send, self.rows_recv, self.grad_send,
                                           self.grad_recv]

    def capture(self, *a, **kw) -> None:
        if self.use_graph and self.shard.world > 1 and dist.get_backend(self.shard.group) != "nccl":
            log.info("field-sharded step over %s: collectives cannot be captured, running eagerly",
                     dist.get_backend(self.shard.group))
            self.use_graph = False
        super().capture(*a, **kw)

    def unbind(self) -> None:
        """Give the model's embedding its own tables back as row source (after ``shard.restore_tables()``:
        plain ``model(batch)`` forwards, ``state_dict()``); the step must not run afterwards."""
        self.emb.pin_plan(self.opt.device, False)
        self.emb.bind_row_source(None)


def sharded_step_class(model):
    """The fused step class of ``model`` (training/fused_step.py) with the sharded embedding wiring."""
    from deepfm_amd.training.fused_step import fused_step_class
    base = fused_step_class(model)
    if base is None:
        raise ValueError("field-sharded steps exist for the models with a fused step only")
    return type("Sharded" + base.__name__, (ShardedStepMixin, base), {})


def make_sharded_step(model, batch_size: int, use_graph: bool = True, group: Optional[dist.ProcessGroup] = None,
                      **optimizer_kw):
    """Shard ``model``'s (packed, row-sparse) tables over the ranks of ``group`` and build the optimizer
    and the training step on them.  Every rank must call this with a model built from the same seed."""
    ready = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if ready else 0
    world = dist.get_world_size(group) if ready else 1
    shard = TableShard(model, rank, world, group)
    opt = ShardedRowAdam(model, shard, **optimizer_kw)
    step = sharded_step_class(model)(model, opt, batch_size, use_graph=use_graph)
    return step, opt, shard
