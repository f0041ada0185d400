"""FeatureEmbedding on MI355X: one fused HIP gather for all fields.

Drop-in for the reference's ``deepfm/models/layers/embedding.py:11-126``: same
constructor, same ``forward(batch) -> (first_order, field_embeddings, flat_embeddings)``,
same ``state_dict`` keys/shapes (``second_order_embeddings.<f>.weight`` …) and the same
initialisation order, so a reference checkpoint loads unchanged and equal seeds give
equal weights.  The ``torch.nn`` sub-modules below are *parameter holders only*: their
``forward`` is never called — every value is produced by ``libdeepfm_hip.so``
(``dfm_embedding_forward`` / ``dfm_embedding_backward_dense`` / ``dfm_rowgrad_build``).

Gradient modes
  ``dense``      (default) reference semantics: autograd returns dense ``(V, d)``
                 gradients for every table (``nn.Embedding(sparse=False)``, embedding.py:35-40).
  ``rowsparse``  fast mode for uniform schemas (every field SPARSE/DENSE, dim == fm_dim):
                 tables leave autograd; backward writes one gradient row per distinct id
                 into ``self.rowsparse`` for ``deepfm_amd.training.RowSparseAdam``.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from deepfm_amd import _lib
from deepfm_amd.data.schema import DatasetSchema, FeatureType, FieldSchema

_KIND = {FeatureType.SPARSE: _lib.SPARSE, FeatureType.DENSE: _lib.DENSE, FeatureType.SEQUENCE: _lib.SEQUENCE}


def _holders(spec: FieldSchema) -> Tuple[nn.Module, nn.Module]:
    """(second-order, first-order) parameter holders of one field (embedding.py:32-56)."""
    if spec.feature_type is FeatureType.SPARSE:
        return (nn.Embedding(spec.vocabulary_size, spec.embedding_dim, padding_idx=0),
                nn.Embedding(spec.vocabulary_size, 1, padding_idx=0))
    if spec.feature_type is FeatureType.SEQUENCE:
        return (nn.EmbeddingBag(spec.vocabulary_size, spec.embedding_dim, mode=spec.combiner, padding_idx=0),
                nn.EmbeddingBag(spec.vocabulary_size, 1, mode=spec.combiner, padding_idx=0))
    if spec.feature_type is FeatureType.DENSE:
        return nn.Linear(1, spec.embedding_dim), nn.Linear(1, 1)
    raise ValueError(f"unsupported feature type {spec.feature_type!r}")


class RowSparseBuffers:
    """Per-step row plan + row gradients of the SPARSE fields (see csrc/rowplan.hip)."""

    def __init__(self, num_sparse: int, dim: int, batch: int, device: torch.device) -> None:
        ch = _lib.ROWPLAN_CHUNK
        self.chunks = (batch + ch - 1) // ch
        self.batch, self.num_sparse, self.dim = batch, num_sparse, dim
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.sorted_pos = torch.empty(self.chunks, num_sparse, ch, **i32)
        self.uniq_rows = torch.empty(self.chunks, num_sparse, ch, **i32)
        self.seg_start = torch.empty(self.chunks, num_sparse, ch + 1, **i32)
        self.num_uniq = torch.zeros(self.chunks, num_sparse, **i32)
        self.row_g2 = torch.empty(self.chunks, num_sparse, ch, dim, **f32)
        self.row_g1 = torch.empty(self.chunks, num_sparse, ch, **f32)
        self.has_grad = False


class FeatureEmbedding(nn.Module):
    def __init__(self, schema: DatasetSchema, fm_embed_dim: int = 16) -> None:
        super().__init__()
        self.schema = schema
        self.fm_embed_dim = fm_embed_dim
        self.field_names: List[str] = list(schema.fields)

        self.second_order_embeddings = nn.ModuleDict()
        self.first_order_embeddings = nn.ModuleDict()
        self.projections = nn.ModuleDict()
        for name, spec in schema.fields.items():
            second, first = _holders(spec)
            self.second_order_embeddings[name] = second
            self.first_order_embeddings[name] = first
            if spec.embedding_dim != fm_embed_dim:
                self.projections[name] = nn.Linear(spec.embedding_dim, fm_embed_dim, bias=False)
        self._init_weights()

        self.grad_mode = "dense"
        self.strict_indices = os.environ.get("DEEPFM_AMD_STRICT_INDICES", "0") == "1"
        self.rowsparse: Optional[RowSparseBuffers] = None
        self.packed: Dict[str, dict] = {}      # filled by pack_tables_()
        self._plan = None
        self._plan_key = None
        self._plan_uniform = False
        self._err: Optional[torch.Tensor] = None
        self._anchor: Optional[torch.Tensor] = None
        self._dense_list: Optional[torch.Tensor] = None
        self._plan_static = False
        self._sparse_pos = [i for i, s in enumerate(schema.fields.values())
                            if s.feature_type is FeatureType.SPARSE]
        self._row_source: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None

    # ------------------------------------------------------------------ init / bookkeeping
    def _init_weights(self) -> None:
        """xavier on rows 1.. of every table (row 0 = padding stays 0), xavier Linear
        weights, zero biases — embedding.py:66-74, same module traversal order."""
        for module in self.modules():
            if isinstance(module, (nn.Embedding, nn.EmbeddingBag)):
                nn.init.xavier_uniform_(module.weight.data[1:])
            elif isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight.data)
                if module.bias is not None:
                    nn.init.zeros_(module.bias.data)

    def __del__(self):  # pragma: no cover - best effort
        try:
            self._drop_plan()
        except Exception:
            pass

    def pin_plan(self, device: torch.device, pinned: bool = True) -> None:
        """Promise that no parameter of this module will be re-homed (``.to()``, ``pack_tables_()``,
        ``p.data = ...``) while ``pinned``: the kernel plan is then reused without re-validation."""
        self._plan_static = False
        if pinned:
            self._ensure_plan(device)
        self._plan_static = pinned

    def bind_row_source(self, views: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]]) -> None:
        """Field-sharded data parallelism (training/sharded.py): the rows of the SPARSE fields do not come
        from this module's tables but from ``views[name] = (e (n, d), w (n, 1))`` — strided views of the
        buffer the rows all-to-all delivers, one row per sample — and the id fed for such a field is the
        sample's own index.  ``None`` restores the module's tables."""
        if views is not None:
            missing = [n for n, sp in self.schema.fields.items()
                       if sp.feature_type is FeatureType.SPARSE and n not in views]
            if missing:
                raise KeyError(f"bind_row_source: no view for SPARSE field(s) {missing[:3]}")
        self._row_source = views
        self._drop_plan()

    def _drop_plan(self) -> None:
        self._plan_static = False
        if self._plan is not None:
            _lib.load().dfm_embedding_plan_destroy(self._plan)
            self._plan = None
            self._plan_key = None

    def _field_params(self) -> List[Tuple[str, FieldSchema, nn.Module, nn.Module, Optional[nn.Module]]]:
        out = []
        for name in self.field_names:
            out.append((name, self.schema.fields[name], self.second_order_embeddings[name],
                        self.first_order_embeddings[name],
                        self.projections[name] if name in self.projections else None))
        return out

    def table_parameters(self) -> List[nn.Parameter]:
        """(V, d) and (V, 1) tables of the SPARSE fields, schema order, [w2, w1] per field."""
        out = []
        for name, spec, second, first, _ in self._field_params():
            if spec.feature_type is FeatureType.SPARSE:
                out += [second.weight, first.weight]
        return out

    def pack_tables_(self) -> "FeatureEmbedding":
        """Re-home every SPARSE table in a packed row-record buffer on the current device.

        One record per id, ``RS = roundup(3*d + 4, 32)`` floats (256 B at d = 16):
        ``[w2 (d) | w1, m1, v1, pad | m2 (d) | v2 (d) | pad]``.  ``second_order_embeddings.<f>.weight``
        and ``first_order_embeddings.<f>.weight`` become strided views of it (same shapes, same
        values, same state_dict keys), so the forward's first-order scalar rides on the row's
        128-B line and ``RowSparseAdam`` — which places its moments in the same records — touches
        one contiguous record per row instead of six scattered locations.  MI355X-first layout:
        4x the table bytes, paid from 288 GB of HBM.  Call after ``.to(device)`` (which
        re-materialises parameters contiguously) and before creating the optimizer."""
        self.packed = {}
        for name, spec, second, first, _ in self._field_params():
            if spec.feature_type is not FeatureType.SPARSE:
                continue
            d, V = spec.embedding_dim, spec.vocabulary_size
            if d % 4:
                raise NotImplementedError("packed row records need embedding_dim % 4 == 0")
            rs = ((3 * d + 4 + 31) // 32) * 32
            buf = torch.zeros(V, rs, dtype=torch.float32, device=second.weight.device)
            buf[:, :d].copy_(second.weight.data)
            buf[:, d:d + 1].copy_(first.weight.data)
            second.weight.data = buf[:, :d]
            first.weight.data = buf[:, d:d + 1]
            self.packed[name] = dict(buffer=buf, m1=buf[:, d + 1:d + 2], v1=buf[:, d + 2:d + 3],
                                     m2=buf[:, d + 4:2 * d + 4], v2=buf[:, 2 * d + 4:3 * d + 4])
        self._drop_plan()
        return self

    def non_table_parameters(self) -> List[nn.Parameter]:
        tables = {id(p) for p in self.table_parameters()}
        return [p for p in self.parameters() if id(p) not in tables]

    def set_grad_mode(self, mode: str) -> "FeatureEmbedding":
        if mode not in ("dense", "rowsparse"):
            raise ValueError(f"grad_mode must be 'dense' or 'rowsparse', got {mode!r}")
        if mode == "rowsparse":
            for spec in self.schema.fields.values():
                if spec.feature_type is FeatureType.SEQUENCE or spec.embedding_dim != self.fm_embed_dim \
                        or self.fm_embed_dim % 4:
                    raise NotImplementedError(
                        "rowsparse gradients need a uniform schema (SPARSE/DENSE fields, "
                        "embedding_dim == fm_embed_dim, multiple of 4)")
        self.grad_mode = mode
        for p in self.table_parameters():
            p.requires_grad_(mode == "dense")
        return self

    # ------------------------------------------------------------------ plan
    def _ensure_plan(self, device: torch.device):
        # a training step that has captured raw parameter pointers (HIP graph) pins the plan: walking
        # ~100 parameters to re-validate it costs ~250 us of host time per call, more than the whole
        # GPU step
        if self._plan_static and self._plan is not None:
            return self._plan
        params = list(self.parameters())
        if self._row_source is not None:
            params = params + [t for pair in self._row_source.values() for t in pair]
        key = (device, tuple((p.data_ptr(), p.stride(0)) for p in params))
        if self._plan is not None and key == self._plan_key:
            return self._plan
        self._drop_plan()
        lib = _lib.load()
        n = len(self.field_names)
        if n > _lib.MAX_FIELDS:
            raise ValueError(f"{n} fields > DFM_MAX_FIELDS={_lib.MAX_FIELDS}")
        arr = (_lib.Field * n)()
        for i, (name, spec, second, first, proj) in enumerate(self._field_params()):
            w2, w1, rows = second.weight, first.weight, spec.vocabulary_size
            if self._row_source is not None and spec.feature_type is FeatureType.SPARSE:
                w2, w1 = self._row_source[name]
                rows = w2.shape[0]
            for p in (w2, w1):
                _lib.require_device(p, f"parameter of field {name!r}")
                # rows may be strided (packed row records), elements of a row are contiguous
                if p.dtype != torch.float32 or p.dim() != 2 or (p.shape[1] > 1 and p.stride(1) != 1):
                    raise TypeError("embedding parameters must be float32 with contiguous rows")
            fd = arr[i]
            fd.kind = _KIND[spec.feature_type]
            fd.dim = spec.embedding_dim
            fd.vocab = rows if spec.feature_type is not FeatureType.DENSE else 0
            fd.max_len = spec.max_length
            fd.combiner = _lib.COMBINER[spec.combiner] if spec.feature_type is FeatureType.SEQUENCE else 0
            fd.w2, fd.w1 = w2.data_ptr(), w1.data_ptr()
            if spec.feature_type is not FeatureType.DENSE:
                fd.stride2, fd.stride1 = w2.stride(0), w1.stride(0)
            if spec.feature_type is FeatureType.DENSE:
                fd.b2, fd.b1 = second.bias.data_ptr(), first.bias.data_ptr()
            fd.proj = proj.weight.data_ptr() if proj is not None else None
        handle = C.c_void_p()
        _lib.check(lib.dfm_embedding_plan_create(arr, n, self.fm_embed_dim, C.byref(handle)))
        self._plan, self._plan_key = handle, key
        self._plan_uniform = bool(lib.dfm_embedding_plan_is_uniform(handle))
        # ONE error flag per module and device, not per plan: captured graphs keep its address, and a re-made plan
        # (restore_tables() -> release_foreign() on a table shard) must go on reporting through the same word
        if self._err is None or self._err.device != torch.device(device):
            self._err = torch.zeros(1, dtype=torch.int32, device=device)
        return handle

    # ------------------------------------------------------------------ inputs
    def _gather_inputs(self, batch: Dict[str, torch.Tensor]) -> Tuple[List[torch.Tensor], int]:
        inputs: List[torch.Tensor] = []
        size = None
        for name in self.field_names:
            spec = self.schema.fields[name]
            x = batch[name]                       # KeyError for a missing field, like the reference
            _lib.require_device(x, f"batch[{name!r}]")
            if spec.feature_type is FeatureType.DENSE:
                if x.dtype != torch.float32:
                    x = x.float()
                if x.dim() != 1:
                    raise ValueError(f"DENSE field {name!r} expects shape (B,), got {tuple(x.shape)}")
            else:
                if x.dtype != torch.int64:
                    x = x.long()
                want = 2 if spec.feature_type is FeatureType.SEQUENCE else 1
                if x.dim() != want:
                    raise ValueError(f"field {name!r} expects {want}-D ids, got {tuple(x.shape)}")
                if want == 2 and x.shape[1] != spec.max_length:
                    raise ValueError(f"SEQUENCE field {name!r} expects (B, {spec.max_length}), got {tuple(x.shape)}")
            x = x.contiguous()
            if size is None:
                size = x.shape[0]
            elif x.shape[0] != size:
                raise ValueError(f"field {name!r}: batch size {x.shape[0]} != {size}")
            inputs.append(x)
        return inputs, int(size)

    @staticmethod
    def _ptr_array(tensors: List[torch.Tensor]):
        arr = (C.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr()
        return arr

    def raise_on_bad_index(self) -> None:
        """Synchronises. The reference raises IndexError from ATen on an out-of-range id."""
        if self._err is not None and int(self._err.item()) != 0:
            self._err.zero_()
            raise IndexError("index out of range in FeatureEmbedding (id < 0 or id >= vocabulary_size)")

    # ------------------------------------------------------------------ kernels
    def forward_into(self, inputs: List[torch.Tensor], B: int, fo: torch.Tensor, fe: torch.Tensor,
                     flat: Optional[torch.Tensor] = None, fm_out: Optional[torch.Tensor] = None,
                     ws: Optional[torch.Tensor] = None, fm_sum: Optional[torch.Tensor] = None) -> None:
        """Enqueue the gather into caller-owned buffers (no allocation: graph/bench path)."""
        plan = self._ensure_plan(inputs[0].device)
        if B > 0:
            _lib.check(_lib.load().dfm_embedding_forward(
                plan, self._ptr_array(inputs), B, fo.data_ptr(), fe.data_ptr(), _lib.ptr(flat),
                _lib.ptr(fm_out), _lib.ptr(fm_sum), _lib.ptr(ws), self._err.data_ptr(), _lib.stream_handle()))

    def forward_staged(self, src_ptrs: List[int], stage_out: List[torch.Tensor], B: int, fo: torch.Tensor,
                       fe: torch.Tensor, fm_out: Optional[torch.Tensor] = None, fm_sum: Optional[torch.Tensor] = None,
                       extra_src_ptr: int = 0, extra_dst: Optional[torch.Tensor] = None) -> None:
        """The gather of ``forward_into`` reading every field's input from ``src_ptrs`` (device addresses
        inside a batch record, schema order) and refreshing ``stage_out`` (the step's static input
        buffers) plus one float per sample (``extra``: the labels) on the way — the per-step
        "load the next batch" copy without a launch of its own.  Uniform plans only."""
        plan = self._ensure_plan(fe.device)
        if not self._plan_uniform:
            raise NotImplementedError("staged gather needs a uniform schema")
        if B > 0:
            src = (C.c_void_p * len(src_ptrs))(*src_ptrs)
            _lib.check(_lib.load().dfm_embedding_forward_staged(
                plan, src, self._ptr_array(stage_out), extra_src_ptr or None, _lib.ptr(extra_dst), B, fo.data_ptr(),
                fe.data_ptr(), _lib.ptr(fm_out), _lib.ptr(fm_sum), self._err.data_ptr(), _lib.stream_handle()))

    def forward_staged_update(self, graph_exec: int, node, src_ptrs: List[int], stage_out: List[torch.Tensor], B: int,
                              fo: torch.Tensor, fe: torch.Tensor, fm_out: Optional[torch.Tensor] = None,
                              fm_sum: Optional[torch.Tensor] = None, extra_src_ptr: int = 0,
                              extra_dst: Optional[torch.Tensor] = None) -> None:
        """``forward_staged`` was captured into a HIP graph: point its kernel node (``node`` from
        ``dfm_graph_last_node``) inside the instantiated graph ``graph_exec`` at another batch record.
        Host-side only; the exec must not have a launch pending."""
        plan = self._ensure_plan(fe.device)
        src = (C.c_void_p * len(src_ptrs))(*src_ptrs)
        _lib.check(_lib.load().dfm_embedding_forward_staged_update(
            plan, C.c_void_p(graph_exec), node, src, self._ptr_array(stage_out), extra_src_ptr or None,
            _lib.ptr(extra_dst), B, fo.data_ptr(), fe.data_ptr(), _lib.ptr(fm_out), _lib.ptr(fm_sum),
            self._err.data_ptr()))

    def _launch_forward(self, inputs: List[torch.Tensor], B: int, want_fm: bool = False):
        dev = inputs[0].device
        self._ensure_plan(dev)
        F, fm = len(self.field_names), self.fm_embed_dim
        fo = torch.empty(B, 1, dtype=torch.float32, device=dev)
        fe = torch.empty(B, F, fm, dtype=torch.float32, device=dev)
        flat = ws = fm_out = None
        if not self._plan_uniform:
            flat = torch.empty(B, self.schema.total_embedding_dim, dtype=torch.float32, device=dev)
            ws = torch.empty(max(B * F, 1), dtype=torch.float32, device=dev)
        elif want_fm:
            fm_out = torch.empty(B, 1, dtype=torch.float32, device=dev)
        self.forward_into(inputs, B, fo, fe, flat, fm_out, ws)
        if B > 0 and self.strict_indices:
            self.raise_on_bad_index()
        return fo, fe, flat, fm_out

    def backward_rowsparse(self, inputs: List[torch.Tensor], g_fo: torch.Tensor, g_fe: torch.Tensor,
                           dense_grads: Dict[int, torch.Tensor], sparse: bool = True,
                           dense_slices: Optional[Tuple[torch.Tensor, int, torch.Tensor]] = None) -> None:
        """Row-sparse backward: DENSE-field Linear gradients are ADDED into ``dense_grads``
        ({id(param): buffer}); one gradient row per distinct id goes to ``self.rowsparse``
        (whose row plan must have been built from the same ``inputs``).  ``sparse=False``: the
        DENSE fields only (field-sharded tables: the SPARSE fields' gradients travel to their owners).
        ``dense_slices = (partials (parts * n), parts, flat)``: the DENSE-field gradients are computed over
        ``parts`` batch slices and STORED into ``partials`` at each element's offset inside ``flat`` (the flat
        gradient buffer whose first n elements hold every ``dense_grads`` view); whoever owns ``flat`` adds the
        slices (training/fused_step.py registers them as a slab reference)."""
        B, F, D = g_fe.shape
        if B == 0:
            return
        lib = _lib.load()
        plan = self._ensure_plan(g_fe.device)
        S = len(self._sparse_pos) if sparse else 0
        rs = self.rowsparse
        fmap = (C.c_int32 * max(S, 1))(*self._sparse_pos[:S])
        if dense_grads and self._plan_uniform and g_fe.data_ptr() % 16 == 0 and self._all_dense_grads(dense_grads):
            # DENSE-field gradients and row gradients in ONE launch (csrc/step_tail.hip)
            if self._dense_list is None or self._dense_list.device != g_fe.device:
                pos = [i for i, sp in enumerate(self.schema.fields.values()) if sp.feature_type is FeatureType.DENSE]
                self._dense_list = torch.tensor(pos, dtype=torch.int32, device=g_fe.device)
            nd = self._dense_list.numel()
            _lib.check(lib.dfm_step_embedding_backward(
                self._dense_list.data_ptr(), nd, self._ptr_array(inputs), self._grad_struct(dense_grads), fmap, S, F, D,
                B, g_fo.data_ptr(), g_fe.data_ptr(), rs.sorted_pos.data_ptr() if S else None,
                rs.seg_start.data_ptr() if S else None, rs.num_uniq.data_ptr() if S else None,
                rs.row_g2.data_ptr() if S else None, rs.row_g1.data_ptr() if S else None,
                dense_slices[0].data_ptr() if dense_slices else None, dense_slices[1] if dense_slices else 0,
                dense_slices[2].data_ptr() if dense_slices else None,
                dense_slices[0].numel() // dense_slices[1] if dense_slices else 0, _lib.stream_handle()))
            if S:
                rs.has_grad = True
            return
        if dense_grads:
            _lib.check(lib.dfm_embedding_backward_dense_fields(
                plan, self._ptr_array(inputs), B, g_fo.data_ptr(), g_fe.data_ptr(), None,
                self._grad_struct(dense_grads), _lib.stream_handle()))
        if S:
            _lib.check(lib.dfm_rowgrad_build(
                fmap, S, F, D, B, g_fo.data_ptr(), g_fe.data_ptr(), rs.sorted_pos.data_ptr(),
                rs.seg_start.data_ptr(), rs.num_uniq.data_ptr(), rs.row_g2.data_ptr(),
                rs.row_g1.data_ptr(), _lib.stream_handle()))
            rs.has_grad = True

    def _all_dense_grads(self, grads: Dict[int, torch.Tensor]) -> bool:
        """Every DENSE field has all four gradient buffers (the grouped launch does not re-check)."""
        for _, spec, second, first, _ in self._field_params():
            if spec.feature_type is FeatureType.DENSE:
                if any(id(p) not in grads for p in (second.weight, second.bias, first.weight, first.bias)):
                    return False
        return True

    def _grad_struct(self, grads: Dict[int, torch.Tensor]):
        """dfm_field_grad[] from {id(param): grad tensor}."""
        arr = (_lib.FieldGrad * len(self.field_names))()
        for i, (name, spec, second, first, proj) in enumerate(self._field_params()):
            g = arr[i]
            g.w2 = _lib.ptr(grads.get(id(second.weight)))
            g.w1 = _lib.ptr(grads.get(id(first.weight)))
            if spec.feature_type is FeatureType.DENSE:
                g.b2 = _lib.ptr(grads.get(id(second.bias)))
                g.b1 = _lib.ptr(grads.get(id(first.bias)))
            if proj is not None:
                g.proj = _lib.ptr(grads.get(id(proj.weight)))
        return arr

    def _touch_tables(self):
        """(dfm_table * S) with the second-order table pointers / row strides: the row plan's touch workgroups."""
        arr = (_lib.Table * len(self._sparse_pos))()
        for j, i in enumerate(self._sparse_pos):
            w = self.second_order_embeddings[self.field_names[i]].weight
            arr[j].w2, arr[j].stride2 = w.data_ptr(), w.stride(0)
        return arr

    def _rowplan_args(self, ids_ptrs, B: int, touch: bool):
        S = len(self._sparse_pos)
        rs = self.rowsparse
        ids = (C.c_void_p * S)(*ids_ptrs)
        specs = list(self.schema.fields.values())
        vocab = (C.c_int32 * S)(*[specs[i].vocabulary_size for i in self._sparse_pos])
        tabs = self._touch_tables() if touch else None
        keep = (ids, vocab, tabs)                    # ctypes arrays must outlive the call
        return keep, (ids, vocab, S, B, rs.sorted_pos.data_ptr(), rs.uniq_rows.data_ptr(), rs.seg_start.data_ptr(),
                      rs.num_uniq.data_ptr(), self._err.data_ptr(), tabs, self.fm_embed_dim)

    def build_rowplan(self, inputs: List[torch.Tensor], B: int, ids_ptrs: Optional[List[int]] = None,
                      touch: bool = False) -> RowSparseBuffers:
        """Row plan of the batch's ids (``inputs``: every field's input tensor, schema order — or ``ids_ptrs``: the
        device addresses of the SPARSE fields' id columns, e.g. inside a batch record).  ``touch``: the launch also
        pulls the batch's table rows and ids into the Infinity Cache (for a gather that follows it)."""
        dev = inputs[0].device
        S = len(self._sparse_pos)
        rs = self.rowsparse
        if rs is None or rs.batch != B or rs.row_g2.device != dev:
            rs = self.rowsparse = RowSparseBuffers(S, self.fm_embed_dim, B, dev)
        rs.has_grad = False
        if S == 0 or B == 0:
            return rs
        if ids_ptrs is None:
            ids_ptrs = [inputs[i].data_ptr() for i in self._sparse_pos]
        keep, args = self._rowplan_args(ids_ptrs, B, touch)
        _lib.check(_lib.load().dfm_rowplan_build(*args, _lib.stream_handle()))
        return rs

    def rowplan_update(self, graph_exec: int, node, ids_ptrs: List[int], B: int, touch: bool) -> None:
        """``build_rowplan`` was captured into a HIP graph: point its node at other id columns (host-side only)."""
        keep, args = self._rowplan_args(ids_ptrs, B, touch)
        _lib.check(_lib.load().dfm_rowplan_build_update(C.c_void_p(graph_exec), node, *args))

    # ------------------------------------------------------------------ forward
    def forward(self, batch: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        inputs, B = self._gather_inputs(batch)
        if self.grad_mode == "rowsparse":
            self._ensure_plan(inputs[0].device)
            if torch.is_grad_enabled():
                self.build_rowplan(inputs, B)
                if self._anchor is None or self._anchor.device != inputs[0].device:
                    self._anchor = torch.zeros(1, device=inputs[0].device, requires_grad=True)
                fo, fe = _RowSparseFn.apply(self, inputs, self._anchor, *self.non_table_parameters())
            else:
                fo, fe, _, _ = self._launch_forward(inputs, B)
            return fo, fe, fe.view(B, fe.shape[1] * fe.shape[2])
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            outs = _DenseGradFn.apply(self, inputs, *params)
        else:
            fo, fe, flat, _ = self._launch_forward(inputs, B)
            outs = (fo, fe) if flat is None else (fo, fe, flat)
        if len(outs) == 2:            # uniform plan: flat_embeddings is the same bytes reshaped
            return outs[0], outs[1], outs[1].view(B, outs[1].shape[1] * outs[1].shape[2])
        return outs


def _c(t: Optional[torch.Tensor], like: torch.Tensor) -> torch.Tensor:
    if t is None:
        return torch.zeros_like(like)
    return t.contiguous()


class _DenseGradFn(torch.autograd.Function):
    """Reference-semantics autograd: dense (V, d) gradients for every parameter."""

    @staticmethod
    def forward(ctx, module: FeatureEmbedding, inputs, *params):
        B = inputs[0].shape[0]
        fo, fe, flat, _ = module._launch_forward(inputs, B)
        ctx.module, ctx.inputs, ctx.params = module, inputs, params
        ctx.flat = flat
        ctx.shapes = (fo, fe)
        if flat is None:
            return fo, fe
        return fo, fe, flat

    @staticmethod
    def backward(ctx, g_fo, g_fe, g_flat=None):
        module, inputs = ctx.module, ctx.inputs
        fo, fe = ctx.shapes
        B = fe.shape[0]
        g_fo, g_fe = _c(g_fo, fo), _c(g_fe, fe)
        if ctx.flat is not None:
            g_flat = _c(g_flat, ctx.flat)
        grads = {id(p): torch.zeros_like(p) for p in ctx.params}
        if B > 0:
            plan = module._ensure_plan(fe.device)
            _lib.check(_lib.load().dfm_embedding_backward_dense(
                plan, module._ptr_array(inputs), B, g_fo.data_ptr(), g_fe.data_ptr(), _lib.ptr(g_flat),
                module._grad_struct(grads), _lib.ptr(ctx.flat), _lib.stream_handle()))
        return (None, None) + tuple(grads[id(p)] if p.requires_grad else None for p in ctx.params)


class _RowSparseFn(torch.autograd.Function):
    """Fast mode: tables are outside autograd; their row gradients go to module.rowsparse."""

    @staticmethod
    def forward(ctx, module: FeatureEmbedding, inputs, anchor, *dense_params):
        B = inputs[0].shape[0]
        fo, fe, _, _ = module._launch_forward(inputs, B)
        ctx.module, ctx.inputs, ctx.params = module, inputs, dense_params
        ctx.shapes = (fo, fe)
        return fo, fe

    @staticmethod
    def backward(ctx, g_fo, g_fe):
        module, inputs = ctx.module, ctx.inputs
        fo, fe = ctx.shapes
        B, F, D = fe.shape
        g_fo, g_fe = _c(g_fo, fo), _c(g_fe, fe)
        grads = {id(p): torch.zeros_like(p) for p in ctx.params}
        module.backward_rowsparse(inputs, g_fo, g_fe, grads)
        return (None, None, None) + tuple(grads[id(p)] for p in ctx.params)
