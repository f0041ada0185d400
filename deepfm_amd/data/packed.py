"""Packed columnar batches: the input side of the hot path.

The reference feeds its models through ``TabularDataset`` + ``DataLoader`` + one ``.to(device)`` per
field (``deepfm/data/dataset.py:28-38``, ``deepfm/training/trainer.py:202-217``): a Python dict of
0-d tensors per SAMPLE, collated per batch — 3.8 K samples/s on 8 cores, four orders
of magnitude below the GPU step.  This module keeps the same data contract (a dict of per-field
numpy columns + a label column, int64 ids / float32 values) but moves batches as ONE record:

    record = [ ids (S, B) int64 | dense (Dn, B) float32 | labels (B) float32 ]      (uint8 view)

``PackedColumns``       the dataset re-laid out once, column-major, in schema order;
``PackedBatchLoader``   host iterator of batch records (shuffle / drop_last like ``DataLoader``),
                        written straight into pinned staging slots: three fancy-index gathers
                        (or three memcpys without shuffle) per batch instead of B x F tensor objects;
``DeviceBatchRing``     H2D on a copy stream into a ring of device records, overlapped with the
                        previous steps; ``RowSparseTrainStep.run_from(record)`` consumes a record
                        directly (the gather reads it and refreshes the step's static inputs);
``unpack_record``       the reference's ``dict[str, Tensor]`` view of a record, for code that calls
                        ``model(batch)``.
Uniform schemas (SPARSE and DENSE fields) only — the schemas the row-sparse step supports.
"""

from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from deepfm_amd.data.schema import DatasetSchema, FeatureType


def record_layout(schema: DatasetSchema, batch_size: int) -> Tuple[int, int, int, int, int]:
    """(n_sparse, n_dense, dense_offset, labels_offset, record_bytes) — the layout of
    ``RowSparseTrainStep.pack_batches`` (at least one slot of each kind is always present)."""
    kinds = [s.feature_type for s in schema.fields.values()]
    if any(k is FeatureType.SEQUENCE for k in kinds):
        raise NotImplementedError("packed records hold SPARSE and DENSE fields only")
    ns = sum(k is FeatureType.SPARSE for k in kinds)
    nd = sum(k is FeatureType.DENSE for k in kinds)
    o1 = max(ns, 1) * batch_size * 8
    o2 = o1 + max(nd, 1) * batch_size * 4
    return ns, nd, o1, o2, o2 + batch_size * 4


class PackedColumns:
    """The whole dataset as two column-major matrices + labels, in schema order."""

    def __init__(self, schema: DatasetSchema, features: Dict[str, np.ndarray], labels: np.ndarray) -> None:
        self.schema = schema
        n = len(labels)
        sparse, dense = [], []
        for name, spec in schema.fields.items():
            col = np.asarray(features[name])                 # KeyError for a missing field, like the reference
            if col.shape != (n,):
                raise ValueError(f"field {name!r}: expected shape ({n},), got {col.shape}")
            if spec.feature_type is FeatureType.SPARSE:
                if not np.issubdtype(col.dtype, np.integer):
                    raise TypeError(f"SPARSE field {name!r} needs integer ids, got {col.dtype}")
                sparse.append(col.astype(np.int64, copy=False))
            elif spec.feature_type is FeatureType.DENSE:
                dense.append(col.astype(np.float32, copy=False))
            else:
                raise NotImplementedError("packed records hold SPARSE and DENSE fields only")
        self.ids = np.ascontiguousarray(np.stack(sparse)) if sparse else np.zeros((0, n), np.int64)
        self.dense = np.ascontiguousarray(np.stack(dense)) if dense else np.zeros((0, n), np.float32)
        self.labels = np.ascontiguousarray(np.asarray(labels, dtype=np.float32))
        self.n = n

    def __len__(self) -> int:
        return self.n


class PackedBatchLoader:
    """Host-side batch records.  ``write(slot_bytes, k)`` fills a caller-owned (pinned) buffer with
    batch ``k`` of the current epoch; iteration order is re-drawn by ``set_epoch``."""

    def __init__(self, columns: PackedColumns, batch_size: int, shuffle: bool = False, drop_last: bool = True,
                 seed: int = 0) -> None:
        if not drop_last:
            raise NotImplementedError("the captured step has a fixed batch size: drop_last must be True")
        if batch_size <= 0 or batch_size > len(columns):
            raise ValueError("batch_size must be in [1, len(dataset)]")
        self.columns, self.batch_size, self.shuffle, self.seed = columns, batch_size, shuffle, seed
        self.ns, self.nd, self.o1, self.o2, self.record_bytes = record_layout(columns.schema, batch_size)
        self.num_batches = len(columns) // batch_size
        self.set_epoch(0)

    def set_epoch(self, epoch: int) -> None:
        n = len(self.columns)
        self.order = np.random.default_rng(self.seed + epoch).permutation(n) if self.shuffle else None

    def __len__(self) -> int:
        return self.num_batches

    def write(self, out: np.ndarray, k: int) -> None:
        """out: uint8 array of record_bytes (e.g. a numpy view of a pinned torch tensor)."""
        B, c = self.batch_size, self.columns
        if not 0 <= k < self.num_batches:
            raise IndexError(k)
        ids = out[:self.o1].view(np.int64).reshape(max(self.ns, 1), B)
        dense = out[self.o1:self.o2].view(np.float32).reshape(max(self.nd, 1), B)
        labels = out[self.o2:].view(np.float32)
        if self.order is None:
            sl = slice(k * B, (k + 1) * B)
            if self.ns:
                ids[:] = c.ids[:, sl]
            if self.nd:
                dense[:] = c.dense[:, sl]
            labels[:] = c.labels[sl]
        else:
            idx = self.order[k * B:(k + 1) * B]
            if self.ns:
                np.take(c.ids, idx, axis=1, out=ids)
            if self.nd:
                np.take(c.dense, idx, axis=1, out=dense)
            np.take(c.labels, idx, out=labels)


def unpack_record(schema: DatasetSchema, record: torch.Tensor, batch_size: int) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """(batch dict, labels) views of one record — the reference's ``model(batch)`` contract."""
    ns, nd, o1, o2, nbytes = record_layout(schema, batch_size)
    if record.numel() != nbytes or record.dtype != torch.uint8:
        raise ValueError("not a packed record of this schema / batch size")
    ids = record[:o1].view(torch.int64).view(max(ns, 1), batch_size)
    dense = record[o1:o2].view(torch.float32).view(max(nd, 1), batch_size)
    batch, si, di = {}, 0, 0
    for name, spec in schema.fields.items():
        if spec.feature_type is FeatureType.SPARSE:
            batch[name] = ids[si]; si += 1
        else:
            batch[name] = dense[di]; di += 1
    return batch, record[o2:].view(torch.float32)


class DeviceBatchRing:
    """Host -> device staging of batch records, overlapped with compute.

    ``depth`` pinned host slots and ``depth`` device records; batch k+depth-1 is packed and copied
    (copy stream) while batch k trains.  Iterating yields device records in order; a record stays
    valid until ``depth - 1`` further records have been requested (the consumer's work on it must
    have been ENQUEUED on the current stream by then, which ``step.run_from`` guarantees)."""

    def __init__(self, loader: PackedBatchLoader, device: torch.device, depth: int = 4) -> None:
        if depth < 2:
            raise ValueError("depth must be at least 2")
        self.loader, self.depth = loader, depth
        nbytes = (loader.record_bytes + 255) // 256 * 256          # records stay 256-byte aligned
        self.host = torch.empty(depth, nbytes, dtype=torch.uint8).pin_memory()
        self.host_np = [self.host[i].numpy() for i in range(depth)]
        self.dev = torch.empty(depth, nbytes, dtype=torch.uint8, device=device)
        self.copy_stream = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event() for _ in range(depth)]    # H2D of the slot has finished
        self.free = [torch.cuda.Event() for _ in range(depth)]     # consumers of the slot have finished
        self.copied = [torch.cuda.Event() for _ in range(depth)]   # host slot may be rewritten
        self._used = [False] * depth

    def _submit(self, k: int) -> None:
        slot = k % self.depth
        if self._used[slot]:
            self.copied[slot].synchronize()                         # the previous H2D out of this host slot is done
        self.loader.write(self.host_np[slot][:self.loader.record_bytes], k)
        with torch.cuda.stream(self.copy_stream):
            if self._used[slot]:
                self.copy_stream.wait_event(self.free[slot])        # the device slot is no longer being read
            self.dev[slot].copy_(self.host[slot], non_blocking=True)
            self.copied[slot].record(self.copy_stream)
            self.ready[slot].record(self.copy_stream)
        self._used[slot] = True

    def __iter__(self) -> Iterator[torch.Tensor]:
        n = len(self.loader)
        cur = torch.cuda.current_stream()
        for k in range(min(self.depth - 1, n)):
            self._submit(k)
        for k in range(n):
            if k + self.depth - 1 < n:
                # slot of batch k+depth-1 == slot of batch k-1: its consumer was enqueued one iteration ago
                prev = (k - 1) % self.depth
                if k >= 1:
                    self.free[prev].record(cur)
                self._submit(k + self.depth - 1)
            slot = k % self.depth
            cur.wait_event(self.ready[slot])
            yield self.dev[slot][:self.loader.record_bytes]
