"""Synthetic Criteo-shaped workloads: field lists as plain dicts, their ``DatasetSchema`` and
random batches in the reference's batch contract (dataset.py:28-38: SPARSE ``(B,)`` int64, SEQUENCE ``(B, L)``
int64 0-padded, DENSE ``(B,)`` float32).  Used by ``bench.py``, ``__graft_entry__.smoke()``, the tools and the
tests; nothing here touches the device."""

from __future__ import annotations

import numpy as np

from deepfm_amd.data.schema import DatasetSchema, FeatureType, FieldSchema

# Distinct values of C1..C26 in the public Criteo display-advertising (Kaggle) training set: three fields with
# fewer than 11 ids, seven with fewer than 64, four with millions — what "Criteo-shaped" means for the
# row plan and the row gradients (runs of one id inside a batch from 1 to B/3).
CRITEO_KAGGLE_CARDINALITIES = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27,
                               14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def criteo_fields(vocab, dim: int, n_sparse: int = 26, n_dense: int = 13):
    """BASELINE.json Criteo shape: C1..C26 SPARSE then I1..I13 DENSE.  ``vocab``: one
    vocabulary size for every SPARSE field, or a list with one per field."""
    vocabs = list(vocab) if isinstance(vocab, (list, tuple)) else [vocab] * n_sparse
    if len(vocabs) != n_sparse:
        raise ValueError(f"{len(vocabs)} vocabulary sizes for {n_sparse} SPARSE fields")
    fs = [dict(name=f"C{i + 1}", type="sparse", vocab=int(vocabs[i]), dim=dim, max_len=1, combiner="mean")
          for i in range(n_sparse)]
    fs += [dict(name=f"I{i + 1}", type="dense", vocab=0, dim=dim, max_len=1, combiner="mean")
           for i in range(n_dense)]
    return fs


def schema_from_fields(fields) -> DatasetSchema:
    """Plain-dict field list (the form the golden fixtures store) -> ``DatasetSchema`` (schema.py:7-59)."""
    kind = {"sparse": FeatureType.SPARSE, "dense": FeatureType.DENSE, "sequence": FeatureType.SEQUENCE}
    return DatasetSchema(fields={
        f["name"]: FieldSchema(name=f["name"], feature_type=kind[f["type"]], vocabulary_size=f["vocab"],
                               embedding_dim=f["dim"], max_length=f["max_len"], combiner=f["combiner"])
        for f in fields})


def random_fields_batch(fields, B: int, rng: np.random.Generator, zero_frac: float = 0.01):
    """One host batch in the reference's dict contract: uniform ids in [1, V) with ``zero_frac`` padding ids."""
    batch = {}
    for f in fields:
        if f["type"] == "sparse":
            x = rng.integers(1, f["vocab"], size=B, dtype=np.int64)
            x[rng.random(B) < zero_frac] = 0
        elif f["type"] == "sequence":
            x = rng.integers(0, f["vocab"], size=(B, f["max_len"]), dtype=np.int64)
        else:
            x = rng.random(B).astype(np.float32)
        batch[f["name"]] = x
    return batch
