"""CPU: packed columnar batches (deepfm_amd/data/packed.py) — layout, epoch coverage, dict view."""
import numpy as np
import pytest
import torch

from deepfm_amd.data.packed import PackedBatchLoader, PackedColumns, record_layout, unpack_record
from tests.helpers import schema_from_fields
from tools_shared import criteo_fields


def _dataset(n=1000, seed=0):
    fields = criteo_fields(50, 8, n_sparse=3, n_dense=2)
    rng = np.random.default_rng(seed)
    feats = {f["name"]: (rng.integers(0, 50, n) if f["type"] == "sparse" else rng.random(n).astype(np.float32)) for f in fields}
    labels = (rng.random(n) < 0.3).astype(np.float32)
    return schema_from_fields(fields), feats, labels


def test_record_layout_matches_the_train_step_packing():
    schema, feats, labels = _dataset()
    B = 64
    ns, nd, o1, o2, nbytes = record_layout(schema, B)
    assert (ns, nd) == (3, 2) and o1 == 3 * B * 8 and o2 == o1 + 2 * B * 4 and nbytes == o2 + B * 4
    cols = PackedColumns(schema, feats, labels)
    loader = PackedBatchLoader(cols, B)
    out = np.zeros(nbytes, np.uint8)
    loader.write(out, 2)
    batch, lab = unpack_record(schema, torch.from_numpy(out), B)
    for name in schema.fields:
        want = feats[name][2 * B:3 * B]
        got = batch[name].numpy()
        assert got.dtype == (np.int64 if np.issubdtype(want.dtype, np.integer) else np.float32)
        assert np.array_equal(got, want.astype(got.dtype))
    assert np.array_equal(lab.numpy(), labels[2 * B:3 * B])


def test_shuffled_epoch_visits_every_sample_once_and_reshuffles():
    schema, feats, labels = _dataset(n=1024)
    feats["C1"] = np.arange(1024)            # sample identity rides in the first id column
    cols = PackedColumns(schema, feats, labels)
    B = 128
    loader = PackedBatchLoader(cols, B, shuffle=True, seed=5)
    _, _, o1, o2, nbytes = record_layout(schema, B)
    seen = []
    for epoch in range(2):
        loader.set_epoch(epoch)
        out = np.zeros(nbytes, np.uint8)
        got = []
        for k in range(len(loader)):
            loader.write(out, k)
            batch, lab = unpack_record(schema, torch.from_numpy(out.copy()), B)
            ids = batch["C1"].numpy()
            got.append(ids)
            assert np.array_equal(lab.numpy(), labels[ids])                 # rows stay aligned across columns
            assert np.array_equal(batch["I1"].numpy(), feats["I1"][ids])
        seen.append(np.concatenate(got))
        assert np.array_equal(np.sort(seen[-1]), np.arange(1024))
    assert not np.array_equal(seen[0], seen[1])


def test_contract_errors():
    schema, feats, labels = _dataset()
    with pytest.raises(KeyError):
        PackedColumns(schema, {k: v for k, v in feats.items() if k != "C2"}, labels)
    with pytest.raises(TypeError):
        PackedColumns(schema, dict(feats, C1=feats["C1"].astype(np.float32)), labels)
    cols = PackedColumns(schema, feats, labels)
    with pytest.raises(NotImplementedError):
        PackedBatchLoader(cols, 64, drop_last=False)
    with pytest.raises(ValueError):
        PackedBatchLoader(cols, 5000)
