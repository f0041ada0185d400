"""GPU: host batches through PackedBatchLoader -> DeviceBatchRing -> run_from train exactly like the
same batches loaded with load_batch (bitwise), with a ring shallower than the number of batches."""
import numpy as np
import pytest
import torch

from tests.helpers import npy, schema_from_fields
from tests.test_gpu_models_step import _small_deepfm
from tools_shared import criteo_fields

pytestmark = pytest.mark.gpu


def test_ring_fed_training_equals_direct_loading():
    from deepfm_amd.data.packed import DeviceBatchRing, PackedBatchLoader, PackedColumns
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    B, nb = 512, 7
    fields = criteo_fields(300, 16)
    schema = schema_from_fields(fields)
    rng = np.random.default_rng(4)
    n = B * nb
    feats = {f["name"]: (rng.integers(0, 300, n) if f["type"] == "sparse" else rng.random(n).astype(np.float32)) for f in fields}
    labels = (rng.random(n) < 0.25).astype(np.float32)
    results = []
    for ring_fed in (False, True):
        _, _, model = _small_deepfm(seed=8)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = FusedDeepFMStep(model, opt, B, use_graph=False)
        if ring_fed:
            loader = PackedBatchLoader(PackedColumns(schema, feats, labels), B)
            assert loader.record_bytes == step.packed_bytes
            for rec in DeviceBatchRing(loader, torch.device("cuda"), depth=3):
                step.run_from(rec)
        else:
            for k in range(nb):
                sl = slice(k * B, (k + 1) * B)
                ids = torch.from_numpy(np.stack([feats[f["name"]][sl] for f in fields[:26]])).cuda()
                dense = torch.from_numpy(np.stack([feats[f["name"]][sl] for f in fields[26:]])).cuda()
                step.load_batch(ids, dense, torch.from_numpy(labels[sl]).cuda())
                step.run()
        torch.cuda.synchronize()
        results.append(({k: npy(v).copy() for k, v in model.state_dict().items()}, float(step.loss)))
    assert results[0][1] == results[1][1]
    for k in results[0][0]:
        assert np.array_equal(results[0][0][k], results[1][0][k]), k
