"""GPU: reference-format checkpoints (utils/io.py:17-26, trainer.py:140-148) through the HIP-backed
modules: a checkpoint of a PACKED, trained model holds the reference's keys with contiguous tensors,
loads into a fresh (unpacked or packed) model, and resuming from it continues bit-identically."""
import numpy as np
import pytest
import torch

from tests.helpers import npy
from tests.test_gpu_models_step import _pool, _small_deepfm
from tools_shared import criteo_fields

pytestmark = pytest.mark.gpu


def _setup(packed, seed=3):
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    _, _, model = _small_deepfm(seed=seed)
    if packed:
        model.embedding.pack_tables_()
    opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
    return model, opt, FusedDeepFMStep(model, opt, 512, use_graph=False)


def test_checkpoint_round_trip_and_resume(tmp_path):
    from deepfm_amd.utils.io import load_checkpoint, save_checkpoint
    rng = np.random.default_rng(2)
    ids, dense, labels = _pool(criteo_fields(300, 16), 4, 512, rng)
    dev = lambda a: torch.from_numpy(a).cuda()
    model, opt, step = _setup(packed=True)
    for i in range(2):
        step.load_batch(dev(ids[i]), dev(dense[i]), dev(labels[i]))
        step.run()
    path = tmp_path / "best_model.pt"
    save_checkpoint({"epoch": 1, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
                     "best_metric": 0.5}, path)
    # packed tables are strided views of 4x larger buffers: the file must hold the tensors, not the storage
    n_params = sum(v.numel() for v in model.state_dict().values())
    assert path.stat().st_size < 3.5 * 4 * n_params + 1_000_000      # params + two Adam moments + slack
    ck = load_checkpoint(path, device="cpu")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "best_metric"} and ck["epoch"] == 1
    for k, v in model.state_dict().items():
        assert ck["model_state_dict"][k].is_contiguous() and tuple(ck["model_state_dict"][k].shape) == tuple(v.shape), k
        assert np.array_equal(ck["model_state_dict"][k].numpy(), npy(v)), k
    # continue training: original vs restored-into-fresh (unpacked AND packed) must agree bit for bit
    finals = []
    for variant in ("original", "fresh-unpacked", "fresh-packed"):
        if variant == "original":
            m, o, s = model, opt, step
        else:
            m, o, s = _setup(packed=variant.endswith("-packed"), seed=99)
            m.load_state_dict(ck["model_state_dict"])
            o.load_state_dict(ck["optimizer_state_dict"])
        for i in range(2, 4):
            s.load_batch(dev(ids[i]), dev(dense[i]), dev(labels[i]))
            s.run()
        torch.cuda.synchronize()
        finals.append({k: npy(v).copy() for k, v in m.state_dict().items()})
    for k in finals[0]:
        if k.endswith("num_batches_tracked"):
            continue
        assert np.array_equal(finals[0][k], finals[1][k]), f"unpacked resume diverged: {k}"
        assert np.array_equal(finals[0][k], finals[2][k]), f"packed resume diverged: {k}"
