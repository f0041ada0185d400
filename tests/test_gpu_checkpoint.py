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


def test_reference_optimizer_state_dict_loads():
    """A ``torch.optim.Adam.state_dict()`` over ``model.parameters()`` — what the reference's trainer stores
    as ``optimizer_state_dict`` (trainer.py:140-148) — loads into RowSparseAdam: moments matched by
    parameter position, step count taken over."""
    from deepfm_amd.training.rowsparse import RowSparseAdam
    _, _, model = _small_deepfm(seed=11)
    model.embedding.set_grad_mode("dense")                   # every parameter under autograd, like the reference
    ref_opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(4)
    from tests.helpers import random_fields_batch
    fields = criteo_fields(300, 16)
    batch = {k: torch.from_numpy(v).cuda() for k, v in random_fields_batch(fields, 64, rng).items()}
    for _ in range(2):
        ref_opt.zero_grad()
        model(batch).sum().backward()
        ref_opt.step()
    sd = ref_opt.state_dict()
    assert "param_groups" in sd and isinstance(next(iter(sd["state"])), int)
    want = {n: sd["state"][i] for i, (n, _) in enumerate(model.named_parameters())}
    weights = {k: v.clone() for k, v in model.state_dict().items()}
    _, _, fresh = _small_deepfm(seed=12)
    fresh.load_state_dict(weights)
    opt = RowSparseAdam(fresh, lr=1e-3)
    opt.load_state_dict(sd)
    assert int(opt.step_count) == 2
    got = opt.state_dict()["state"]
    for name, st in want.items():
        assert torch.equal(got[name]["exp_avg"].cpu(), st["exp_avg"].cpu().reshape(got[name]["exp_avg"].shape)), name
        assert torch.equal(got[name]["exp_avg_sq"].cpu(), st["exp_avg_sq"].cpu().reshape(got[name]["exp_avg_sq"].shape)), name
