"""GPU parity of the train-step tail (SURVEY.md §8 a14 + f-1) against REFERENCE-generated goldens.

``tests/golden/train_steps_*.npz`` (tools/make_golden.py::case_train_steps) hold three steps of the body of
the reference's ``Trainer._train_epoch`` (trainer.py:212-240): BCE + ``get_l2_reg_loss`` (base.py:78-83),
``clip_grad_norm_`` (trainer.py:232-235), ``torch.optim.Adam`` (trainer.py:67-70, 237), run on the
reference's own layer classes.  Every batch touches every table row, so the row-wise lazy step of this
package and the reference's dense step are the same computation; three step implementations are held to it:

  * exact-semantics mode: dense (V, d) autograd gradients from the HIP kernels + torch's own Adam/clip,
  * ``RowSparseTrainStep`` (autograd over the HIP ops + the fused optimizer tail kernels),
  * the fused steps (no autograd: tower kernels + the same tail), eager and as a HIP graph:
    ``FusedDeepFMStep``, ``FusedXDeepFMStep`` (xdeepfm.py:36-48) and ``FusedAttentionDeepFMStep``
    (attention_deepfm.py:48-66) — the code ``bench.py`` times for BASELINE.json's configurations 2, 3 and 4.
"""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, assert_close_mostly, cfg_of, fields_of, group, load, load_params, npy, schema_from_fields
from tests.test_gpu_models_step import _config
from tests.test_oracle_golden import TRAIN_CASES, assert_adam_moments, assert_step_params, zero_grad_param

pytestmark = pytest.mark.gpu


def _model(g):
    from deepfm_amd.models import create_model
    c = cfg_of(g)
    model = create_model(c["kind"], schema_from_fields(fields_of(g)), _config(c))
    init = group(g, "init/")
    assert sorted(model.state_dict().keys()) == sorted(init.keys())
    load_params(model, init)
    model.config.feature.embedding_l2_reg = float(g["l2"])
    return model.train()


def _pool(g, t):
    fields = fields_of(g)
    b = group(g, f"step{t}/batch/")
    ids = np.stack([b[f["name"]] for f in fields if f["type"] == "sparse"])
    dense = np.stack([b[f["name"]] for f in fields if f["type"] == "dense"])
    return torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(g[f"step{t}/labels"]).cuda()


def _state(model):
    return {k: npy(v) for k, v in model.state_dict().items()}


@pytest.mark.parametrize("case", TRAIN_CASES)
def test_l2_reg_loss_vs_reference(case):
    """BaseCTRModel.get_l2_reg_loss (base.py:78-83) in the reference-semantics (dense) mode."""
    g = load(case)
    model = _model(g)
    want = float(g["step0/l2_term"])
    assert abs(float(model.get_l2_reg_loss()) - want) <= 1e-5 * want


@pytest.mark.parametrize("case", TRAIN_CASES)
def test_exact_mode_steps_vs_reference(case):
    """Dense-gradient mode: the trainer's own lines (trainer.py:219-237) over the HIP-backed model."""
    g = load(case)
    model = _model(g)
    model.embedding.strict_indices = True
    lr, clip = float(g["lr"]), float(g["clip"])
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    for t in range(int(g["steps"])):
        b = {k: torch.from_numpy(v).cuda() for k, v in group(g, f"step{t}/batch/").items()}
        logits = model(b).squeeze(1)
        assert_close(npy(logits), g[f"step{t}/logits"], what=f"logits {t}")
        bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.from_numpy(g[f"step{t}/labels"]).cuda())
        loss = bce + model.get_l2_reg_loss()
        assert abs(float(loss) - float(g[f"step{t}/loss"])) < 1e-4 * float(g[f"step{t}/loss"])
        opt.zero_grad()
        loss.backward()
        if t == 0:
            for k, p in model.named_parameters():      # d(bce + l2)/dp incl. the full-table 2*l2*w
                # identically-zero gradients (the reference's value is summation noise): absolute floor
                # xDeepFM: gradients that pass through the CIN's split-bf16 products (2^-16 per product) are held
                # to 2e-4 relative + 2e-5 of the tensor's largest gradient, and ONE (sample, d) column may sit on
                # the other side of a ReLU kink than in the reference's summation order (train_steps_xdeepfm_l2clip
                # holds such a pre-activation: the same sample's d = 15 element in every field's table): at most 1 %
                # of a tensor's elements (one element of a 16-element DENSE-field parameter) outside.  Everything else: 1e-4 + 1e-5 of the scale, no outliers.
                cin_path = cfg_of(g)["kind"] == "xdeepfm" and (k.startswith("cin.") or k.startswith("embedding."))
                if cin_path:
                    assert_close_mostly(npy(p.grad), g[f"step0/grad/{k}"], max(0.01, 1.0 / p.numel()), rtol=2e-4, atol_scale=2e-5,
                                        what="grad " + k)
                else:
                    assert_close(npy(p.grad), g[f"step0/grad/{k}"], what="grad " + k,
                                 floor=1e-6 if zero_grad_param(k, g) else 1e-8)
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        assert abs(float(total) - float(g[f"step{t}/grad_norm"])) < 1e-4 * float(g[f"step{t}/grad_norm"])
        opt.step()
        assert_step_params(_state(model), g, t, lr, "exact mode")


@pytest.mark.parametrize("case", TRAIN_CASES)
@pytest.mark.parametrize("impl", ["autograd", "fused", "fused_graph", "fused_packed_graph"])
def test_rowsparse_steps_vs_reference(case, impl):
    from deepfm_amd.training.fused_step import fused_step_class
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    g = load(case)
    model = _model(g)
    if "packed" in impl:
        model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    lr, l2, clip = float(g["lr"]), float(g["l2"]), float(g["clip"])
    opt = RowSparseAdam(model, lr=lr, l2=l2, max_grad_norm=clip)
    B = g["step0/labels"].shape[0]
    if impl == "autograd":
        step = RowSparseTrainStep(model, opt, B, use_graph=False)
    else:
        cls = fused_step_class(model)       # FusedDeepFMStep / FusedXDeepFMStep / FusedAttentionDeepFMStep
        assert cls is not None and cls.__name__.lower() == "fused" + cfg_of(g)["kind"].replace("_", "") + "step"
        step = cls(model, opt, B, use_graph="graph" in impl)
    if step.use_graph:
        step.load_batch(*_pool(g, 0))
        step.capture()                      # must leave model and optimizer state untouched
        for k, v in _state(model).items():
            assert np.array_equal(v, group(g, "init/")[k]), f"capture() changed {k}"
    for t in range(int(g["steps"])):
        step.load_batch(*_pool(g, t))
        step.run()
        bce = float(g[f"step{t}/bce"])
        assert abs(float(step.loss) - bce) < 1e-4 * bce, (t, float(step.loss), bce)
        # clip_grad_norm_'s total norm over ALL parameters incl. the L2 gradients (trainer.py:232-235)
        norm = float(g[f"step{t}/grad_norm"])
        assert abs(float(opt.sq_norm) ** 0.5 - norm) < 1e-4 * norm
        assert abs(float(opt.clip_coef) - min(1.0, clip / (norm + 1e-6))) < 1e-4
        assert_step_params(_state(model), g, t, lr, impl)
    for k, v in _state(model).items():
        if "embeddings.C" in k:
            assert not v[0].any(), "padding row moved"
    # Adam moments of EVERY parameter (tables and dense) against torch.optim.Adam's own state
    osd = opt.state_dict()["state"]
    assert_adam_moments(lambda kind, k: npy(osd[k]["exp_avg" if kind == "m" else "exp_avg_sq"]), g, impl)
